"""Every valid msc_conv_igemm configuration timed on the decoder's 3x3 layers (and a few others): which tile / ring / taps-per-barrier
variant wins, and by how much -- the table the per-layer tuner keeps only the first line of.  GPU box only:
    python tools/conv_cfg_table.py [--shapes dec] > gpurun_out/conv_cfg_table.txt
Columns: configuration number, its (pixels x channels, waves, k-step bytes, ring) from the library's table, microseconds, TFLOP/s."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

import hip_ops as ops

# name, N, H, W, Cin, Cout, K, flip
SHAPES = {
    'dec': [('dec1 fwd 128->128 @128', 32, 128, 128, 128, 128, 3, 0), ('dec2 fwd 320->128 @64', 32, 64, 64, 320, 128, 3, 0),
            ('dec2 dgrad 128->320 @64', 32, 64, 64, 128, 320, 3, 1), ('dec3 fwd 768->256 @32', 32, 32, 32, 768, 256, 3, 0),
            ('dec3 dgrad 256->768 @32', 32, 32, 32, 256, 768, 3, 1), ('dec4 fwd 1280->512 @16', 32, 16, 16, 1280, 512, 3, 0),
            ('dec4 dgrad 512->1280 @16', 32, 16, 16, 512, 1280, 3, 1)],
    'enc': [('layer3 3x3 256->256 @16', 32, 16, 16, 256, 256, 3, 0), ('layer2 3x3 128->128 @32', 32, 32, 32, 128, 128, 3, 0),
            ('layer1 3x3 64->64 @64', 32, 64, 64, 64, 64, 3, 0), ('layer3 1x1 1024->256', 32, 16, 16, 1024, 256, 1, 0),
            ('layer3 1x1 256->1024', 32, 16, 16, 256, 1024, 1, 0)],
}


def time_cfg(x, w, out, k, flip, cfg, reps=20, stats=None):
    pad = k // 2
    kw = dict(relu=False, stats=stats) if stats is not None else dict(relu=True)
    try:
        ops.conv_igemm(x, w, out, stride=1, pad=pad, flip=flip, cfg=cfg, **kw)
    except Exception:
        return None
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        ops.conv_igemm(x, w, out, stride=1, pad=pad, flip=flip, cfg=cfg, **kw)
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--shapes', default='dec')
    ap.add_argument('--cfgs', default='')
    ap.add_argument('--stats', action='store_true', help='with the BatchNorm statistics epilogue (training forward)')
    args = ap.parse_args()
    only = [int(c) for c in args.cfgs.split(',') if c]
    for group in args.shapes.split(','):
        for name, N, H, W, Cin, Cout, K, flip in SHAPES[group]:
            x = (torch.randn(N, H, W, Cin, device='cuda') * 0.5).to(torch.bfloat16)
            w = (torch.randn(Cout, K, K, Cin, device='cuda') * 0.05).to(torch.bfloat16)
            out = torch.empty(N, H, W, Cout, device='cuda', dtype=torch.bfloat16)
            gf = 2.0 * N * H * W * Cin * Cout * K * K / 1e9
            rows = []
            stats = torch.zeros(8 * Cout * 2, dtype=torch.float64, device='cuda') if args.stats else None
            for cfg in ops.conv_valid_cfgs(x, w, out, stride=1, pad=K // 2):
                if only and cfg not in only:
                    continue
                t = time_cfg(x, w, out, K, flip, cfg, stats=stats)
                if t is not None:
                    rows.append((t, cfg))
            rows.sort()
            print('%s  (%.1f GFLOP)%s' % (name, gf, '  + statistics epilogue' if args.stats else ''))
            for t, cfg in rows[:12]:
                print('   cfg %2d  %8.1f us  %7.0f TFLOP/s' % (cfg, t, gf / t * 1e3))
            sys.stdout.flush()


if __name__ == '__main__':
    main()
