"""ConvTranspose2d(128 -> Cout, k4, s2, p1) + bias + ReLU at the network's shapes: every valid configuration of msc_conv_igemm timed with HIP
events (the halo-tile kernel is configuration 28), the result of each checked against configuration 0's."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import hip_ops as ops
from mapping_challenge_amd import _lib
torch.manual_seed(0)
for name, n, hw, cout, ld in (('dec2 128->128 @64x64', 32, 64, 128, 128), ('dec1 128->32 @128x128', 32, 128, 32, 32)):
    x = (torch.randn(n, hw, hw, 128, device='cuda') * 0.5).to(torch.bfloat16)
    w = (torch.randn(cout, 4, 4, 128, device='cuda') * 0.05).to(torch.bfloat16)
    bias = torch.randn(cout, device='cuda')
    out = torch.empty((n, 2 * hw, 2 * hw, ld), dtype=torch.bfloat16, device='cuda')
    ref = None
    rows = []
    for cfg in [0] + ops.conv_valid_cfgs(x, w, out, 2, 1, mode=1):
        ops.conv_igemm(x, w, out, stride=2, pad=1, mode=1, relu=True, shift=bias, cfg=cfg)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.float().clone()
        err = (out.float() - ref).abs().max().item()
        best = 1e9
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(4):
                ops.conv_igemm(x, w, out, stride=2, pad=1, mode=1, relu=True, shift=bias, cfg=cfg)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 4)
        rows.append((best, cfg, err))
    flop = 2.0 * n * hw * hw * 128 * cout * 16
    print(name, ' '.join('cfg %d: %.1f us (%.0f TF, err %.3g)' % (c, t * 1e3, flop / t / 1e9, e) for t, c, e in sorted(rows)[:6]), '| cfg 28: %s' % next(('%.1f us' % (t * 1e3) for t, c, e in rows if c == 28), 'n/a'))
