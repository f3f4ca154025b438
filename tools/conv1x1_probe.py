"""Times the 1x1 / stride 1 layers of ResNet101 layer1-layer3 at the bench shape (batch 32, 256x256) under every valid configuration of
msc_conv_igemm: best implicit-GEMM tile against the streaming kernel (configuration 57).  MODE=fwd (statistics epilogue, as in
training), eval (scale/shift/residual/ReLU), dgrad (BatchNorm-backward sums)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapping_challenge_amd import _lib      # noqa: E402

lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
dt = torch.bfloat16
shapes = [(64, 64, 64), (64, 256, 64), (256, 64, 64), (256, 128, 64), (128, 512, 32), (512, 128, 32), (512, 256, 32), (256, 1024, 16), (512, 2048, 8)]
mode = os.environ.get('MODE', 'fwd')
for cin, cout, hw in shapes:
    n = 32
    x = (torch.randn(n, hw, hw, cin, device='cuda') * 0.5).to(dt)
    w = (torch.randn(cout, 1, 1, cin, device='cuda') * 0.05).to(dt)
    out = torch.empty(n, hw, hw, cout, device='cuda', dtype=dt)
    y = torch.randn(n, hw, hw, cout, device='cuda').to(dt)
    sc, sh = torch.rand(cout, device='cuda') + 0.5, torch.randn(cout, device='cuda') * 0.1
    stats = torch.zeros(_lib.BN_SLOTS, cout, 2, dtype=torch.float64, device='cuda')
    d = _lib.ConvDesc()
    d.in_, d.wt, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.in_ld, d.out_ld, d.dtype, d.mode = cin, cout, _lib.BF16, 0
    d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad = n, hw, hw, cin, hw, hw, cout, 1, 1, 1, 0
    if mode == 'fwd':
        d.stats = stats.data_ptr()
    elif mode == 'eval':
        d.scale, d.shift, d.res, d.res_ld, d.relu = sc.data_ptr(), sh.data_ptr(), y.data_ptr(), cout, 1
    else:
        d.stats, d.stats_kind, d.stats_y, d.stats_y_ld, d.scale, d.shift = stats.data_ptr(), 1, y.data_ptr(), cout, sc.data_ptr(), sh.data_ptr()
    res = {}
    for cfg in range(1, lib.msc_conv_num_cfgs() + 1):
        d.cfg = cfg
        if not lib.msc_conv_cfg_ok(C.byref(d), cfg):
            continue
        for _ in range(3):
            lib.msc_conv_igemm(C.byref(d), st)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            lib.msc_conv_igemm(C.byref(d), st)
        b.record()
        torch.cuda.synchronize()
        res[cfg] = 1e3 * a.elapsed_time(b) / 20
    gemm = {c: t for c, t in res.items() if c < 57}
    best = min(gemm, key=gemm.get)
    mb = n * hw * hw * (cin + cout * (2 if mode != 'fwd' else 1)) * 2 / 1e6      # input + output (+ the residual / y tensor)
    line = '%s %4d->%4d @%3d  best tile cfg %2d %6.1f us | stream57 %s | %.0f MB' % (mode, cin, cout, hw, best, gemm[best], '%6.1f' % res[57] if 57 in res else '   -  ', mb)
    line += ' -> %.2f TB/s' % (mb / min(res.values()))
    print(line, flush=True)
