"""stem7_halo_kernel (configuration 58) at the train shape (N=32, 256x256): time with BatchNorm statistics (training) and with folded coefficients + ReLU (eval)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mapping_challenge_amd import _lib

lib = _lib.load()
n, hw = 32, 256
ho = hw // 2
dt = torch.bfloat16
st = torch.cuda.current_stream().cuda_stream
x = torch.randn(n, 3, hw, hw, device='cuda')
w = torch.randn(64, 3, 7, 7, device='cuda') * 0.1
xp = torch.empty((n, hw + 6, hw + 8, 4), dtype=dt, device='cuda')
wp = torch.empty((64, 7, 32), dtype=dt, device='cuda')
_lib.check(lib.msc_stem_prepare(x.data_ptr(), xp.data_ptr(), _lib.BF16, n, hw, hw, st), 'prepare')
_lib.check(lib.msc_stem_pack(w.data_ptr(), wp.data_ptr(), _lib.BF16, 64, st), 'pack')
out = torch.empty((n, ho, ho, 64), dtype=dt, device='cuda')
sc, sh = torch.rand(64, device='cuda') + 0.5, torch.randn(64, device='cuda')
stats = torch.zeros((_lib.BN_SLOTS, 64, 2), dtype=torch.float64, device='cuda')
for cfg in (58, 2, 6):
    for mode in ('stats', 'eval'):
        d = _lib.ConvDesc()
        d.in_, d.wt, d.out = xp.data_ptr(), wp.data_ptr(), out.data_ptr()
        d.in_ld, d.out_ld, d.dtype, d.mode = 4, 64, _lib.BF16, 0
        d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw + 6, hw + 8, 32, ho, ho, 64, 7, 1, 2, 0, cfg
        if mode == 'stats':
            d.stats = stats.data_ptr()
        else:
            d.scale, d.shift, d.relu = sc.data_ptr(), sh.data_ptr(), 1
        if not lib.msc_conv_cfg_ok(C.byref(d), cfg):
            continue
        _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'conv')
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            lib.msc_conv_igemm(C.byref(d), st)
        b.record()
        torch.cuda.synchronize()
        print('cfg %2d %-5s %6.1f us' % (cfg, mode, 1e3 * a.elapsed_time(b) / 50))
