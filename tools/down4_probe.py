"""Conv2d(k4, s2, p1) 32 -> 128 at the train shape (N=32, 256x256 -> 128x128): every valid msc_conv_igemm configuration timed, with and without the
ReLU-backward / bias-sum epilogue (stats_kind 2), and configuration 59 (down4_c32_halo_kernel) compared with the first other one element by element."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch

from mapping_challenge_amd import _lib

lib = _lib.load()
n, hw, cin, cout = int(os.environ.get('N', 32)), int(os.environ.get('HW', 256)), 32, 128
ho = hw // 2
dt = torch.bfloat16
g = torch.Generator().manual_seed(1)
x = (torch.randn(n, hw, hw, cin, generator=g) * 0.5).to(dt).cuda()
w = (torch.randn(cout, 4, 4, cin, generator=g) * 0.05).to(dt).cuda()
act = torch.relu(torch.randn(n, ho, ho, cout, generator=g)).to(dt).cuda()
out = torch.empty(n, ho, ho, cout, dtype=dt, device='cuda')
st = torch.cuda.current_stream().cuda_stream


def desc(cfg, kind2):
    d = _lib.ConvDesc()
    d.in_, d.wt, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.in_ld, d.out_ld, d.dtype, d.mode = cin, cout, _lib.BF16, 0
    d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout, d.KH, d.KW, d.stride, d.pad, d.cfg = n, hw, hw, cin, ho, ho, cout, 4, 4, 2, 1, cfg
    if kind2:
        d.stats_kind, d.stats_y, d.stats_y_ld = 2, act.data_ptr(), cout
        d._stats = torch.zeros((_lib.BN_SLOTS, cout, 2), dtype=torch.float64, device='cuda')
        d.stats = d._stats.data_ptr()
    return d


def run(d, reps):
    _lib.check(lib.msc_conv_igemm(C.byref(d), st), 'conv')
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        lib.msc_conv_igemm(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps


gflop = 2.0 * n * ho * ho * cout * 16 * cin / 1e9
mb = (x.numel() + out.numel()) * 2 / 1e6
for kind2 in (False, True):
    ref = None
    rows = []
    for c in range(1, lib.msc_conv_num_cfgs() + 1):
        d = desc(c, kind2)
        if not lib.msc_conv_cfg_ok(C.byref(d), c):
            continue
        us = run(d, 20)
        res = out.clone()
        if ref is None and c != 59:
            ref = res
        rows.append((us, c, res))
    print('stats_kind 2' if kind2 else 'plain', '(%.1f GFLOP, %.0f MB in + out)' % (gflop, mb))
    for us, c, res in sorted(rows, key=lambda r: r[0])[:8]:
        err = (res.float() - ref.float()).abs().max().item()
        print('  cfg %2d  %7.1f us  %6.0f TFLOP/s  %5.2f TB/s   max |diff to first cfg| %.4f' % (c, us, gflop / us * 1e-3 * 1e3 / 1e3 * 1e3, mb / us * 1e-6 * 1e6 / 1e3 / 1e3 * 1e3, err))
