"""Ablation timing of the fused Bottleneck kernel (csrc/bottleneck.hip) at the ResNet101 layer3 shape (N=32, 16x16, Cmid=256):
hot (one weight stream re-used: L2-resident) and cold (24 different streams in turn, as in the network).  MSC_BNECK_ABL selects the
ablated instantiation (bit 0 no MFMA, bit 1 no weight traffic, bit 2 no LDS fragment reads); run once per value."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapping_challenge_amd import _lib      # noqa: E402

lib = _lib.load()
cmid, n, h, w = int(os.environ.get('CMID', 256)), 32, int(os.environ.get('HW', 16)), int(os.environ.get('HW', 16))
cfg = int(os.environ.get('PHCFG', 0))
c4 = 4 * cmid
dt = torch.bfloat16
x = (torch.randn(n, h, w, c4, device='cuda') * 0.5).to(dt)
out = torch.empty_like(x)
co = [torch.rand(c, device='cuda') + 0.5 for c in (cmid, cmid, cmid, cmid, c4, c4)]
nb = int(lib.msc_bottleneck_pack_bytes(cmid))
streams = [(torch.randn(nb // 2, device='cuda') * 0.02).to(dt) for _ in range(24)]
st = torch.cuda.current_stream().cuda_stream


def desc(wpk):
    d = _lib.BneckDesc()
    d.x, d.out, d.wpk = x.data_ptr(), out.data_ptr(), wpk.data_ptr()
    d.scale1, d.shift1, d.scale2, d.shift2, d.scale3, d.shift3 = [c.data_ptr() for c in co]
    d.x_ld = d.out_ld = c4
    d.dtype, d.N, d.H, d.W, d.Cmid, d.cfg = _lib.BF16, n, h, w, cmid, cfg
    return d


descs = [desc(s) for s in streams]
assert lib.msc_bottleneck_ok(C.byref(descs[0])) == 1


def run(ds, reps):
    for d in ds:
        _lib.check(lib.msc_bottleneck_fused(C.byref(d), st), 'fused')
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for d in ds:
            lib.msc_bottleneck_fused(C.byref(d), st)
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / (reps * len(ds))


if os.environ.get('MSC_BNECK_ABL') == '8':
    import numpy as np
    blocks = n * (h // 2) * (w // 16)
    dbg = torch.zeros(blocks * 16, dtype=torch.int64, device='cuda')
    lib.msc_bottleneck_debug_buffer.argtypes = [C.c_void_p]
    lib.msc_bottleneck_debug_buffer(dbg.data_ptr())
    for _ in range(3):
        lib.msc_bottleneck_fused(C.byref(descs[0]), st)
    torch.cuda.synchronize()
    t = dbg.cpu().numpy().reshape(blocks, 16).astype(np.float64)
    rel = (t[:, 1:11] - t[:, :1])
    names = ['prologue issued', 'phase-1 k-loop done', 'phase-1 epilogue + barrier', 'phase-2 k-loop done', 'phase-2 epilogue + barrier',
             'pass 0 done', 'pass 1 done', 'pass 2 done', 'pass 3 done', 'pass-0 k-steps + residual wait']
    print('shader-clock stamps since block start, median over %d blocks (shader clock, ~2 GHz):' % blocks)
    for k, nm in enumerate(names):
        print('  %-32s median %8.0f   min %8.0f   max %8.0f' % (nm, np.median(rel[:, k]), rel[:, k].min(), rel[:, k].max()))
    print('  block start spread: %.0f' % (t[:, 0].max() - t[:, 0].min()))
hot = run(descs[:1], 200)
cold = run(descs, 10)
print('ABL=%s cmid=%d hw=%d cfg=%d: hot %.1f us  cold(24 streams in turn) %.1f us' % (os.environ.get('MSC_BNECK_ABL', '0'), cmid, h, cfg, hot, cold))
