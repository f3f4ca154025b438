"""GPU probe: the fixed cost of a dependent launch -- a trivial torch kernel, msc_bn_apply on a 64-pixel tensor (launch + its statistics
prologue) and on the 4 MB layer3 tensor, back to back on one stream and replayed from a hipGraph."""
import sys
import torch
sys.path.insert(0, '.')
from mapping_challenge_amd import _lib
lib = _lib.load()
s = torch.cuda.Stream()
torch.cuda.set_stream(s)
st = s.cuda_stream


def bn(M, Cc):
    y = torch.randn(M, Cc, device='cuda').bfloat16(); out = torch.empty_like(y)
    slots = torch.rand(8, Cc, 2, dtype=torch.float64, device='cuda') * M
    slots[:, :, 1] += M
    v = [torch.ones(Cc, device='cuda') for _ in range(10)]
    keep = (y, out, slots, v)
    return lambda: lib.msc_bn_apply(y.data_ptr(), Cc, None, 0, out.data_ptr(), Cc, slots.data_ptr(), M, v[0].data_ptr(), v[1].data_ptr(), 1e-5, 0.1,
                                    v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), v[5].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), 1, _lib.BF16, M, Cc, st), keep


one = torch.zeros(1, device='cuda')
cases = [('torch add_ on 1 element', lambda: one.add_(1.0), None)]
for M, Cc in ((64, 64), (8192, 256), (8192, 1024), (131072, 256)):
    f, keep = bn(M, Cc)
    cases.append(('msc_bn_apply M=%d C=%d' % (M, Cc), f, keep))
for name, f, _ in cases:
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(s)
    for _ in range(200):
        f()
    b.record(s); b.synchronize()
    eager = a.elapsed_time(b) / 200 * 1e3
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(200):
            f()
    g.replay(); torch.cuda.synchronize()
    a.record(s); g.replay(); b.record(s); b.synchronize()
    print('%-34s stream %6.2f us/launch   graph %6.2f us/launch' % (name, eager, a.elapsed_time(b) / 200 * 1e3), flush=True)
