"""GPU probe: achieved GB/s of the BatchNorm elementwise launches on the shapes of the ResNet101 train step (bf16)."""
import ctypes as C
import sys
import torch
sys.path.insert(0, '.')
from mapping_challenge_amd import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
SHAPES = [(524288, 64), (131072, 64), (131072, 256), (32768, 128), (32768, 512), (8192, 256), (8192, 1024), (2048, 512), (2048, 2048)]


def t(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); b.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for M, Cc in SHAPES:
    dt = _lib.BF16
    y = torch.randn(M, Cc, device='cuda').bfloat16(); out = torch.empty_like(y); res = torch.randn(M, Cc, device='cuda').bfloat16()
    dout = torch.randn(M, Cc, device='cuda').bfloat16(); dres = torch.zeros_like(y); dy = torch.empty_like(y)
    slots = torch.rand(8, Cc, 2, dtype=torch.float64, device='cuda') * M
    slots[:, :, 1] += M
    v = [torch.ones(Cc, device='cuda') for _ in range(10)]
    mb = M * Cc * 2 / 1e6

    def apply(r):
        return lambda: lib.msc_bn_apply(y.data_ptr(), Cc, res.data_ptr() if r else None, Cc if r else 0, out.data_ptr(), Cc, slots.data_ptr(), M,
                                        v[0].data_ptr(), v[1].data_ptr(), 1e-5, 0.1, v[2].data_ptr(), v[3].data_ptr(), v[4].data_ptr(), v[5].data_ptr(),
                                        v[6].data_ptr(), v[7].data_ptr(), 1, dt, M, Cc, st)

    def bwd(mask, withres):
        return lambda: lib.msc_bn_bwd_apply(dout.data_ptr(), Cc, out.data_ptr(), Cc, y.data_ptr(), Cc, mask, v[4].data_ptr(), v[5].data_ptr(), slots.data_ptr(), M,
                                            v[0].data_ptr(), v[6].data_ptr(), v[7].data_ptr(), v[8].data_ptr(), v[9].data_ptr(), dy.data_ptr(), Cc,
                                            dres.data_ptr() if withres else None, Cc if withres else 0, 0, dt, M, Cc, st)

    def red(mask):
        return lambda: lib.msc_bn_bwd_reduce(dout.data_ptr(), Cc, out.data_ptr(), Cc, y.data_ptr(), Cc, mask, v[4].data_ptr(), v[5].data_ptr(), slots.data_ptr(), dt, M, Cc, st)

    rows = [('apply', apply(False), 2), ('apply+res', apply(True), 3), ('bwd m2', bwd(2, False), 3), ('bwd m1+dres', bwd(1, True), 5),
            ('reduce m1', red(1), 3), ('reduce m2', red(2), 2)]
    print('M=%7d C=%4d (%6.1f MB): ' % (M, Cc, mb) + ' | '.join('%s %6.1f us %5.0f GB/s' % (n, us, k * mb / us * 1e3 / 1e3) for n, f, k in rows for us in [t(f)]))
