"""coverage check of the transposed-conv halo kernel (configuration 28): the output is pre-filled with NaN, every element must be written and equal
configuration 0's result, at shapes with one / two / many patches per persistent block and on recycled (dirty) allocator memory"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import hip_ops as ops
junk = torch.full((1 << 30,), float('nan'), device='cuda')      # 4 GB of NaN back into the caching allocator
del junk
torch.manual_seed(0)
bad = 0
for n, h, w, cout in ((4, 64, 64, 128), (4, 32, 32, 256), (4, 16, 16, 256), (32, 64, 64, 128), (3, 64, 64, 128), (4, 64, 64, 64), (2, 16, 32, 128), (4, 128, 128, 32), (5, 8, 16, 96), (8, 128, 128, 128)):
    x = (torch.randn(n, h, w, 128, device='cuda') * 0.5).to(torch.bfloat16)
    wt = (torch.randn(cout, 4, 4, 128, device='cuda') * 0.05).to(torch.bfloat16)
    bias = torch.randn(cout, device='cuda')
    ref = torch.full((n, 2 * h, 2 * w, cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    ops.conv_igemm(x, wt, ref, stride=2, pad=1, mode=1, relu=True, shift=bias, cfg=0)
    out = torch.full((n, 2 * h, 2 * w, cout), float('nan'), dtype=torch.bfloat16, device='cuda')
    ok28 = 28 in ops.conv_valid_cfgs(x, wt, out, 2, 1, mode=1)
    if ok28:
        ops.conv_igemm(x, wt, out, stride=2, pad=1, mode=1, relu=True, shift=bias, cfg=28)
    torch.cuda.synchronize()
    nan_ref, nan_out = int(torch.isnan(ref.float()).sum()), int(torch.isnan(out.float()).sum()) if ok28 else -1
    diff = float((out.float() - ref.float()).abs().nan_to_num(1e9).max()) if ok28 else -1
    print('N=%d %dx%d Cout=%d: cfg28 valid %s, NaN left: ref %d out %d, max |diff| %g' % (n, h, w, cout, ok28, nan_ref, nan_out, diff))
    bad += int(ok28 and (nan_out or diff))
    for c in ops.conv_valid_cfgs(x, wt, out, 2, 1, mode=1):       # every other configuration of the layer, same check
        o2 = torch.full((n, 2 * h, 2 * w, cout), float('nan'), dtype=torch.bfloat16, device='cuda')
        ops.conv_igemm(x, wt, o2, stride=2, pad=1, mode=1, relu=True, shift=bias, cfg=c)
        torch.cuda.synchronize()
        nn, dd = int(torch.isnan(o2.float()).sum()), float((o2.float() - ref.float()).abs().nan_to_num(1e9).max())
        if nn or dd:
            print('   cfg %d: NaN left %d, max |diff| %g' % (c, nn, dd)); bad += 1
print('BAD' if bad else 'all covered')
