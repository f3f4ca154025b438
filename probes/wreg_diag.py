"""diagnostic for configurations 59 / 60 (halo3r.hip): structured inputs that show which taps / channels / pixels go wrong"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import torch.nn.functional as F
import hip_ops as ops
torch.set_printoptions(linewidth=200, precision=2, sci_mode=False)

def run(cfg, x_nchw, w_oihw, flip=0):
    n, cin, h, w_ = x_nchw.shape
    cout = w_oihw.shape[0]
    x = x_nchw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    wk = w_oihw.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).cuda()
    out = torch.zeros(n, h, w_, cout, dtype=torch.bfloat16, device='cuda')
    ops.conv_igemm(x, wk, out, stride=1, pad=1, flip=flip, cfg=cfg)
    torch.cuda.synchronize()
    return out.float().cpu().permute(0, 3, 1, 2)

for cfg, cout in ((60, 64), (59, 128), (42, 128)):
    cin, h = 64, 16
    print('==== cfg', cfg)
    # A: all ones
    x = torch.ones(1, cin, h, h); w = torch.ones(cout, cin, 3, 3)
    o = run(cfg, x, w); ref = F.conv2d(x, w, padding=1)
    print('A all-ones: max|d|', (o - ref).abs().max().item(), ' out[0,0] rows 0,1,15:', o[0, 0, 0, :4].tolist(), o[0, 0, 1, :4].tolist(), o[0, 0, 15, :4].tolist())
    bad = (o - ref).abs() > 1
    print('   wrong elements', int(bad.sum()), 'of', bad.numel(), ' by channel (first 16):', bad.sum((0, 2, 3))[:16].tolist(), ' by row:', bad.sum((0, 1, 3)).tolist())
    # C: weights = channel index / 16
    w = (torch.arange(cout).float() / 16).view(-1, 1, 1, 1).expand(cout, cin, 3, 3).contiguous()
    o = run(cfg, x, w); ref = F.conv2d(x, w, padding=1)
    print('C w=co/16: out[0,:,8,8] / (9*64) * 16 (should be 0..cout-1):', (o[0, :, 8, 8] / (9 * 64) * 16).round().int().tolist())
    # B: single tap, input channel 0 carries the pixel index
    xi = torch.zeros(1, cin, h, h); xi[0, 0] = (torch.arange(h * h).float().view(h, h) % 64) / 4
    for t in (0, 4, 8):
        w = torch.zeros(cout, cin, 3, 3); w[:, 0, t // 3, t % 3] = 1
        o = run(cfg, xi, w); ref = F.conv2d(xi, w, padding=1)
        print('B tap', t, 'max|d|', (o - ref).abs().max().item(), ' out[0,0,8,6:10]', o[0, 0, 8, 6:10].tolist(), 'ref', ref[0, 0, 8, 6:10].tolist())
    # D: one input channel ci nonzero, w = 1: tests the k-chunk mapping
    for ci in (0, 8, 31, 32, 63):
        xi = torch.zeros(1, cin, h, h); xi[0, ci] = 1
        w = torch.zeros(cout, cin, 3, 3); w[:, ci] = 1
        o = run(cfg, xi, w); ref = F.conv2d(xi, w, padding=1)
        print('D ci', ci, 'max|d|', (o - ref).abs().max().item(), 'out[0,0,8,8]', o[0, 0, 8, 8].item(), 'ref', ref[0, 0, 8, 8].item())
    # E: random weights over taps, x = 1 in one channel: w sum check per tap order
    w = torch.zeros(cout, cin, 3, 3); w[:, 5] = torch.arange(9).float().view(3, 3) + 1
    xi = torch.zeros(1, cin, h, h); xi[0, 5, 8, 8] = 1
    o = run(cfg, xi, w); ref = F.conv2d(xi, w, padding=1)
    print('E impulse: out[0,0,7:10,7:10]', o[0, 0, 7:10, 7:10].tolist(), 'ref', ref[0, 0, 7:10, 7:10].tolist())
