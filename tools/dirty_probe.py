"""Which gradients does live tuning on recycled (dirty) allocator memory damage?  Net A is built (and its flip = 1 data-gradient convolutions
tuned) right after 6 GB of NaN went back into the caching allocator; net B is built next in the same process (choices now cached: no tuning
launches).  Same weights, same batch: per-parameter gradients of the second step of each are compared."""
import os, sys
os.environ.setdefault('MSC_TUNE_DB', '0')
os.environ.setdefault('MSC_TUNE_ONLY', 'c1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
from mapping_challenge_amd.unet_models import UNetResNet
from oracle import losses_ref, unet_ref
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)}, 'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
spec = LossSpec.mixed(ARCH)
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31).cuda()
x = (unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5).cuda() + 2.0 * tgt[:, :1]
junk = [torch.full((n,), float('nan'), device='cuda') for n in (1 << 30, 1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 5400000, 21600000)]
del junk
sd = None
def grads(tag):
    global sd
    net = UNetResNet(101, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='bf16')
    if sd is None:
        sd = unet_ref.seeded_state_dict(net)
    net.load_state_dict(sd)
    net.train()
    loss, sums = torch.zeros(1, device='cuda'), torch.zeros(4, dtype=torch.float64, device='cuda')
    out = []
    for it in range(2):
        prog = net.train_forward(x)
        loss_forward_backward(prog.logits, tgt, spec, prog.dlogits, loss, sums)
        net.train_backward(prog)
        torch.cuda.synchronize()
        out.append((loss.item(), [g.float().clone() for g in net._grad_views()]))
    print(tag, 'losses', [o[0] for o in out], 'non-finite grads', sum(int(not torch.isfinite(g).all()) for g in out[1][1]))
    return [n for n, _ in net._trainable()], out
names, A = grads('A (dirty memory, live tuning)')
_, B = grads('B (same process, cached choices)')
for it in (0, 1):
    bad = []
    for n, ga, gb in zip(names, A[it][1], B[it][1]):
        d = (ga - gb).norm().item() / (gb.norm().item() + 1e-20)
        if not (d < 0.05):
            bad.append((d, n))
    print('step %d: %d of %d tensors differ by > 5 %%' % (it, len(bad), len(names)))
    for d, n in sorted(bad, reverse=True)[:25]:
        print('   %-46s rel %.3g' % (n, d))
