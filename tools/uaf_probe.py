"""Does a captured training graph use memory the allocator considers free?  Train a few replayed steps, then fill every cached free block of the
caching allocator with NaN (allocate assorted sizes without releasing, fill, free), train on: a loss that turns NaN or jumps means some buffer of
the program was released while captured launches still point at it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
from mapping_challenge_amd.unet_models import UNetResNet
from oracle import losses_ref, unet_ref
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)}, 'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
if os.environ.get('UAF_DIRTY') == '1':
    junk = [torch.full((n,), float('nan'), device='cuda') for n in (1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 5400000, 21600000)]
    del junk
enc = int(os.environ.get('UAF_ENC', '101'))
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
net = UNetResNet(enc, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='bf16')
net.load_state_dict(unet_ref.seeded_state_dict(net))
net.train()
opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
step = TrainStep(net, LossSpec.mixed(ARCH), opt, use_graph=True)
xd, td = x.cuda(), tgt.cuda()
a = [step(xd, td).item() for _ in range(20)]
torch.cuda.synchronize()
free_before = torch.cuda.memory_reserved() - torch.cuda.memory_allocated()
hold = []
for n in [1 << k for k in range(29, 9, -1)] * 3:          # 2 GB ... 4 KB, three of each while they last in the cache
    if torch.cuda.memory_reserved() - torch.cuda.memory_allocated() < n * 4:
        continue
    hold.append(torch.full((n,), float('nan'), device='cuda'))
torch.cuda.synchronize()
print('cached free bytes before %.1f MB, NaN-filled %.1f MB in %d blocks' % (free_before / 1e6, sum(h.numel() for h in hold) * 4 / 1e6, len(hold)))
del hold
b = [step(xd, td).item() for _ in range(20)]
print('before:', ' '.join('%.4f' % v for v in a[::2]))
print('after :', ' '.join('%.4f' % v for v in b[::2]))
