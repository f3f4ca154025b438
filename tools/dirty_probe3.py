"""the training-trajectory test's flow on a dirty allocator, with the hand-over actions separated: 40 replayed steps, then ACTION, then 30 more.
ACTION: none | sd (state_dict -> .cpu() of every tensor: GPU temporaries for the permuted views) | sleep | alloc (assorted GPU allocations)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mapping_challenge_amd import postprocessing as post
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
from mapping_challenge_amd.unet_models import UNetResNet
from oracle import losses_ref, unet_ref, post_ref
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)}, 'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
probs = post_ref.synthetic_probs(3, 256, 256, seed=91)
post.postprocess_batch(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=[1, 19])
post.postprocess_device(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=[1, 19])
for i in range(3):
    r = post.resize_image(probs[i], (300, 300))
    post.threshold_batch(torch.from_numpy(r[None]).cuda(), [1, 19])
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
net = UNetResNet(101, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='bf16')
net.load_state_dict(unet_ref.seeded_state_dict(net))
net.train()
opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
step = TrainStep(net, LossSpec.mixed(ARCH), opt, use_graph=True)
a = [step(x.cuda(), tgt.cuda()).item() for _ in range(40)]
torch.cuda.synchronize()
action = os.environ.get('ACTION', 'none')
if action == 'sd':
    keep = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    mv = [m.detach().cpu().contiguous().clone() for m in net.flat_views(opt.m)] + [v.detach().cpu().contiguous().clone() for v in net.flat_views(opt.v)]
    n_steps = opt.steps
elif action == 'sleep':
    time.sleep(30)
elif action == 'alloc':
    hold = [torch.full((n,), float('nan'), device='cuda') for n in [1 << k for k in range(24, 9, -1)] * 4]
    torch.cuda.synchronize()
    del hold
b = [step(x.cuda(), tgt.cuda()).item() for _ in range(30)]
print('ACTION=%-5s before %s | after %s' % (action, ' '.join('%.3f' % v for v in a[-6:]), ' '.join('%.3f' % v for v in b[::3])))
