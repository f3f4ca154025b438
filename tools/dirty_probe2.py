"""twin training runs in one process: A built (and tuned) on a dirty allocator, B built afterwards with the cached choices; same weights, same
batch, 70 replayed steps each.  Prints both loss curves: where (if anywhere) A leaves B."""
import os, sys
os.environ.setdefault('MSC_TUNE_DB', '0')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from mapping_challenge_amd import postprocessing as post
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
from mapping_challenge_amd.unet_models import UNetResNet
from oracle import losses_ref, unet_ref, post_ref
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)}, 'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
mode = os.environ.get('DIRTY', 'post')
if mode == 'post':        # what tests/test_gpu_configs.py::test_category_layers_1_19... leaves behind
    probs = post_ref.synthetic_probs(3, 256, 256, seed=91)
    for _ in range(2):
        post.postprocess_batch(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=[1, 19])
        post.postprocess_device(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=[1, 19])
elif mode == 'nan':
    junk = [torch.full((n,), float('nan'), device='cuda') for n in (1 << 30, 1 << 28, 1 << 26, 1 << 24, 1 << 22, 1 << 20, 5400000, 21600000)]
    del junk
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
sd = None
def run(tag, n=70):
    global sd
    net = UNetResNet(101, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='bf16')
    if sd is None:
        sd = unet_ref.seeded_state_dict(net)
    net.load_state_dict(sd)
    net.train()
    step = TrainStep(net, LossSpec.mixed(ARCH), HipAdam(net, lr=5e-4, weight_decay=1e-4), use_graph=True)
    out = [step(x.cuda(), tgt.cuda()).item() for _ in range(n)]
    print(tag, ' '.join('%.4f' % v for v in out[::5]))
    return np.array(out)
a = run('A dirty+tuned ')
b = run('B cached      ')
c = run('C cached again')
rel = np.abs(a - b) / np.abs(b)
print('A vs B: max rel %.4f at step %d; first step above 2 %%: %s' % (rel.max(), int(rel.argmax()), next((i for i, r in enumerate(rel) if r > 0.02), None)))
rel = np.abs(c - b) / np.abs(b)
print('C vs B: max rel %.4f at step %d' % (rel.max(), int(rel.argmax())))
