"""the failing order of tests/test_gpu_configs.py in one script: the [1, 19] post-processing test, then the training-trajectory flow with the
engine's and the oracle's FIRST update after the hand-over compared tensor by tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import test_gpu_configs as T
from oracle import losses_ref, unet_ref
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
if os.environ.get('SKIP_CATEGORY') != '1':
    T.test_category_layers_1_19_chain_matches_the_oracle_including_the_score_zip_quirk()
pre = 40
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
ref, net = T.build(101, 'bf16')
net.train()
opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
step = TrainStep(net, LossSpec.mixed(T.ARCH), opt, use_graph=True)
warm = []
for i in range(pre):
    warm.append(step(x.cuda(), tgt.cuda()).item())
    if i in (0, 1, 2, 5, 10, 20, 39):
        print('step', i, 'dev_state', [round(v, 6) for v in opt.dev_state.cpu().tolist()[:9]], 'grad norm final.weight %.4g' % net._grad_views()[-2].norm().item())
torch.cuda.synchronize()
V = os.environ.get('VARIANT', '')
if 'noload' not in V:
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
names = [n for n, _ in net._trainable()]
pw = dict(ref.named_parameters())
topt = torch.optim.Adam([pw[n] for n in names], lr=5e-4, weight_decay=1e-4)
for n, m, v in (zip(names, net.flat_views(opt.m), net.flat_views(opt.v)) if 'nomom' not in V else []):
    topt.state[pw[n]] = {'step': torch.tensor(float(opt.steps)), 'exp_avg': m.detach().cpu().contiguous().clone(), 'exp_avg_sq': v.detach().cpu().contiguous().clone()}
if 'nothreads' not in V:
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
ref.train()
p0 = {n: pw[n].detach().clone() for n in names}
nsteps = int(os.environ.get('ORACLE_STEPS', '3'))
ref_losses = []
first_ref = None
for i in range(nsteps):
    topt.zero_grad()
    loss = losses_ref.mixed_dice_ce(ref(x), tgt)
    loss.backward()
    if i == 0:
        gref = {n: pw[n].grad.detach().clone() for n in names}
    topt.step()
    if i == 0:
        first_ref = {n: pw[n].detach() - p0[n] for n in names}
    ref_losses.append(loss.item())
if first_ref is None:
    first_ref = {n: torch.ones_like(pw[n]) for n in names}; gref = {n: torch.ones_like(pw[n]) for n in names}
e0 = None
if 'fill' in V:          # no copies at all: a NaN-filled temporary of every parameter's size, one at a time (what .cpu() of a permuted view allocates)
    lo, hi = [int(t) for t in os.environ.get('FILL_RANGE', '0,1000000000').split(',')]
    hit = 0
    for n, q in net._trainable():
        if lo <= q.numel() * 4 < hi:
            t = torch.full((q.numel(),), float('nan'), device='cuda'); hit += 1
            del t
    torch.cuda.synchronize()
    print('filled', hit, 'temporaries in', (lo, hi))
elif 'noe0' not in V:
    e0 = {n: q.detach().cpu().clone() for n, q in net._trainable()}
hip = [step(x.cuda(), tgt.cuda()).item()]
geng = {n: g.detach().cpu().clone() for n, g in zip(names, net._grad_views())}
first_eng = {n: q.detach().cpu() - e0[n] for n, q in net._trainable()} if e0 is not None else first_ref
hip += [step(x.cuda(), tgt.cuda()).item() for _ in range(11)]
print('dev_state after the run:', opt.dev_state.cpu().tolist())
print('warm', ' '.join('%.3f' % v for v in warm[::6]))
print('non-finite gradient tensors', sum(int(not torch.isfinite(g).all()) for g in geng.values()), 'ratio of norms eng/ref: final.weight %.3g dec0.conv.weight %.3g layer3.5.conv2 %.3g layer1.0.conv1 %.3g' % tuple(geng[n].double().norm().item() / gref[n].double().norm().item() for n in ('final.weight', 'dec0.conv.weight', 'encoder.layer3.5.conv2.weight', 'encoder.layer1.0.conv1.weight')))
print('oracle', ' '.join('%.4f' % v for v in ref_losses), '| engine', ' '.join('%.4f' % v for v in hip))
rows = []
for n in names:
    du, dr = first_eng[n].double(), first_ref[n].double()
    gu, gr = geng[n].double(), gref[n].double()
    rows.append(((du - dr).norm().item() / (dr.norm().item() + 1e-30), (gu - gr).norm().item() / (gr.norm().item() + 1e-30), n, du.norm().item() / (dr.norm().item() + 1e-30)))
bad = sorted(rows, reverse=True)[:14]
print('first update after the hand-over, engine vs oracle: median update rel diff %.3f, median gradient rel diff %.3f' % (np.median([r[0] for r in rows]), np.median([r[1] for r in rows])))
for u, g, n, ratio in bad:
    print('   %-44s update rel %.3f (norm ratio %.2f)   gradient rel %.3f' % (n, u, ratio, g))
