"""which FREE allocator block does the training program read?  After the dirty set-up and a few steps: for every free block of the caching
allocator, allocate exactly that block, fill it with NaN, run one eager forward + loss + backward of the program and look at the loss and the
gradient norm; a block whose content changes them is read by some launch.  The allocated blocks right below and above it are then named."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_configs as T
from oracle import losses_ref, unet_ref
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep, loss_forward_backward
if os.environ.get('SKIP_CATEGORY') != '1':
    T.test_category_layers_1_19_chain_matches_the_oracle_including_the_score_zip_quirk()
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
ref, net = T.build(101, 'bf16')
net.train()
opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
spec = LossSpec.mixed(T.ARCH)
step = TrainStep(net, spec, opt, use_graph=True)
for _ in range(5):
    step(x.cuda(), tgt.cuda())
torch.cuda.synchronize()
xd, td = x.cuda(), tgt.cuda()
loss, sums = torch.zeros(1, device='cuda'), torch.zeros(4, dtype=torch.float64, device='cuda')
def probe():
    prog = net.train_forward(xd)
    loss_forward_backward(prog.logits, td, spec, prog.dlogits, loss, sums)
    net.train_backward(prog)
    g = net.flat_grads
    gn, gbad = g.double().norm().item(), int((~torch.isfinite(g)).sum())
    # the optimizer's launches too (eagerly), on saved state that is put back afterwards
    from mapping_challenge_amd.unet_models import _Program, _stream_of
    p0, m0, v0, s0 = net.flat_params.clone(), opt.m.clone(), opt.v.clone(), opt.dev_state.clone()
    _Program.run(opt.launches(), _stream_of(xd.device))
    torch.cuda.synchronize()
    upd = (net.flat_params - p0).double().norm().item()
    wcopy = sum(float(t.float().abs().sum()) for t in list(net._pack['w'].values())[:40])
    net.flat_params.copy_(p0); opt.m.copy_(m0); opt.v.copy_(v0); opt.dev_state.copy_(s0)
    net._packed_version = -1
    net._refresh_weights(_stream_of(xd.device))
    torch.cuda.synchronize()
    return loss.item(), gn, gbad, upd, wcopy
base = probe()
print('baseline loss %.5f grad norm %.5g non-finite %d update norm %.5g copies %.6g' % base)
def snapshot():
    out = []
    for seg in torch.cuda.memory_snapshot():
        a = seg['address']
        for b in seg['blocks']:
            out.append((a, b['size'], b['state']))
            a += b['size']
    return sorted(out)
named = {}
def name(t, label):
    if torch.is_tensor(t) and t.is_cuda:
        named[t.untyped_storage().data_ptr()] = (label, t.untyped_storage().nbytes())
for i, t in enumerate(step.prog.keep):
    name(t, 'prog.keep[%d] %s %s' % (i, tuple(t.shape) if torch.is_tensor(t) else '', getattr(t, 'dtype', '')))
for k in ('x_in', 'logits', 'probs', 'dlogits', 'stem_dw'):
    name(getattr(step.prog, k, None), 'prog.' + k)
name(net._flat[0], 'flat params'); name(net._flat[1], 'flat grads'); name(opt.m, 'adam m'); name(opt.v, 'adam v'); name(opt.dev_state, 'opt dev_state')
for kind in ('w', 'wt'):
    for n_, t in net._pack[kind].items():
        name(t, 'pack %s %s' % (kind, n_))
for t in net._pack['keep']:
    name(t, 'pack table')
for n_, b in net.named_buffers():
    name(b, 'buffer ' + n_)
blocks = snapshot()
free = [(a, s) for a, s, st in blocks if st != 'active_allocated' and s >= 512]
print('%d free blocks: %s' % (len(free), ', '.join('%d' % s for _, s in free)))
for a, s in free:
    t = torch.full((s // 4,), float('nan'), device='cuda')
    where = 'same block' if a <= t.data_ptr() < a + s else 'ELSEWHERE (%x)' % t.data_ptr()
    r = probe()
    changed = abs(r[0] - base[0]) > 1e-3 * abs(base[0]) or not (0.5 < r[1] / base[1] < 2) or r[2] or not (0.9 < r[3] / base[3] < 1.1) or not (0.999 < r[4] / base[4] < 1.001)
    print('free block %x +%d (%s): loss %.5f grad norm %.5g non-finite %d update %.5g copies %.6g %s' % (a, s, where, r[0], r[1], r[2], r[3], r[4], '<== READ BY THE PROGRAM' if changed else ''))
    if changed:
        below = [b for b in blocks if b[0] + b[1] == a]
        above = [b for b in blocks if b[0] == a + s]
        for tag, bb in (('below', below), ('above', above)):
            for ba, bs, bst in bb:
                print('    %s: block %x +%d %s -> %s' % (tag, ba, bs, bst, named.get(ba, ('unnamed', 0))[0]))
    del t
