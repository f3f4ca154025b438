"""every device pointer the captured training step passes to a launch, checked against the caching allocator's snapshot: a pointer inside a block
the allocator holds as FREE (or outside every torch segment) is memory the graph uses but does not own."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_configs as T
from oracle import losses_ref, unet_ref
from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
if os.environ.get('SKIP_CATEGORY') != '1':
    T.test_category_layers_1_19_chain_matches_the_oracle_including_the_score_zip_quirk()
tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
ref, net = T.build(101, 'bf16')
net.train()
opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
step = TrainStep(net, LossSpec.mixed(T.ARCH), opt, use_graph=True)
for _ in range(5):
    step(x.cuda(), tgt.cuda())
torch.cuda.synchronize()
blocks = []
for seg in torch.cuda.memory_snapshot():
    a = seg['address']
    for b in seg['blocks']:
        blocks.append((a, a + b['size'], b['state']))
        a += b['size']
blocks.sort()
def state_of(p):
    for lo, hi, st in blocks:
        if lo <= p < hi:
            return st, lo, hi
    return 'OUTSIDE', 0, 0
def ptrs_of(args):
    for i, a in enumerate(args):
        if isinstance(a, int) and a > (1 << 32):
            yield 'arg%d' % i, a
        obj = getattr(a, '_obj', None)
        if obj is not None and hasattr(obj, '_fields_'):
            for name, typ in obj._fields_:
                v = getattr(obj, name)
                if typ is C.c_void_p and v:
                    yield name, v
            for extra in ('_in_bn',):
                sub = getattr(obj, extra, None)
                if sub is not None:
                    for name, typ in sub._fields_:
                        v = getattr(sub, name)
                        if typ is C.c_void_p and v:
                            yield 'in_bn.' + name, v
        if hasattr(a, 'descs'):
            for d in a.descs:
                for name in ('p', 'q', 'dw'):
                    yield 'group.' + name, getattr(d, name)
prog = step.prog
bad, seen = {}, 0
for lst_name, lst in (('fwd', prog.fwd), ('bwd', prog.bwd), ('opt', opt.launches())):
    for i, (fn, args) in enumerate(lst):
        for field, p in ptrs_of(args):
            seen += 1
            st, lo, hi = state_of(p)
            if st != 'active_allocated':
                bad.setdefault((st, lo, hi), []).append('%s[%d] %s.%s' % (lst_name, i, fn.__name__, field))
print('%d pointers checked, %d allocator blocks (%d free)' % (seen, len(blocks), sum(1 for b in blocks if b[2] != 'active_allocated')))
for (st, lo, hi), users in sorted(bad.items()):
    print('%s block [%x, %x) %d bytes: %d uses, e.g. %s' % (st, lo, hi, hi - lo, len(users), '; '.join(users[:6])))
if not bad:
    print('every pointer lies in a block the allocator holds as allocated')
