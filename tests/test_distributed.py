"""CPU, world_size 2 over gloo: the data-parallel protocol (loss-sum all-reduce between the loss phases,
bucketed gradient all-reduce, parameter broadcast, sharding) reproduces the single-process full-batch
result -- the semantics nn.DataParallel gives the reference (loss on the gathered batch, per-replica BN)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _start_ranks(procs, world_size):
    """start the rank processes with the host's cores divided between them: W processes of torch / BLAS with all cores each spend their time in
    each other's spin-waits (the two-rank protocol test took 72 s that way, 10 s with 4 threads per rank)"""
    threads = str(max(1, (os.cpu_count() or 2) // world_size))
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    os.environ.update({k: threads for k in saved})
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _worker(rank, world_size, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    from mapping_challenge_amd.distributed import World
    from oracle import unet_ref, losses_ref
    world = World.from_env(backend='gloo')
    torch.manual_seed(0)
    n, hw = 4, 64
    x = unet_ref.synthetic_batch(n, hw, hw)
    tgt = losses_ref.synthetic_target(n, hw, hw)
    lo, hi = world.shard(n)
    net = unet_ref.UNetResNetRef(34)
    sd = unet_ref.seeded_state_dict(net)
    if rank != 0:                                   # wrong weights on rank 1: sync_model must repair them
        sd = {k: v + 1 for k, v in sd.items()}
    net.load_state_dict(sd)
    world.sync_model(net)
    net.train()
    out = net(x[lo:hi])
    t = tgt[lo:hi]
    # phase 1: the four sums of msc_loss_sums, here from the oracle's pieces
    p1 = torch.softmax(out, 1)[:, 1]
    t1 = (t[:, 0].long() == 1).float()
    w = losses_ref.loss_weights(t[:, 1:], 50., 10., (256, 256))
    ce = torch.nn.functional.cross_entropy(out, t[:, 0].long(), reduction='none')
    sums = torch.stack([(w * ce).sum(), (p1 * t1).sum(), p1.sum(), t1.sum()]).double()
    local = sums.detach().clone()
    world.all_reduce(local)
    # phase 2: global-batch loss from the reduced sums; gradient of the LOCAL contribution to it
    total = float(n * hw * hw)
    s0 = sums[0] + (local[0] - sums[0].detach())
    A = 2 * (sums[1] + (local[1] - sums[1].detach())) + 1.0
    B = (sums[2] + (local[2] - sums[2].detach())) + local[3] + 1.0 + 1e-7
    loss = 1.0 * s0 / total + 0.2 * (1 - A / B)
    loss.backward()
    params = [p for n_, p in net.named_parameters() if not n_.startswith('encoder.fc')]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    world.all_reduce_grads(flat, bucket_bytes=1 << 20)
    q.put((rank, float(loss.item()), flat.numpy(), lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_protocol_matches_full_batch():
    from oracle import unet_ref, losses_ref
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    _start_ranks(procs, 2)
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    assert [(r[3], r[4]) for r in res] == [(0, 2), (2, 4)]
    # single process: full batch loss; BN per replica means per-shard forward, loss on the gathered logits
    n, hw = 4, 64
    x = unet_ref.synthetic_batch(n, hw, hw)
    tgt = losses_ref.synthetic_target(n, hw, hw)
    net = unet_ref.UNetResNetRef(34)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    net.train()
    out = torch.cat([net(x[0:2]), net(x[2:4])])
    loss = losses_ref.mixed_dice_ce(out, tgt)
    loss.backward()
    flat = torch.cat([p.grad.reshape(-1) for n_, p in net.named_parameters() if not n_.startswith('encoder.fc')]).numpy()
    assert abs(res[0][1] - loss.item()) < 1e-5 and abs(res[1][1] - loss.item()) < 1e-5
    assert np.allclose(res[0][2], res[1][2])
    assert np.abs(res[0][2] - flat).max() <= 1e-4 * np.abs(flat).max()


def _engine_worker(rank, world_size, port, q, wire='fp32'):
    """the PRODUCT's distributed backward (TrainStep._backward_overlapped: backward in pieces, asynchronous all-reduce of the
    gradient range each piece finishes) on the CPU interpreter of the launch lists, over gloo"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    os.environ.pop('MSC_GRAD_WIRE', None)
    import emu
    import mapping_challenge_amd.unet_models as um
    from mapping_challenge_amd.distributed import World, wire_for
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    from oracle import unet_ref, losses_ref
    emu.install()
    world = World.from_env(backend='gloo')
    assert world.grad_wire == 'fp32'                # the constructor's default ...
    if wire != 'fp32':
        world.grad_wire = wire_for(wire)            # ... and, as bench.py does, the 16-bit wire chosen AFTER construction (W = 2: the
        assert world.grad_wire == 'bf16'            # all_to_all_single / all_gather_into_tensor branch really runs between two ranks)
    n, hw = (4 if world_size == 2 else world_size), 64
    x = unet_ref.synthetic_batch(n, hw, hw)
    tgt = losses_ref.synthetic_target(n, hw, hw)
    lo, hi = world.shard(n)
    net = um.UNetResNet(34, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='fp32')
    net.load_state_dict(unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34)))
    net.flatten_parameters('cpu')
    net.train()
    prog = net.train_forward(x[lo:hi])
    # dlogits of the GLOBAL-batch loss for the local logits (the two-phase loss kernels are HIP; here from the oracle's pieces)
    out = prog.logits.clone().requires_grad_(True)
    t = tgt[lo:hi]
    p1 = torch.softmax(out, 1)[:, 1]
    t1 = (t[:, 0].long() == 1).float()
    w = losses_ref.loss_weights(t[:, 1:], 50., 10., (256, 256))
    ce = torch.nn.functional.cross_entropy(out, t[:, 0].long(), reduction='none')
    sums = torch.stack([(w * ce).sum(), (p1 * t1).sum(), p1.sum(), t1.sum()]).double()
    glob = sums.detach().clone()
    world.all_reduce(glob)
    s = [sums[i] + (glob[i] - sums[i].detach()) for i in range(3)]
    loss = 1.0 * s[0] / float(n * hw * hw) + 0.2 * (1 - (2 * s[1] + 1.0) / (s[2] + glob[3] + 1.0 + 1e-7))
    loss.backward()
    prog.dlogits.copy_(out.grad)
    step = TrainStep(net, LossSpec.plain_ce(), HipAdam(net, lr=0.0), world=world)
    step._backward_overlapped(prog)
    plan = prog._ddp_plan
    grads = {name: gv.numpy().copy() for (name, _), gv in zip(net._trainable(), net._grad_views())}
    q.put((rank, float(loss.item()), grads, sum(1 for _, a, _ in plan if a is not None)))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize('wire,W', [('fp32', 2), ('bf16', 3), ('fp32', 4), ('fp32', 8)])       # 8: one image per rank, the node size bench.py --gpus 8 runs at
def test_product_backward_in_pieces_with_async_allreduce_matches_full_batch(wire, W):
    """wire 'bf16': the 16-bit gradient exchange (cast -> all-to-all -> fp32 accumulate, one rounding -> all-gather -> widen);
    W = 3: a world size that divides none of the gradient ranges -- every exchanged range is padded to W shards of a multiple of 8
    elements (distributed.all_reduce_grad_range), the padding must neither leak into the sum nor shift a shard"""
    from oracle import unet_ref, losses_ref
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if wire == 'bf16' else 0) + 13 * (W - 2)
    procs = [ctx.Process(target=_engine_worker, args=(r, W, port, q, wire)) for r in range(W)]
    _start_ranks(procs, W)                                     # the ranks share the host's cores: no BLAS oversubscription
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    n, hw = (4 if W == 2 else W), 64
    per = (n + W - 1) // W
    x = unet_ref.synthetic_batch(n, hw, hw)
    tgt = losses_ref.synthetic_target(n, hw, hw)
    ref = unet_ref.UNetResNetRef(34)
    ref.load_state_dict(unet_ref.seeded_state_dict(ref))
    ref.train()
    loss = losses_ref.mixed_dice_ce(torch.cat([ref(x[r * per:(r + 1) * per]) for r in range(W)]), tgt)      # per-replica BatchNorm
    loss.backward()
    pr = dict(ref.named_parameters())
    assert abs(res[0][1] - loss.item()) < 1e-5
    assert res[0][3] >= 3                                      # several pieces really went out while backward continued
    assert set(res[0][2]) == set(res[1][2]) and len(res[0][2]) > 100
    for name, g in res[0][2].items():
        for r in range(1, W):
            assert np.array_equal(g, res[r][2][name]), name    # every rank holds the same reduced gradient
        gref = pr[name].grad.numpy()
        if wire == 'fp32':
            err = np.abs(g - gref).max() / (np.abs(gref).max() + 1e-12)
            assert err < 2e-3, (name, err)
        else:
            # every rank's partial gradient is rounded to bf16 (2^-9 relative to the PARTIAL, which cancellation can make
            # larger than the sum), then the sum once more: bounded relative to the tensor's largest element
            assert np.abs(g - gref).max() <= ((W + 2) * 2.0 ** -8 + 2e-3) * np.abs(gref).max() + 1e-12, (name, np.abs(g - gref).max() / np.abs(gref).max())
            assert np.array_equal(g, g.astype(np.float32)) and (g.view(np.uint32) & 0xffff == 0).all(), name   # bf16-representable


def _fit_worker(rank, world_size, port, q, tmp):
    """the PRODUCT's fit() (callback protocol, validation, best checkpoint, early stopping) as one of two ranks over gloo,
    on the CPU interpreter of the launch lists; the ranks validate on DIFFERENT data, so rank-local decisions would diverge"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world_size))
    import emu
    import mapping_challenge_amd.unet_models as um
    from mapping_challenge_amd import models as hip_models
    from mapping_challenge_amd.distributed import World
    from oracle import unet_ref, losses_ref
    emu.install(models=True)
    World.from_env(backend='gloo')
    ckpt = os.path.join(tmp, 'checkpoints', 'best.torch')
    arch = {'model_params': {'encoder': 'ResNet34', 'compute_dtype': 'fp32'}, 'optimizer_params': {'lr': 5e-4},
            'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4}}
    cb = {'model_checkpoint': {'filepath': ckpt, 'epoch_every': 1, 'minimize': True},
          'validation_monitor': {'epoch_every': 1}, 'early_stopping': {'patience': 0, 'minimize': True}}
    t = hip_models.PyTorchUNet(arch, {'epochs': 3}, cb)
    assert t.world.size == 2 and t.world.rank == rank
    sd = unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34))
    if rank == 1:                                   # fit() must start from rank 0's weights (World.sync_model)
        sd = {k: (v + 0.5 if v.dtype.is_floating_point else v) for k, v in sd.items()}
    t.model.load_state_dict(sd)
    hw = 64
    train = [[unet_ref.synthetic_batch(1, hw, hw, seed=100 + 10 * b + rank), losses_ref.synthetic_target(1, hw, hw, seed=200 + 10 * b + rank)[:, :1].contiguous()]
             for b in range(2)]
    valid = [[unet_ref.synthetic_batch(1, hw, hw, seed=300 + rank), losses_ref.synthetic_target(1, hw, hw, seed=400 + rank)[:, :1].contiguous()]]
    # the rank-LOCAL validation loss of the initial weights, to show below that what the callbacks saw was not it
    t.fit((train, len(train)), validation_datagen=(valid, len(valid)))
    vals = [float(t.validation_loss[e]['sum']) for e in sorted(t.validation_loss)]
    es = [c for c in t.callbacks.callbacks if type(c).__name__ == 'EarlyStopping'][0]
    q.put((rank, len(t.epoch_losses), vals, bool(es.training_break()), t.model.flat_params.clone().numpy(),
           sorted(os.listdir(os.path.dirname(ckpt)))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fit_agrees_on_validation_checkpoints_on_rank0_and_stops_together(tmp_path):
    """every rank saw its own validation loss, saved the same file and could leave the epoch loop alone"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    threads = str(max(1, (os.cpu_count() or 2) // 2))
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    os.environ.update({k: threads for k in saved})
    try:
        for p in procs:
            p.start()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
    (r0, n0, v0, b0, p0, f0), (r1, n1, v1, b1, p1, f1) = res
    assert n0 == n1 and 1 <= n0 <= 3                  # both ranks ran the same number of epochs
    assert v0 == v1 and len(v0) == n0                 # ... on the same (rank-averaged) validation loss
    assert b0 == b1                                   # ... and took the same early-stopping decision
    assert np.array_equal(p0, p1)                     # parameters stayed in lock-step (broadcast at start, identical gradients)
    assert f0 == ['best.torch'] and f1 == ['best.torch']      # one file, no temporaries left behind
    ckpt = torch.load(os.path.join(str(tmp_path), 'checkpoints', 'best.torch'))
    assert all(k.startswith('module.') for k in ckpt)
