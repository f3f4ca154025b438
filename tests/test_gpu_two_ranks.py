"""GPU, world_size 2 on ONE device: two processes share cuda:0 and talk over gloo (RCCL refuses two ranks on one GPU).
What this pins on the hardware, beyond the world-1 RCCL tests of test_gpu_unet.py: the data-parallel step with world.size > 1 --
World.from_env, the parameter broadcast, the piecewise hipGraphs with eager collectives between them, the loss-sum
all-reduce and the bucketed gradient exchange of trainer.ddp_plan -- replaces nn.DataParallel (src/models.py:65,
src/steps/pytorch/models.py:53,151-152).  The collectives themselves are gloo's, not RCCL's: the wire is not what is tested.

Known answer: both ranks get the SAME shard.  With a plain cross-entropy every rank's loss gradient is exactly half the
single-process one (the count doubles, a power of two), the backward is linear, and a + a is exact -- so in deterministic
mode the summed gradients, and the parameters after Adam, equal the single-process run bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import unet_ref, losses_ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
        'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}


def _run(world, rank, spec_name, same_shard, steps, use_graph):
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    from mapping_challenge_amd.unet_models import UNetResNet
    # the same kernels in every process: the multi-rank builder's weight-gradient grouping for the single-process twin too, and the heuristic
    # kernel configurations instead of per-process timing (two tiles of one convolution add their k-steps in different groupings)
    os.environ['MSC_WGRAD_GROUP'] = '24'
    os.environ['MSC_AUTOTUNE'] = '0'
    dev = torch.device('cuda', 0)
    net = UNetResNet(34, 2, num_filters=32, dropout_2d=0.0, pretrained=True, is_deconv=True, compute_dtype='bf16', deterministic=True)
    sd = unet_ref.seeded_state_dict(net)
    if rank != 0:                      # wrong weights on rank 1: sync_model must repair them
        sd = {k: (v + 0.25 if v.is_floating_point() else v) for k, v in sd.items()}
    net.load_state_dict(sd)
    net.flatten_parameters(dev)
    if world is not None:
        world.sync_model(net)
    net.train()
    seed = 0 if same_shard else rank
    x = unet_ref.synthetic_batch(2, 64, 64, seed=1234 + seed).to(dev)
    tgt = losses_ref.synthetic_target(2, 64, 64, seed=seed).to(dev)
    spec = LossSpec.plain_ce() if spec_name == 'ce' else LossSpec.mixed(ARCH)
    step = TrainStep(net, spec, HipAdam(net, lr=5e-4, weight_decay=1e-4), world=world, use_graph=use_graph)
    losses = [float(step(x, tgt).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    captured = step.pieces is not None if world is not None else step.graph is not None
    return losses, net.flat_params.detach().cpu().numpy().copy(), net.flat_grads.detach().cpu().numpy().copy(), captured


def _worker(rank, port, spec_name, same_shard, steps, use_graph, q, env=None):
    os.environ.update(env or {})
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2', LOCAL_RANK='0',
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from mapping_challenge_amd.distributed import World
    try:
        world = World.from_env(backend='gloo')
        assert world.size == 2 and world.rank == rank
        out = _run(world, rank, spec_name, same_shard, steps, use_graph)
        q.put((rank,) + out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:              # the parent must not wait for a rank that died
        import traceback
        q.put((rank, 'error', '%s\n%s' % (e, traceback.format_exc())))


def _two_ranks(spec_name, same_shard, steps=3, use_graph=True, env=None):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, port, spec_name, same_shard, steps, use_graph, q, env)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(2):
            item = q.get(timeout=600)
            assert item[1] != 'error', 'rank %d failed: %s' % (item[0], item[2])
            got[item[0]] = item[1:]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    return got


@pytest.fixture(autouse=True)
def _restore_env():
    saved = {k: os.environ.get(k) for k in ('MSC_WGRAD_GROUP', 'MSC_AUTOTUNE')}
    yield
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def test_two_ranks_same_shard_plain_ce_equals_the_single_process_step_bit_for_bit():
    got = _two_ranks('ce', same_shard=True)
    single = _run(None, 0, 'ce', True, 3, True)
    for r in (0, 1):
        assert got[r][3], 'rank %d did not run the piecewise hipGraphs' % r
        assert got[r][0] == single[0], (got[r][0], single[0])            # the global-batch loss of duplicated shards is the shard's loss
        assert np.array_equal(got[r][2], single[2]), 'summed gradients of rank %d differ from the single-process gradients' % r
        assert np.array_equal(got[r][1], single[1]), 'parameters of rank %d differ from the single-process run' % r
    assert single[0][-1] < single[0][0]


def test_two_ranks_different_shards_mixed_loss_stay_in_step():
    got = _two_ranks('mixed', same_shard=False, steps=4)
    assert got[0][3] and got[1][3]
    assert got[0][0] == got[1][0]                                          # one global-batch loss, the same number on both ranks
    assert np.array_equal(got[0][2], got[1][2])                            # the reduced gradients
    assert np.array_equal(got[0][1], got[1][1])                            # and so the replicas never drift
    assert np.isfinite(got[0][1]).all() and got[0][0][-1] < got[0][0][0]
    single = _run(None, 0, 'mixed', False, 4, True)                        # rank 0's shard alone: a different batch, a different trajectory
    assert not np.array_equal(single[1], got[0][1])


@pytest.fixture(scope='module')
def piecewise_run():
    """three steps of the mixed loss on different shards through the piecewise hipGraphs: the twin of the two tests below"""
    return _two_ranks('mixed', same_shard=False, steps=3, use_graph=True)


def test_two_ranks_eager_step_equals_the_piecewise_graphs(piecewise_run):
    a, b = _two_ranks('mixed', same_shard=False, steps=3, use_graph=False), piecewise_run
    assert not a[0][3] and b[0][3]
    assert a[0][0] == b[0][0]
    assert np.array_equal(a[0][1], b[0][1]) and np.array_equal(a[1][1], b[1][1])


def test_one_graph_mode_is_refused_for_a_backend_that_cannot_be_captured(piecewise_run):
    """MSC_DDP_ONE_GRAPH=1 with gloo (whose device collectives synchronise on the host: a capture around them never returns): TrainStep warns
    and runs the piecewise graphs, same results"""
    a, b = piecewise_run, _two_ranks('mixed', same_shard=False, steps=3, use_graph=True, env={'MSC_DDP_ONE_GRAPH': '1'})
    assert b[0][3] and b[1][3]
    assert a[0][0] == b[0][0]
    assert np.array_equal(a[0][1], b[0][1]) and np.array_equal(a[1][1], b[1][1])
