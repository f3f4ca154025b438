"""CPU: host logic of the HIP engine (program construction, buffer wiring, concat slices, gradient
accumulation flags, weight layouts) checked by executing its launch lists with tests/emu.py -- a numpy
interpreter of the documented C-ABI semantics -- against the torch-CPU oracle."""
import os

import numpy as np
import pytest
import torch

import emu
from mapping_challenge_amd import unet_models as um
from oracle import unet_ref, losses_ref


@pytest.fixture()
def interpreted(monkeypatch):
    emu.install(monkeypatch.setattr)


def build(depth):
    ref = unet_ref.UNetResNetRef(depth)
    sd = unet_ref.seeded_state_dict(ref)
    ref.load_state_dict(sd)
    net = um.UNetResNet(depth, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='fp32')
    net.load_state_dict({'module.' + k: v for k, v in sd.items()})      # DataParallel-prefixed, like reference checkpoints
    net.flatten_parameters('cpu')
    return ref, net


@pytest.mark.parametrize('depth', [34, 101])
def test_eval_and_train_match_oracle(interpreted, depth):
    ref, net = build(depth)
    x = unet_ref.synthetic_batch(2, 64, 64)
    ref.eval(); net.eval()
    with torch.no_grad():
        assert (ref(x) - net(x)).abs().max() < 1e-4
    ref.train(); net.train()
    tgt = losses_ref.synthetic_target(2, 64, 64)
    loss = losses_ref.mixed_dice_ce(ref(x), tgt)
    loss.backward()
    loss2 = losses_ref.mixed_dice_ce(net(x), tgt)
    loss2.backward()
    assert abs(loss.item() - loss2.item()) < 1e-5
    pr = dict(ref.named_parameters())
    for n, p in net._trainable():
        scale = pr[n].grad.abs().max().item() + 1e-12
        assert (p.grad - pr[n].grad).abs().max().item() / scale < 2e-3, n   # tiny BN populations at 64x64
    for (n, b), (_, b2) in zip(sorted(ref.named_buffers()), sorted(net.named_buffers())):
        if 'running' in n:
            assert (b - b2).abs().max() < 1e-5, n


def test_deterministic_flag_comes_from_the_argument_the_environment_or_the_training_config(monkeypatch):
    """UNetResNet(deterministic=...) / MSC_DETERMINISTIC / training_config['deterministic'] (fixed-order gradient sums: MSC_WGRAD_ORDERED groups and
    msc_final_bwd's ordered_ws on the device; the CPU interpreter's per-layer launches have one order anyway)"""
    mk = lambda **kw: um.UNetResNet(34, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='fp32', **kw)
    monkeypatch.delenv('MSC_DETERMINISTIC', raising=False)
    assert mk().deterministic is False and mk(deterministic=True).deterministic is True
    monkeypatch.setenv('MSC_DETERMINISTIC', '1')
    assert mk().deterministic is True and mk(deterministic=False).deterministic is False
    from mapping_challenge_amd import _lib
    assert _lib.WGRAD_ORDERED == 1 and _lib.FINAL_BWD_WS_ROWS == 1024
    hdr = open(os.path.join(os.path.dirname(__file__), '..', 'include', 'msc.h')).read()
    assert 'MSC_WGRAD_ORDERED = 1' in hdr and '#define MSC_FINAL_BWD_WS_ROWS 1024' in hdr


def test_bn_on_load_launch_list_matches_oracle_and_drops_the_apply_launches(interpreted, monkeypatch):
    """MSC_BN_ON_LOAD=1 (ABI v9): bn2 + ReLU of every unfused Bottleneck ride on conv3's operand fetch (msc_conv_desc.in_bn), bn1 + ReLU on
    conv2's where that is a stride-1 3x3 on a 16-pixel-wide map (layer1 at 64x64 input) -- 33 + 3 msc_bn_apply launches fewer for ResNet101, the
    same loss, gradients and running statistics as the oracle (the activation the weight gradient reads is stored by the consuming conv)"""
    x = unet_ref.synthetic_batch(2, 64, 64)
    tgt = losses_ref.synthetic_target(2, 64, 64)
    counts = {}
    for on in ('0', '1'):
        monkeypatch.setenv('MSC_BN_ON_LOAD', on)
        monkeypatch.setenv('MSC_BN_ON_LOAD_MIN_PIXELS', '0')        # every layer
        ref, net = build(101)
        ref.train(); net.train()
        loss = losses_ref.mixed_dice_ce(ref(x), tgt)
        loss.backward()
        loss2 = losses_ref.mixed_dice_ce(net(x), tgt)
        loss2.backward()
        prog = next(iter(net._programs.values()))
        names = [getattr(fn, '__name__', str(fn)) for fn, _ in prog.fwd]
        counts[on] = (names.count('msc_bn_apply'), sum(1 for fn, a in prog.fwd if getattr(fn, '__name__', '') == 'msc_conv_igemm' and a[0]._obj.in_bn))
        assert abs(loss.item() - loss2.item()) < 1e-5
        pr = dict(ref.named_parameters())
        for n, p in net._trainable():
            scale = pr[n].grad.abs().max().item() + 1e-12
            assert (p.grad - pr[n].grad).abs().max().item() / scale < 2e-3, n
        for (n, b), (_, b2) in zip(sorted(ref.named_buffers()), sorted(net.named_buffers())):
            if 'running' in n:
                assert (b - b2).abs().max() < 1e-5, n
    assert counts['0'][1] == 0 and counts['1'][1] == 36
    assert counts['0'][0] - counts['1'][0] == 36


def test_state_dict_roundtrip_and_flat_views(interpreted):
    ref, net = build(34)
    sd = net.state_dict()
    assert set(sd) == set(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert v.shape == sd[k].shape and torch.equal(v, sd[k].cpu()), k
    # parameters are views of ONE flat buffer; conv weights are physically [Cout][KH][KW][Cin]
    w = net.encoder.layer1[0].conv1.weight
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)
    flat = net.flat_params
    assert flat.data_ptr() <= w.data_ptr() < flat.data_ptr() + flat.numel() * 4
    assert not any(n.startswith('encoder.fc') for n, _ in net._trainable())


def test_rejects_unsupported_configurations(monkeypatch):
    with pytest.raises(NotImplementedError):
        um.UNetResNet(50, 2, is_deconv=True, dropout_2d=0.0)
    with pytest.raises(NotImplementedError):
        um.UNetResNet(34, 2, is_deconv=False, dropout_2d=0.0)
    net = um.UNetResNet(34, 2, is_deconv=True, dropout_2d=0.0)
    with pytest.raises(Exception, match='no CPU path'):
        net(torch.zeros(1, 3, 64, 64))
    monkeypatch.setattr(um, '_require_device', lambda x: None)
    with pytest.raises(ValueError, match='divisible by 64'):
        net.eval()(torch.zeros(1, 3, 300, 300))       # 300x300 is not a legal network input


def test_ddp_plan_only_releases_finished_gradient_ranges(interpreted):
    """the data-parallel step all-reduces a suffix of the flat gradient buffer after each quarter of backward: every
    parameter inside a released range must have no later writer, and the ranges must tile the whole buffer"""
    from mapping_challenge_amd.trainer import ddp_plan
    ref, net = build(34)
    net.train()
    prog = net.train_forward(unet_ref.synthetic_batch(1, 64, 64))
    flat_g = net.flat_grads
    plan = ddp_plan(prog, flat_g, nchunks=4)
    base = flat_g.data_ptr()
    covered = []
    for end, lo, hi in plan:
        if lo is None:
            continue
        covered.append((lo, hi))
        for idx, ptr in prog.grad_writes:
            off = (ptr - base) // 4
            if lo <= off < hi:
                assert idx < end, (idx, end, off)
    covered.sort()
    assert covered[0][0] == 0 and covered[-1][1] == flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
    assert len(covered) >= 3          # the overlap is real: gradients are released in several pieces
    # cuts are placed by bytes: the LAST exchange (nothing left to overlap it with) carries at most a quarter of the buffer,
    # and what leaves before it does so in launch order
    assert [e for e, _, _ in plan] == sorted(e for e, _, _ in plan) and plan[-1][0] == len(prog.bwd)
    last_lo, last_hi = plan[-1][1], plan[-1][2]
    assert (last_hi - last_lo) <= 0.25 * flat_g.numel(), (last_lo, last_hi, flat_g.numel())


def test_backward_twice_for_one_forward_is_refused_and_backward_sums_start_clean(interpreted):
    """the backward consumes the saved raw conv outputs in place (dy over y), so one forward supports ONE backward: a second
    call fails loudly; the BatchNorm-backward / bias-gradient sums it accumulates atomically live in arena slots that the
    backward zeroes at its own head (forward + backward again reproduces the gradients exactly)"""
    from mapping_challenge_amd._lib import MscError
    ref, net = build(34)
    net.train()
    x = unet_ref.synthetic_batch(1, 64, 64)
    prog = net.train_forward(x)
    dl = torch.randn_like(prog.logits) * 1e-3
    net.train_backward(prog, dl)
    g1 = net.flat_grads.clone()
    with pytest.raises(MscError, match='twice'):
        net.train_backward(prog, dl)
    prog = net.train_forward(x)
    net.train_backward(prog, dl)
    assert torch.equal(g1, net.flat_grads)
    assert prog.bwd[0][0].__name__ == 'msc_memset_zero' and prog.bwd[0][1][1] > 0


def test_hipadam_state_dict_roundtrip_resume_order_and_fresh_compute_copies(interpreted):
    """HipAdam.state_dict / load_state_dict: moments, step count and learning rate (also the device-resident state a captured graph
    reads) come back, a restored optimizer continues identically -- also in the usual resume order (build model and optimizer, load
    both state dicts, THEN train: the state is applied when the flat buffers appear; round-3 advisory) -- and the fused update leaves
    the packed weight copies current: no re-pack at the next forward, equal to what a fresh pack of the new masters gives"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    ref, net = build(34)
    net.train()
    opt = HipAdam(net, lr=1e-3, weight_decay=1e-4)
    step = TrainStep(net, LossSpec.plain_ce(), opt)
    x = unet_ref.synthetic_batch(1, 64, 64)
    t = losses_ref.synthetic_target(1, 64, 64)[:, :1].contiguous()
    step(x, t); step(x, t)
    assert net._packed_version == net._version                       # the Adam launch wrote the copies
    name = 'encoder.layer1.0.conv1'
    w, wt = net._pack['w'][name].clone(), net._pack['wt'][name].clone()
    net._packed_version = -1
    net._refresh_weights(0)                                           # what msc_pack_multi makes of the same masters
    assert torch.equal(w, net._pack['w'][name]) and torch.equal(wt, net._pack['wt'][name])
    opt.set_lr(2.5e-4)
    state = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in opt.state_dict().items()}
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    params = net.flat_params.clone()
    step(x, t)
    after = net.flat_params.clone()
    # (a) a fresh optimizer on the SAME (already flattened) model, restored from the state, continues identically
    net.flat_params.copy_(params)
    net.weights_changed()
    opt2 = HipAdam(net, lr=9.0)
    opt2.load_state_dict(state)
    assert opt2.steps == 2 and opt2.lr == 2.5e-4 and opt2.param_groups[0]['lr'] == 2.5e-4
    assert float(opt2.dev_state[0]) == 2.0 and abs(float(opt2.dev_state[1]) - 2.5e-4) < 1e-10
    TrainStep(net, LossSpec.plain_ce(), opt2)(x, t)
    assert torch.equal(after, net.flat_params)
    with pytest.raises(ValueError):
        opt2.load_state_dict(dict(state, m=torch.zeros(3)))
    # (c) betas / eps / weight_decay are ARGUMENTS of already captured launches: loading a state that changes them drops the captured graphs
    # of every TrainStep bound to the optimizer (round-3 advisory); the same hyper-parameters keep them
    ts = TrainStep(net, LossSpec.plain_ce(), opt2)
    ts(x, t)
    ts.cur.graph = 'captured'                                          # stands in for a CUDAGraph (none on the CPU interpreter)
    opt2.load_state_dict(opt2.state_dict())
    assert ts.cur.graph == 'captured'
    changed = opt2.state_dict()
    changed['param_groups'] = [dict(changed['param_groups'][0], betas=(0.8, 0.99))]
    opt2.load_state_dict(changed)
    assert ts.cur.graph is None and opt2.betas == (0.8, 0.99)
    # (b) resume order: nothing is flattened yet when the state is loaded
    ref3, net3 = build(34)
    net3.load_state_dict(sd)
    net3.train()
    opt3 = HipAdam(net3, lr=9.0)
    opt3.load_state_dict(state)
    assert opt3.steps == 2 and opt3.lr == 2.5e-4
    TrainStep(net3, LossSpec.plain_ce(), opt3)(x, t)
    assert opt3.steps == 3 and torch.equal(after, net3.flat_params)


def test_dice_activation_reaches_the_loss_kernels(interpreted):
    """round-4 verdict: `dice_activation: sigmoid` (src/models.py:437-442) used to train with softmax-Dice without a word; LossSpec.mixed reads
    it, the launch carries msc_loss_cfg.dice_sigmoid, and loss + dlogits of the launch pair equal the oracle's for both activations"""
    from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)}, 'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}}
    torch.manual_seed(3)
    logits = (torch.randn(2, 2, 32, 32) * 2)
    tgt = losses_ref.synthetic_target(2, 32, 32, seed=5)
    for act in ('softmax', 'sigmoid'):
        spec = LossSpec.mixed(dict(arch, dice={'smooth': 1, 'dice_activation': act}))
        assert spec.cfg.dice_sigmoid == int(act == 'sigmoid')
        lg = logits.clone().requires_grad_(True)
        ref = losses_ref.mixed_dice_ce(lg, tgt, dice_activation=act)
        ref.backward()
        d, loss, sums = torch.empty_like(logits), torch.zeros(1), torch.zeros(4, dtype=torch.float64)
        loss_forward_backward(logits, tgt, spec, d, loss, sums)
        assert abs(loss.item() - ref.item()) < 1e-5 and torch.allclose(d, lg.grad, atol=1e-8, rtol=1e-4)
    with pytest.raises(NotImplementedError):
        LossSpec.mixed(dict(arch, dice={'dice_activation': 'tanh'}))


class _ReplayOf:
    """stands in for a captured hipGraph on the CPU interpreter: replay() executes the launches the capture recorded"""

    def __init__(self, body):
        self.body = body

    def replay(self):
        self.body()


def test_graph_replay_repacks_weights_a_loaded_state_dict_changed(interpreted):
    """round-4 advisory: the captured step reads the COMPUTE COPIES of the weights, which only the Adam launch at the end of a step
    rewrites; a load_state_dict between two replays must be followed by a repack before the next replay (and a checkpointed dynamic
    loss scale comes back whatever order optimizer and TrainStep are built in)"""
    from mapping_challenge_amd import _lib
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    x = unet_ref.synthetic_batch(1, 64, 64)
    t = losses_ref.synthetic_target(1, 64, 64)[:, :1].contiguous()
    ref, net = build(34)
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    net.train()
    opt = HipAdam(net, lr=1e-3)
    ts = TrainStep(net, LossSpec.plain_ce(), opt)
    l_first = ts(x, t).item()                       # eager first step of the shape, as under use_graph
    ts.use_graph, ts.cur.graph = True, _ReplayOf(ts._body_captured)
    ts(x, t)
    net.load_state_dict(sd0)                        # back to the initial weights between two replays
    opt.load_state_dict(dict(opt.state_dict(), m=None, v=None, steps=0))
    assert net._packed_version != net._version
    l_again = ts(x, t).item()
    assert abs(l_again - l_first) < 1e-6            # stale compute copies would give the loss of the twice-updated weights
    assert net._packed_version == net._version
    # loss scale of a checkpoint: restored into a flattened optimizer BEFORE the TrainStep is built (fp16 resume order)
    opt.set_loss_scale(512.0, dynamic=True, growth_interval=7)
    state = opt.state_dict()
    assert state['dynamic_scale'] is True and state['growth_interval'] == 7
    opt2 = HipAdam(net, lr=1e-3)
    opt2.load_state_dict(state)
    net.compute_dtype = 'fp16'                      # only the flag TrainStep looks at
    try:
        ts2 = TrainStep(net, LossSpec.plain_ce(), opt2)
    finally:
        net.compute_dtype = 'fp32'
    assert ts2.loss_scale == 512.0 and opt2.current_loss_scale() == 512.0 and opt2.dynamic_scale and opt2.growth_interval == 7
    assert float(opt2.dev_state[_lib.OPT_GROWTH]) == 7.0 and float(opt2.dev_state[_lib.OPT_GOOD]) == 0.0


def test_dynamic_loss_scale_skips_an_overflowed_step(interpreted):
    """fp16 `fit()` safety (the device-side protocol, run here on the interpreter in fp32 arithmetic): with a dynamic scale an
    overflowing gradient leaves parameters, moments and the step count untouched and halves the scale; the next clean step trains"""
    from mapping_challenge_amd import _lib
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    ref, net = build(34)
    net.train()
    opt = HipAdam(net, lr=1e-3)
    step = TrainStep(net, LossSpec.plain_ce(), opt)
    opt.set_loss_scale(1024.0, dynamic=True, growth_interval=2)
    x = unet_ref.synthetic_batch(1, 64, 64)
    t = losses_ref.synthetic_target(1, 64, 64)[:, :1].contiguous()
    l0 = step(x, t).item()
    assert opt.steps == 1 and opt.current_loss_scale() == 1024.0
    assert any(fn.__name__ == 'msc_grad_check' for fn, _ in opt.launches())
    before = net.flat_params.clone()
    bad = x.clone()
    bad[0, 0, 0, 0] = float('inf')
    step(bad, t)
    assert torch.equal(before, net.flat_params) and opt.steps == 1 and opt.skipped_steps == 1 and opt.current_loss_scale() == 512.0
    step(x, t); step(x, t)
    assert opt.steps == 3 and opt.current_loss_scale() == 1024.0       # two clean steps: doubled again
    assert not torch.equal(before, net.flat_params) and np.isfinite(l0)
    # the same gradients at any scale: scaled dlogits, divided out inside the Adam kernel
    ref2, net2 = build(34)
    net2.train()
    opt3 = HipAdam(net2, lr=1e-3)
    s3 = TrainStep(net2, LossSpec.plain_ce(), opt3)
    s3(x, t)
    ref3, net3 = build(34)
    net3.train()
    opt4 = HipAdam(net3, lr=1e-3)
    s4 = TrainStep(net3, LossSpec.plain_ce(), opt4)
    opt4.set_loss_scale(64.0, dynamic=True, growth_interval=1)      # the scale doubles at EVERY tick: the update must still divide by
    s4(x, t)                                                        # the scale the loss gradient was multiplied with
    assert opt4.current_loss_scale() == 128.0
    assert torch.allclose(net2.flat_params, net3.flat_params, atol=1e-6)


def test_residual_join_reduces_ride_on_the_last_writer(interpreted, monkeypatch):
    """ABI v6 host logic: for every residual join inside a stage the data-gradient conv that accumulates the last addend of the join's
    gradient carries the BatchNorm-backward sums (stats_kind 1 + stats_z + residual) and msc_bn_bwd_reduce is not launched -- 12 of
    ResNet34's 20 (29 of ResNet101's 41) since round 3; round 4 adds the three stage ends, whose last writer is the transposed-mode data
    gradient of the next stage's downsample branch (statistics in transposed mode), and takes the stem's sums from the fused pool
    backward.  What remains: the layer4 end (last writer: the centre max-pool backward) and the BatchNorms of the downsample branches
    (their gradient is written by a BatchNorm kernel).  Gradients equal the unfused program's."""
    x = unet_ref.synthetic_batch(1, 64, 64)
    tgt = losses_ref.synthetic_target(1, 64, 64)
    grads, counts = [], []
    for fuse in ('1', '0'):
        monkeypatch.setenv('MSC_FUSE_JOIN_BWD', fuse)
        ref, net = build(34)
        net.train()
        losses_ref.mixed_dice_ce(net(x), tgt).backward()
        prog = next(iter(net._programs.values()))
        names = [getattr(f, '__name__', '') for f, _ in prog.bwd]
        joins = sum(1 for f, a in prog.bwd if getattr(f, '__name__', '') == 'msc_conv_igemm' and a[0]._obj.stats_z)
        counts.append((names.count('msc_bn_bwd_reduce'), joins))
        grads.append({n: p.grad.clone() for n, p in net._trainable()})
    assert counts == [(1, 15), (16, 0)]      # round 4: stem (fused pool backward), three stage ends (transposed-mode data gradient of the downsample branch), three downsample BatchNorms (bn3's msc_bn_bwd_apply); left: the layer4 end
    for n, g in grads[0].items():
        scale = grads[1][n].abs().max().item() + 1e-12
        assert (g - grads[1][n]).abs().max().item() / scale < 1e-5, n
