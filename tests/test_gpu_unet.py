"""GPU: the whole network through the HIP engine against the torch-CPU oracle and the golden vectors
produced by the reference's own UNetResNet.  fp32 mode: logits within 1e-4 (the north star's bound);
bf16 mode: documented looser bound + mask agreement."""
import os

import numpy as np
import pytest
import torch

from oracle import unet_ref, losses_ref

pytestmark = pytest.mark.gpu


def build(depth, dtype, sd=None):
    from mapping_challenge_amd.unet_models import UNetResNet
    ref = unet_ref.UNetResNetRef(depth)
    sd = sd or unet_ref.seeded_state_dict(ref)
    ref.load_state_dict(sd)
    net = UNetResNet(depth, 2, num_filters=32, dropout_2d=0.0, pretrained=True, is_deconv=True, compute_dtype=dtype)
    net.load_state_dict(sd)
    net.flatten_parameters('cuda')
    return ref, net


@pytest.mark.parametrize('depth', [34, 101])
def test_eval_logits_fp32_within_1e4_of_reference_golden(golden_dir, depth):
    g = np.load(os.path.join(golden_dir, 'unet_r%d_64.npz' % depth))
    ref, net = build(depth, 'fp32')
    n = g['logits_eval'].shape[0]
    x = unet_ref.synthetic_batch(n, 64, 64)
    net.eval()
    with torch.no_grad():
        y = net(x.cuda()).cpu().numpy()
    assert y.shape == g['logits_eval'].shape
    assert np.abs(y - g['logits_eval']).max() < 1e-4


@pytest.mark.parametrize('depth,hw,n', [(34, 128, 3), (101, 128, 2), (152, 64, 2), (34, 256, 2)])
def test_eval_logits_fp32_vs_oracle(depth, hw, n):
    ref, net = build(depth, 'fp32')
    x = unet_ref.synthetic_batch(n, hw, hw, seed=5)
    ref.eval(); net.eval()
    with torch.no_grad():
        yr = ref(x)
        yh = net(x.cuda()).cpu()
        ph = net.predict_proba(x.cuda()).cpu()
    assert (yr - yh).abs().max().item() < 1e-4
    assert (torch.softmax(yr, 1) - ph).abs().max().item() < 3e-5     # softmax of logits that agree to 1e-4


@pytest.mark.parametrize('depth', [34, 101])
def test_eval_bf16_logits_within_the_derived_16bit_bound(depth):
    """bf16 storage of every activation: d tensors stored one after the other on the longest path of the eval network (BatchNorm
    folded: one per conv), independent roundings of unit roundoff u = 2^-8  =>  relative L2 error of the logits <= u * sqrt(d)
    (the same derivation tests/test_gpu_parity_timed.py uses at the timed size)"""
    import math
    ref, net = build(depth, 'bf16')
    x = unet_ref.synthetic_batch(2, 128, 128, seed=6)
    ref.eval()
    with torch.no_grad():
        yr = ref(x)
    yh = net.eval()(x.cuda()).cpu()
    d = 1 + {34: 16 * 2, 101: 33 * 3}[depth] + 13
    err = (yh.double() - yr.double()).norm().item() / yr.double().norm().item()
    assert err < 2.0 ** -8 * math.sqrt(d), (err, d)
    ph, pr = net.predict_proba(x.cuda()).cpu(), torch.softmax(yr, 1)
    assert (pr - ph).abs().mean().item() < 2.0 ** -8 * math.sqrt(d) / 4          # softmax slope <= 1/4


@pytest.mark.parametrize('depth', [34, 101])
def test_train_step_fp32_matches_reference_golden(golden_dir, depth):
    """forward (batch-stat BN) + mixed loss + backward: loss, selected gradients and BN running stats against
    the reference's own modules (golden), all gradients against the oracle"""
    from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
    g = np.load(os.path.join(golden_dir, 'unet_r%d_64.npz' % depth))
    ref, net = build(depth, 'fp32')
    n = g['logits_train'].shape[0]
    x = unet_ref.synthetic_batch(n, 64, 64)
    tgt = losses_ref.synthetic_target(n, 64, 64)
    net.train()
    prog = net.train_forward(x.cuda())
    assert np.abs(prog.logits.cpu().numpy() - g['logits_train']).max() < 2e-4
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    loss = torch.zeros(1, device='cuda')
    sums = torch.zeros(4, dtype=torch.float64, device='cuda')
    loss_forward_backward(prog.logits, tgt.cuda(), LossSpec.mixed(arch), prog.dlogits, loss, sums)
    net.train_backward(prog)
    assert abs(loss.item() - float(g['loss'])) < 1e-4
    grads = {n_: gv.cpu() for (n_, _), gv in zip(net._trainable(), net._grad_views())}

    assert np.allclose(net.encoder.bn1.running_mean.cpu().numpy(), g['rm_bn1'], atol=1e-5)
    assert np.allclose(net.encoder.bn1.running_var.cpu().numpy(), g['rv_bn1'], atol=1e-5)
    if depth != 34:
        # 64x64 tiles leave ResNet101's layer4 BatchNorms 8 samples per channel (ill-conditioned: a rounding difference in a
        # near-zero variance is amplified by 1/sqrt(var + eps)); the tight per-element checks run at 128x128 / batch 4 below,
        # against the golden the reference produced there.  Here: every tensor against the oracle with a bound on the
        # distribution of the per-tensor errors instead of on each one
        ref.train()
        losses_ref.mixed_dice_ce(ref(x), tgt).backward()
        errs = sorted((grads[n_] - p.grad).norm().item() / (p.grad.norm().item() + 1e-12)
                      for n_, p in ref.named_parameters() if n_ in grads and p.grad is not None)
        assert len(errs) > 300 and all(np.isfinite(errs))
        # measured: median 5.9e-3, 90th percentile 1.5e-2, worst 1.9e-2
        assert errs[len(errs) // 2] < 1e-2 and errs[int(0.9 * len(errs))] < 3e-2 and errs[-1] < 0.1, (errs[len(errs) // 2], errs[int(0.9 * len(errs))], errs[-1])
        return

    def close(a, b, rel=2e-3):
        return (a - torch.from_numpy(b)).abs().max().item() <= rel * (np.abs(b).max() + 1e-12)
    assert close(grads['final.weight'], g['g_final_w']) and close(grads['final.bias'], g['g_final_b'])
    assert close(grads['encoder.conv1.weight'][:8], g['g_conv1'])
    assert close(grads['encoder.bn1.weight'], g['g_bn1_w'])
    assert close(grads['dec1.block.1.weight'][:4, :4], g['g_dec1_deconv'])
    assert close(grads['center.block.0.conv.bias'], g['g_center_conv_b'])
    assert close(grads['encoder.layer2.0.conv1.weight'][:4, :8], g['g_l2_conv1'])
    # every gradient against the oracle (tensor-wise relative L2 error)
    ref.train()
    lr = losses_ref.mixed_dice_ce(ref(x), tgt)
    lr.backward()
    for n_, p in ref.named_parameters():
        if n_ in grads and p.grad is not None:
            err = (grads[n_] - p.grad).norm().item() / (p.grad.norm().item() + 1e-12)
            assert err < 5e-3, (n_, err)


def test_train_step_fp32_resnet101_128_matches_reference_golden_digest(golden_dir):
    """ResNet101, 128x128, batch 4: loss, logits and EVERY parameter gradient (L2 norm and leading elements) against what
    the reference's own UNetResNet + mixed loss + autograd produced (tests/golden/make_golden.py).  Gradient bound: the
    reference's fp32 backward is itself 5.2e-3 (median over tensors) / 1.1e-2 (worst) away from a float64 evaluation at this
    configuration (ReLU masks flip under last-bit differences; tests/test_gpu_parity_timed.py holds the engine to that same
    distance from the float64 gradient), so two fp32 implementations agree to about twice that, not to 2e-3"""
    from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
    g = np.load(os.path.join(golden_dir, 'unet_r101_128.npz'))
    ref, net = build(101, 'fp32')
    x = unet_ref.synthetic_batch(4, 128, 128, seed=12)
    tgt = losses_ref.synthetic_target(4, 128, 128, seed=12)
    net.train()
    prog = net.train_forward(x.cuda())
    assert np.abs(prog.logits.cpu().numpy() - g['logits_train']).max() < 2e-4
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    loss = torch.zeros(1, device='cuda')
    sums = torch.zeros(4, dtype=torch.float64, device='cuda')
    loss_forward_backward(prog.logits, tgt.cuda(), LossSpec.mixed(arch), prog.dlogits, loss, sums)
    net.train_backward(prog)
    assert abs(loss.item() - float(g['loss'])) < 1e-4
    checked = 0
    for (name, _), gv in zip(net._trainable(), net._grad_views()):
        flat = gv.contiguous().reshape(-1).cpu().double()
        norm = float(g['n|' + name])
        assert abs(flat.norm().item() - norm) <= 1e-2 * norm + 1e-12, (name, flat.norm().item(), norm)
        head = torch.from_numpy(g['h|' + name]).double()
        assert (flat[:head.numel()] - head).abs().max().item() <= 2.5e-2 * norm / np.sqrt(flat.numel()) * 8 + 1e-12, name
        checked += 1
    assert checked == len([k for k in g.files if k.startswith('n|')]) and checked > 300


def test_autograd_node_drives_reference_style_loop():
    """loss.backward(); optimizer.step() exactly as src/steps/pytorch/models.py:104-111 does, with torch's Adam"""
    ref, net = build(34, 'fp32')
    x = unet_ref.synthetic_batch(2, 64, 64)
    tgt = losses_ref.synthetic_target(2, 64, 64)
    net.train(); ref.train()
    opt_h = torch.optim.Adam([p for _, p in net._trainable()], lr=5e-4, weight_decay=1e-4)
    opt_r = torch.optim.Adam([p for n_, p in ref.named_parameters() if not n_.startswith('encoder.fc')], lr=5e-4, weight_decay=1e-4)
    for _ in range(2):
        opt_h.zero_grad(); opt_r.zero_grad()
        lh = losses_ref.mixed_dice_ce(net(x.cuda()), tgt.cuda())
        lh.backward(); opt_h.step(); net.weights_changed()
        lr = losses_ref.mixed_dice_ce(ref(x), tgt)
        lr.backward(); opt_r.step()
        assert abs(lh.item() - lr.item()) < 1e-3
    a = dict(ref.named_parameters())['dec0.conv.weight']
    b = dict(net.named_parameters())['dec0.conv.weight'].detach().cpu()
    assert (a - b).abs().max().item() < 5e-4


def test_hip_train_loop_graph_equals_eager_and_tracks_oracle():
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    x = unet_ref.synthetic_batch(2, 64, 64).cuda()
    tgt = losses_ref.synthetic_target(2, 64, 64).cuda()
    out = {}
    for mode in ('eager', 'graph'):
        ref, net = build(34, 'fp32')
        net.train()
        step = TrainStep(net, LossSpec.mixed(arch), HipAdam(net, lr=5e-4, weight_decay=1e-4), use_graph=(mode == 'graph'))
        losses = [step(x, tgt).item() for _ in range(4)]
        out[mode] = (losses, net.flat_params.clone())
    # the weight-gradient kernel accumulates with fp32 atomics (order varies run to run) and Adam's normalised
    # update amplifies last-bit gradient differences, so the two trajectories agree closely, not bitwise
    assert np.allclose(out['eager'][0], out['graph'][0], rtol=2e-3)
    assert (out['eager'][1] - out['graph'][1]).abs().mean().item() < 2e-5
    # oracle: same 4 steps with torch Adam
    ref.train()
    opt = torch.optim.Adam([p for n_, p in ref.named_parameters() if not n_.startswith('encoder.fc')], lr=5e-4, weight_decay=1e-4)
    ref_losses = []
    for _ in range(4):
        opt.zero_grad()
        l = losses_ref.mixed_dice_ce(ref(x.cpu()), tgt.cpu())
        l.backward(); opt.step()
        ref_losses.append(l.item())
    assert np.allclose(out['eager'][0], ref_losses, rtol=5e-3)
    assert out['eager'][0][-1] < out['eager'][0][0]


@pytest.mark.parametrize('dtype,depth', [('bf16', 34), ('fp32', 34), ('bf16', 101)])
def test_deterministic_mode_repeats_a_training_run_bit_for_bit(dtype, depth):
    """UNetResNet(deterministic=True): the weight gradients' partial sums are added in a fixed order (MSC_WGRAD_ORDERED,
    msc_final_bwd ordered_ws) -- two runs of the same four steps end in the same bits, eager and as a replayed graph; the
    statistics sums go through fp64 atomics of fp32 partials, which are exact at these sizes.  (The per-layer kernel choices are timed
    once per process and shape, so the three builds run the same kernels.)"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    x = unet_ref.synthetic_batch(4, 64, 64).cuda()
    tgt = losses_ref.synthetic_target(4, 64, 64).cuda()
    runs = []
    for mode in ('graph', 'graph', 'eager'):
        _, net = build(depth, dtype)
        net.deterministic = True
        net.train()
        step = TrainStep(net, LossSpec.mixed(arch), HipAdam(net, lr=5e-4, weight_decay=1e-4), use_graph=(mode == 'graph'))
        losses = [step(x, tgt).item() for _ in range(4)]
        runs.append((losses, net.flat_params.clone()))
    assert runs[0][0] == runs[1][0] == runs[2][0]
    assert torch.equal(runs[0][1], runs[1][1])
    assert torch.equal(runs[0][1], runs[2][1])
    assert runs[0][0][-1] < runs[0][0][0]


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_bn_on_load_training_matches_the_separate_apply_launches(dtype, monkeypatch):
    """MSC_BN_ON_LOAD=1: bn2 + ReLU of every unfused Bottleneck applied by conv3 on load (msc_conv_desc.in_bn) -- the same four training steps
    as with the msc_bn_apply launches: losses and parameters agree to the rounding of a different summation order in conv3, the running
    statistics agree, and the forward holds 36 msc_bn_apply launches fewer (conv3 of the 33 blocks, conv2 of layer1's three at 64x64)"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    x = unet_ref.synthetic_batch(4, 64, 64).cuda()
    tgt = losses_ref.synthetic_target(4, 64, 64)[:, :1].contiguous().cuda()
    runs = {}
    for on in ('0', '1'):
        monkeypatch.setenv('MSC_BN_ON_LOAD', on)
        monkeypatch.setenv('MSC_BN_ON_LOAD_MIN_PIXELS', '0')        # every layer
        _, net = build(101, dtype)
        net.deterministic = True
        net.train()
        step = TrainStep(net, LossSpec.plain_ce(), HipAdam(net, lr=1e-3), use_graph=True)
        losses = [step(x, tgt).item()]
        bn2 = net.encoder.layer3[5].bn2
        first = (bn2.running_mean.clone(), bn2.running_var.clone(), net.encoder.layer1[0].bn1.running_var.clone())     # after ONE step
        losses += [step(x, tgt).item() for _ in range(3)]
        prog = next(p for k, p in net._programs.items() if p.training)
        nb = sum(1 for fn, _ in prog.fwd if fn.__name__ == 'msc_bn_apply')
        runs[on] = (losses, net.flat_params.clone(), first, nb)
    assert runs['0'][3] - runs['1'][3] == 36
    # the first step's forward differs only by the summation order inside conv3 / conv2 (same bits in, measured by the kernel tests): loss and
    # the running statistics the consumer's prologue now writes agree tightly.  (Four Adam steps at lr 1e-3 on 64-pixel BatchNorm populations
    # amplify a last-bit difference -- sign flips of near-zero gradients move a weight by 2 lr -- so later steps are held to a band, as the
    # first GPU run of this test showed: running_var of layer3 after four steps differs by up to 5 % between the two modes.)
    assert abs(runs['0'][0][0] - runs['1'][0][0]) < 2e-3 * abs(runs['0'][0][0])
    for a, b in zip(runs['0'][2], runs['1'][2]):
        # measured: 0.15-0.2 % on layer3's running mean, 22 blocks of 16-bit storage deep (round 6: 1.7e-4 absolute on entries of 1e-2 once the
        # statistics' row sums changed their summation order -- the two paths round differently from the first BatchNorm on, the band is the noise of that)
        assert torch.allclose(a, b, rtol=1e-2, atol=5e-4)
    assert np.allclose(runs['0'][0], runs['1'][0], rtol=3e-2)
    assert (runs['0'][1] - runs['1'][1]).abs().mean().item() < 1e-3        # four Adam steps of 1e-3 each; measured 4.9e-4
    assert runs['1'][0][-1] < runs['1'][0][0]


def test_bf16_train_step_runs_and_reduces_loss():
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    ref, net = build(34, 'bf16')
    net.train()
    step = TrainStep(net, LossSpec.plain_ce(), HipAdam(net, lr=1e-3))
    x = unet_ref.synthetic_batch(4, 64, 64).cuda()
    tgt = losses_ref.synthetic_target(4, 64, 64)[:, :1].contiguous().cuda()
    losses = [step(x, tgt).item() for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_fp16_training_with_the_dynamic_loss_scale_inside_the_captured_graph():
    """fp16 `fit()` safety on the device (ABI v7): TrainStep in fp16 uses a DYNAMIC loss scale kept in device memory -- the captured
    hipGraph holds msc_grad_check / msc_adam_tick / msc_adam_pack and replays with the current scale.  Clean steps train; a scale
    that overflows the fp16 activations' gradients makes the step a no-op (parameters, step count) and halves the scale until
    training resumes -- without re-capturing."""
    from mapping_challenge_amd import _lib
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    _, net = build(34, 'fp16')
    net.train()
    opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
    step = TrainStep(net, LossSpec.mixed(arch), opt, use_graph=True)
    assert opt.dynamic_scale and opt.current_loss_scale() == 4096.0
    x = unet_ref.synthetic_batch(2, 64, 64).cuda()
    t = losses_ref.synthetic_target(2, 64, 64).cuda()
    losses = [step(x, t).item() for _ in range(4)]          # first call eager + capture, then replays
    assert step.graph is not None and np.isfinite(losses).all() and losses[-1] < losses[0]
    assert opt.steps == 4 and opt.skipped_steps == 0
    before = net.flat_params.clone()
    opt.dev_state[_lib.OPT_SCALE] = 2.0 ** 40                 # dlogits * 2^40 overflows fp16 in the first backward layer
    step(x, t)
    assert torch.equal(before, net.flat_params) and opt.steps == 4 and opt.skipped_steps == 1
    assert opt.current_loss_scale() == 2.0 ** 39
    opt.dev_state[_lib.OPT_SCALE] = 4096.0
    l5 = step(x, t).item()
    assert opt.steps == 5 and not torch.equal(before, net.flat_params) and np.isfinite(l5)
    assert torch.isfinite(net.flat_params).all()


def test_rccl_world1_overlapped_backward_equals_plain():
    """RCCL smoke (one GPU is all gpurun exposes): process group of size 1 on backend nccl; the piecewise backward
    with asynchronous per-piece all-reduce must give the gradients of the plain backward"""
    import os
    import torch.distributed as dist
    from mapping_challenge_amd.distributed import World
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 1000))
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        world = World()
        ref, net = build(34, 'fp32')
        net.train()
        x = unet_ref.synthetic_batch(2, 64, 64).cuda()
        tgt = losses_ref.synthetic_target(2, 64, 64).cuda()
        step = TrainStep(net, LossSpec.plain_ce(), HipAdam(net, lr=0.0), world=world)
        step._setup(x, tgt[:, :1].contiguous())
        step.x.copy_(x); step.t.copy_(tgt[:, :1])
        from mapping_challenge_amd.trainer import loss_forward_backward

        def fwd_loss():
            prog = net.train_forward(step.x)
            loss_forward_backward(prog.logits, step.t, step.spec, prog.dlogits, step.loss, step.sums, world)
            return prog
        # backward consumes its saved activations in place (dy overwrites y), so each backward needs its own forward;
        # momentum 0.1 running statistics are the only state a forward changes and they do not enter the gradients
        net.train_backward(fwd_loss())
        plain = net.flat_grads.clone()
        step._backward_overlapped(fwd_loss())
        torch.cuda.synchronize()
        scale = plain.abs().max().item()
        assert (net.flat_grads - plain).abs().max().item() <= 1e-5 * scale
        world.sync_model(net)
    finally:
        dist.destroy_process_group()


def test_rccl_world1_piecewise_graph_step_equals_eager():
    """the multi-GPU training step (graphs around eager RCCL collectives) on a process group of size 1: the same loss
    trajectory and parameters as the plain eager step"""
    import os
    import torch.distributed as dist
    from mapping_challenge_amd.distributed import World
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 1000))
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
                'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
        x = unet_ref.synthetic_batch(2, 64, 64).cuda()
        tgt = losses_ref.synthetic_target(2, 64, 64).cuda()
        out = {}
        for mode in ('eager', 'pieces'):
            ref, net = build(34, 'fp32')
            net.train()
            kw = dict(world=World(), use_graph=True, force_collectives=True) if mode == 'pieces' else {}
            step = TrainStep(net, LossSpec.mixed(arch), HipAdam(net, lr=5e-4, weight_decay=1e-4), **kw)
            losses = [step(x, tgt).item() for _ in range(4)]
            if mode == 'pieces':
                assert step.pieces is not None and len(step.pieces[1]) == 4
            out[mode] = (losses, net.flat_params.clone())
        assert np.allclose(out['eager'][0], out['pieces'][0], rtol=2e-3)
        assert (out['eager'][1] - out['pieces'][1]).abs().mean().item() < 2e-5
        assert out['pieces'][0][-1] < out['pieces'][0][0]
    finally:
        dist.destroy_process_group()


def test_rccl_world1_one_graph_step_equals_the_piecewise_graphs(monkeypatch):
    """round 6: the distributed step as ONE hipGraph with the RCCL calls captured inside (MSC_DDP_ONE_GRAPH=1) on a process group of size 1:
    the capture succeeds (no silent fall-back), and six replayed steps give the loss trajectory and parameters of the piecewise-graph step
    -- bit for bit in the deterministic mode, the launches being the same ones in the same order"""
    import os
    import torch.distributed as dist
    from mapping_challenge_amd.distributed import World
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 1000))
        dist.init_process_group('nccl', rank=0, world_size=1)
    try:
        arch = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
                'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
        x = unet_ref.synthetic_batch(2, 64, 64).cuda()
        tgt = losses_ref.synthetic_target(2, 64, 64).cuda()
        out = {}
        for mode in ('pieces', 'one'):
            monkeypatch.setenv('MSC_DDP_ONE_GRAPH', '1' if mode == 'one' else '0')
            ref, net = build(34, 'bf16')
            net.deterministic = True
            net.train()
            step = TrainStep(net, LossSpec.mixed(arch), HipAdam(net, lr=5e-4, weight_decay=1e-4), world=World(), use_graph=True, force_collectives=True)
            losses = [step(x, tgt).item() for _ in range(6)]
            if mode == 'one':
                assert step.cur.one is not None and step.pieces is None
            else:
                assert step.pieces is not None and step.cur.one is None
            out[mode] = (losses, net.flat_params.clone(), net.flat_grads.clone())
        assert out['one'][0] == out['pieces'][0]
        assert torch.equal(out['one'][1], out['pieces'][1]) and torch.equal(out['one'][2], out['pieces'][2])
        assert out['one'][0][-1] < out['one'][0][0]
    finally:
        dist.destroy_process_group()


def test_ddp_plan_of_the_multi_gpu_program_shape_resnet101(monkeypatch):
    """the backward as torch.distributed runs it (weight gradients grouped 24 layers at a time, MSC_WGRAD_GROUP=24) for the
    timed network: the gradient exchange is cut by bytes -- several pieces leave while backward still runs, the decoder's
    share early, and the LAST one (nothing left to hide it behind) carries at most a quarter of the buffer"""
    from mapping_challenge_amd.trainer import ddp_plan
    monkeypatch.setenv('MSC_WGRAD_GROUP', '24')
    ref, net = build(101, 'bf16')
    net.train()
    prog = net.train_forward(unet_ref.synthetic_batch(2, 64, 64).cuda())
    assert sum(1 for fn, _ in prog.bwd if fn.__name__ == 'msc_wgrad_group_run') >= 4
    flat_g = net.flat_grads
    plan = ddp_plan(prog, flat_g)
    n, base = flat_g.numel(), flat_g.data_ptr()
    sent = [(end, lo, hi) for end, lo, hi in plan if lo is not None]
    assert len(plan) == 4 and len(sent) >= 3 and plan[-1][0] == len(prog.bwd)
    assert sent[0][2] == n and sent[-1][1] == 0 and all(a[1] == b[2] for a, b in zip(sent, sent[1:]))      # the ranges tile the buffer
    for end, lo, hi in sent:                      # nothing inside a released range has a later writer
        assert all(idx < end for idx, ptr in prog.grad_writes if lo <= (ptr - base) // 4 < hi)
    assert (sent[-1][2] - sent[-1][1]) <= 0.25 * n, [(e, (h - l) / n) for e, l, h in sent]
    assert (sent[0][2] - sent[0][1]) >= 0.35 * n and sent[0][0] <= 0.6 * len(prog.bwd), [(e, (h - l) / n) for e, l, h in sent]


@pytest.mark.parametrize('depth,dtype', [(101, 'bf16'), (152, 'fp16')])
def test_eval_with_fused_bottlenecks_equals_the_three_launch_path(monkeypatch, depth, dtype):
    """eval forward with the identity Bottlenecks of layer1..layer3 fused into one launch each (the default) against the same
    network run layer by layer (MSC_FUSE_BNECK=0): same storage points, same products, fp32 accumulation in a different order"""
    x = unet_ref.synthetic_batch(2, 256, 256, seed=8).cuda()
    outs, launches = {}, {}
    for flag in ('1', '0'):
        monkeypatch.setenv('MSC_FUSE_BNECK', flag)
        ref, net = build(depth, dtype)
        outs[flag] = net.eval()(x).float().cpu()
        prog = net._program(2, 256, 256, False, x.device)
        launches[flag] = [fn.__name__ for fn, _ in prog.fwd]
    nb = {101: 3 + 4 + 23, 152: 3 + 8 + 36}[depth] - 3            # identity blocks of layer1..layer3
    assert launches['1'].count('msc_bottleneck_fused') == nb and 'msc_bottleneck_fused' not in launches['0']
    assert len(launches['0']) - len(launches['1']) == 2 * nb
    err = (outs['1'] - outs['0']).norm().item() / outs['0'].norm().item()
    assert torch.isfinite(outs['1']).all() and err < {'bf16': 1e-2, 'fp16': 2e-3}[dtype], err
