"""GPU: parity of the configuration bench.py times (ResNet101-U-Net, 256x256 network input) against the torch-CPU oracle
-- forward logits, loss, dlogits and EVERY parameter gradient, in the exact-fp32 mode (north-star bound 1e-4 on the logits)
and in the 16-bit throughput modes (bf16: the timed one; fp16: BASELINE.json configs[4]) -- plus the agreement of the
post-processed instance masks between the 16-bit and the fp32 path.

Forward tolerances of the 16-bit modes are derived: every activation tensor is stored once in the 16-bit type (unit
roundoff u = 2^-8 for bf16, 2^-11 for fp16; accumulation, BatchNorm statistics and the loss are fp32), the roundings of
successive layers are independent, so the relative L2 error of the logits, d stored tensors downstream of the input, grows
like u * sqrt(d).  ResNet101-U-Net: 113 convolution layers on the longest path, two stored tensors each in training (raw
conv output, BN+ReLU output)  =>  d = 226; bound = K * u * sqrt(d) with K = 1.

Gradients cannot be bounded that way: the gradient of a deep ReLU network is DISCONTINUOUS in the activations (a
pre-activation crossing zero flips a mask), so the distance of any finite-precision backward pass to the exact one is set
by the storage precision, not by the kernels.  Measured on the CPU (oracle/lowp_ref.py, ResNet101 128x128 batch 4, relative
L2 per parameter tensor against a float64 evaluation): the reference's own fp32 path 5.2e-3 median / 1.1e-2 worst; the same
arithmetic with bf16 storage of activations and activation gradients 0.59 / 0.75; with fp16 storage 0.23 / 0.30 -- and two
evaluations that round the same tensors but accumulate in fp32 vs fp64 are 0.5 apart from each other.  The ground truth here
is therefore the float64 oracle, and the bound on the engine's error is the error of the REFERENCE ARITHMETIC AT THE SAME
STORAGE PRECISION (oracle/lowp_ref.py) against that truth: the engine must be as close to the exact gradient as the
reference would be if it stored what the engine stores (factor 1.2 on the median / 90th percentile / maximum over tensors).
The measured numbers are written to gpurun_out/parity_timed.json.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref, lowp_ref, post_ref, unet_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
        'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
UNIT = {'bf16': 2.0 ** -8, 'fp16': 2.0 ** -11}
D_FWD, K = 226, 1.0


def record(key, value):
    path = os.path.join(ROOT, 'gpurun_out', 'parity_timed.json')
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[key] = value
    json.dump(data, open(path, 'w'), indent=1, sort_keys=True)


def build(depth, dtype, sd=None):
    from mapping_challenge_amd.unet_models import UNetResNet
    ref = unet_ref.UNetResNetRef(depth)
    sd = sd or unet_ref.seeded_state_dict(ref)
    ref.load_state_dict(sd)
    net = UNetResNet(depth, 2, num_filters=32, dropout_2d=0.0, pretrained=False, is_deconv=True, compute_dtype=dtype)
    net.load_state_dict(sd)
    net.flatten_parameters('cuda')
    return ref, net


def rel_l2(a, b):
    return (a.double() - b.double()).norm().item() / (b.double().norm().item() + 1e-30)


def oracle_step(depth, x, tgt, base=torch.float32, storage=None):
    """oracle in `base` arithmetic (float64 = ground truth), optionally at 16-bit storage: train-mode forward, mixed loss,
    backward -> logits, loss, dlogits, {name: grad} (all as float64)"""
    ref = unet_ref.UNetResNetRef(depth)
    ref.load_state_dict(unet_ref.seeded_state_dict(ref))
    ref = ref.to(base)
    if storage:
        lowp_ref.install(ref, storage)
    ref.train()
    out = ref(x.to(base))
    out.retain_grad()
    loss = losses_ref.mixed_dice_ce(out, tgt.to(base))
    loss.backward()
    grads = {n: p.grad.double().clone() for n, p in ref.named_parameters() if p.grad is not None}
    return out.detach().double(), loss.item(), out.grad.double().clone(), grads


def stats(errs):
    v = np.array(list(errs.values()))
    worst = max(errs.items(), key=lambda kv: kv[1])
    return {'median': float(np.median(v)), 'p90': float(np.quantile(v, 0.9)), 'max': float(v.max()), 'argmax': worst[0]}


def hip_step(net, x, tgt, loss_scale=1.0):
    from mapping_challenge_amd.trainer import LossSpec, loss_forward_backward
    net.train()
    prog = net.train_forward(x.cuda())
    loss = torch.zeros(1, device='cuda')
    sums = torch.zeros(4, dtype=torch.float64, device='cuda')
    loss_forward_backward(prog.logits, tgt.cuda(), LossSpec.mixed(ARCH), prog.dlogits, loss, sums, None, loss_scale)
    dlogits = prog.dlogits.clone()
    net.train_backward(prog)
    grads = {n: g.cpu() / loss_scale for (n, _), g in zip(net._trainable(), net._grad_views())}
    return prog.logits.cpu(), loss.item(), dlogits.cpu() / loss_scale, grads


def test_fp32_mode_resnet101_256_logits_within_1e4_and_gradients_as_exact_as_the_reference_path():
    """the exact-fp32 mode at the timed resolution: eval logits at 256x256 N=2 within the north star's 1e-4 of the fp32
    oracle; train step at 128x128 N=4: logits / loss / dlogits against the fp32 oracle, and every gradient tensor as close
    to the float64 gradient as the fp32 oracle (= the reference's own path) is"""
    ref, net = build(101, 'fp32')
    x = unet_ref.synthetic_batch(2, 256, 256, seed=11)
    ref.eval(); net.eval()
    with torch.no_grad():
        yr = ref(x)
        yh = net(x.cuda()).cpu()
    err = (yr - yh).abs().max().item()
    record('fp32_r101_256_eval_logits_maxabs', err)
    assert err < 1e-4
    x = unet_ref.synthetic_batch(4, 128, 128, seed=12)
    tgt = losses_ref.synthetic_target(4, 128, 128, seed=12)
    to, tl, td, tg = oracle_step(101, x, tgt, torch.float64)           # ground truth
    lo, ll, ld, lg = oracle_step(101, x, tgt, torch.float32)           # the reference's own arithmetic
    ho, hl, hd, hg = hip_step(net, x, tgt)
    assert (lo - ho.double()).abs().max().item() < 2e-4 and abs(ll - hl) < 1e-4 * max(1.0, abs(ll))
    assert rel_l2(hd, ld) < 1e-4
    e_hip = stats({n: rel_l2(hg[n], tg[n]) for n in hg if n in tg})
    e_ref = stats({n: rel_l2(lg[n], tg[n]) for n in hg if n in lg})
    record('fp32_r101_128_grad_rel_l2_vs_fp64', {'engine': e_hip, 'fp32_oracle': e_ref})
    for q in ('median', 'p90', 'max'):
        assert e_hip[q] <= 1.5 * e_ref[q] + 1e-4, (q, e_hip, e_ref)
    assert e_hip['max'] < 3e-2


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_16bit_train_step_resnet101_256_vs_oracle(dtype):
    """the timed configuration (ResNet101, 256x256, train step) at batch 4: forward against the fp32 oracle within the derived
    bound; gradients against the float64 oracle, as close as the reference arithmetic at the same storage precision"""
    _, net = build(101, dtype)
    x = unet_ref.synthetic_batch(4, 256, 256, seed=21)
    tgt = losses_ref.synthetic_target(4, 256, 256, seed=21)
    scale = 4096.0 if dtype == 'fp16' else 1.0            # trainer.TrainStep's static loss scale for fp16
    to, tl, td, tg = oracle_step(101, x, tgt, torch.float64)
    eo, el, ed, eg = oracle_step(101, x, tgt, torch.float64, dtype)    # reference arithmetic, 16-bit storage
    ho, hl, hd, hg = hip_step(net, x, tgt, scale)
    u = UNIT[dtype]
    tol_fwd = K * u * math.sqrt(D_FWD)
    e_logits, e_dlogits = rel_l2(ho, to), rel_l2(hd, td)
    e_hip = stats({n: rel_l2(hg[n], tg[n]) for n in hg if n in tg})
    e_emu = stats({n: rel_l2(eg[n], tg[n]) for n in hg if n in eg})
    cos = stats({n: 1.0 - float((hg[n].double().flatten() @ tg[n].flatten()) / (hg[n].double().norm() * tg[n].norm() + 1e-30)) for n in hg if n in tg})
    record('%s_r101_256_train' % dtype, {'logits_rel_l2': e_logits, 'logits_rel_l2_same_storage_oracle': rel_l2(eo, to), 'dlogits_rel_l2': e_dlogits,
                                          'loss': [hl, tl, el], 'tol_fwd': tol_fwd, 'grad_rel_l2_vs_fp64': {'engine': e_hip, 'same_storage_oracle': e_emu},
                                          'one_minus_cosine_vs_fp64': cos})
    assert torch.isfinite(ho).all() and all(torch.isfinite(g).all() for g in hg.values())
    assert e_logits < tol_fwd, (e_logits, tol_fwd)
    assert abs(hl - tl) < tol_fwd * max(1.0, abs(tl)), (hl, tl)
    assert e_dlogits < tol_fwd, (e_dlogits, tol_fwd)
    assert len(hg) > 300
    for q in ('median', 'p90', 'max'):
        assert e_hip[q] <= 1.2 * e_emu[q] + 0.02, (q, e_hip, e_emu)


def test_batch32_timed_shape_resnet101_256_vs_oracle():
    """the EXACT shape bench.py times -- ResNet101, 256x256, batch 32 (the tile configurations the tuner picks depend on the batch
    size: tune/gfx950.json is keyed by N) -- against the fp32 oracle (src/steps/pytorch/models.py:76-113: forward, loss, backward):
    fp32 mode: eval logits within the north star's 1e-4;  bf16 mode: eval logits within u*sqrt(d), and one training step's
    train-mode logits, loss and dlogits within the same band; every parameter gradient as close to the fp32 oracle's as the
    reference arithmetic with bf16 storage is (the band of test_16bit_train_step_resnet101_256_vs_oracle, at batch 32)"""
    N = 32
    x = unet_ref.synthetic_batch(N, 256, 256, seed=41)
    tgt = losses_ref.synthetic_target(4, 256, 256, seed=41).repeat(N // 4, 1, 1, 1)
    ref, fp = build(101, 'fp32')
    ref.eval(); fp.eval()
    with torch.no_grad():
        yr = ref(x)
    yf = fp(x.cuda()).cpu()
    err32 = (yr - yf).abs().max().item()
    del fp
    _, net = build(101, 'bf16')
    net.eval()
    yb = net(x.cuda()).cpu()
    u = UNIT['bf16']
    tol = K * u * math.sqrt(D_FWD)
    e_eval = rel_l2(yb, yr)
    # training step: fp32 oracle as the truth (its own distance to the float64 gradient, 5e-3, is far inside the 16-bit band)
    to, tl, td, tg = oracle_step(101, x, tgt, torch.float32)
    eo, el, ed, eg = oracle_step(101, x, tgt, torch.float32, 'bf16')
    ho, hl, hd, hg = hip_step(net, x, tgt)
    e_logits, e_dlogits = rel_l2(ho, to), rel_l2(hd, td)
    e_hip = stats({n: rel_l2(hg[n], tg[n]) for n in hg if n in tg})
    e_emu = stats({n: rel_l2(eg[n], tg[n]) for n in hg if n in eg})
    record('batch32_r101_256', {'fp32_eval_logits_maxabs': err32, 'bf16_eval_logits_rel_l2': e_eval, 'bf16_train_logits_rel_l2': e_logits,
                                'bf16_dlogits_rel_l2': e_dlogits, 'loss': [hl, tl, el], 'tol_fwd': tol,
                                'grad_rel_l2_vs_fp32_oracle': {'engine': e_hip, 'same_storage_oracle': e_emu}})
    assert err32 < 1e-4, err32
    assert e_eval < tol, (e_eval, tol)
    assert torch.isfinite(ho).all() and all(torch.isfinite(g).all() for g in hg.values())
    assert e_logits < tol and e_dlogits < tol, (e_logits, e_dlogits, tol)
    assert abs(hl - tl) < tol * max(1.0, abs(tl)), (hl, tl)
    assert len(hg) > 300
    for q in ('median', 'p90', 'max'):
        assert e_hip[q] <= 1.2 * e_emu[q] + 0.02, (q, e_hip, e_emu)


def _trained_state(depth=101, steps=40):
    """a few dozen optimizer steps on inputs that carry the target (noise + mask), so that the eval masks are building-like
    blobs instead of the near-0.5 noise of random weights; returns (state_dict on the host, inputs)"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    _, net = build(depth, 'bf16')
    tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
    x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
    net.train()
    step = TrainStep(net, LossSpec.mixed(ARCH), HipAdam(net, lr=5e-4, weight_decay=1e-4))
    losses = [step(x.cuda(), tgt.cuda()).item() for _ in range(steps)]
    assert losses[-1] < 0.5 * losses[0], losses[::8]
    return {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}, x, tgt


def _instances_iou(la, lb, min_area=16, with_diff=False):
    """for every instance of label image la with >= min_area pixels: IoU with its best-overlapping instance of lb (with_diff: also the
    number of pixels the two differ in)"""
    out = []
    for i in range(1, int(la.max()) + 1):
        ma = la == i
        if ma.sum() < min_area:
            continue
        cand = np.unique(lb[ma])
        cand = cand[cand > 0]
        best, diff = 0.0, int(ma.sum())
        for j in cand:
            mb = lb == j
            iou = (ma & mb).sum() / float((ma | mb).sum())
            if iou > best:
                best, diff = iou, int((ma ^ mb).sum())
        out.append((best, diff) if with_diff else best)
    return out


@pytest.mark.parametrize('dtype', ['bf16', 'fp16'])
def test_16bit_masks_agree_with_fp32_path_after_postprocessing(dtype):
    """eval forward + the shipped post-processing chain (resize 256->300, threshold, label, 2x2 dilation) in the 16-bit mode
    and in the fp32 mode (itself held to 1e-4 of the oracle here) on trained weights: the INSTANCES must agree"""
    from mapping_challenge_amd import postprocessing as post
    sd, x, tgt = _trained_state()
    ref, fp = build(101, 'fp32', sd)
    _, lo = build(101, dtype, sd)
    ref.eval()
    with torch.no_grad():
        yr = ref(x[:2])
    pf = fp.predict_proba(x.cuda()).clone()
    err = (torch.softmax(yr, 1) - pf[:2].cpu()).abs().max().item()
    assert err < 1e-4, err                                   # the fp32 path is the oracle's, also on trained weights at 256x256
    pl = lo.predict_proba(x.cuda()).clone()
    dp = (pf - pl).abs()
    lab_f = post.postprocess_batch(pf, (300, 300), 0, 2)
    lab_l = post.postprocess_batch(pl, (300, 300), 0, 2)
    agree, fg_iou, inst_d = [], [], []
    for (a, _), (b, _) in zip(lab_f, lab_l):
        ma, mb = a[1] > 0, b[1] > 0
        agree.append((ma == mb).mean())
        fg_iou.append((ma & mb).sum() / max(1.0, float((ma | mb).sum())))
        inst_d += _instances_iou(a[1], b[1], with_diff=True)
    inst = [i for i, _ in inst_d]
    frac_fg = float(np.mean([(a[1] > 0).mean() for a, _ in lab_f]))
    stats = {'prob_maxabs': dp.max().item(), 'prob_mean_abs': dp.mean().item(), 'pixel_agreement': float(np.mean(agree)),
             'foreground_iou': float(np.mean(fg_iou)), 'instances': len(inst), 'instance_iou_mean': float(np.mean(inst)) if inst else None,
             'instance_iou_min': float(np.min(inst)) if inst else None, 'foreground_fraction': frac_fg, 'fp32_vs_oracle_prob_maxabs': err}
    record('%s_r101_256_masks' % dtype, stats)
    assert 0.02 < frac_fg < 0.9 and len(inst) >= 4, stats      # the trained net draws blobs, not an empty / full mask
    u = UNIT[dtype]
    assert dp.mean().item() < K * u * math.sqrt(D_FWD) / 4, stats      # probabilities: softmax slope <= 1/4
    # measured (gpurun_out/parity_timed.json): bf16 pixel agreement 0.9998, foreground IoU 0.9993, instance IoU mean 0.9993 / min 0.986
    assert np.mean(agree) > 0.999 and np.mean(fg_iou) > 0.995, stats
    # per instance: IoU > 0.9 -- or, for the smallest ones (16-40 pixels, where every boundary pixel is 3-6 % of the area), at most three
    # pixels of difference: a 17-pixel instance that differs in two boundary pixels has IoU 0.88 and is the same building
    worst = min(inst_d, key=lambda t: t[0])
    record('%s_r101_256_masks_worst_instance' % dtype, {'iou': worst[0], 'pixels_differing': worst[1]})
    # (round 6) the labelling is not continuous in the mask: one pixel of a one-pixel bridge between two buildings, on one side of 0.5 in one mode and on the
    # other in the other, merges two instances into one -- IoU 0.33 for a 2820-pixel instance while the masks agree in 99.98 % of the pixels (seen once in
    # ~a dozen runs of the suite; the weights are trained with atomics in arrival order, so every run tests a slightly different network).  Such a topology flip
    # is allowed for at most one instance in a hundred; everything else keeps the per-instance bound, and the masks themselves are held above
    flips = [(i, d) for i, d in inst_d if not (i > 0.9 or d <= 3)]
    record('%s_r101_256_masks_topology_flips' % dtype, {'count': len(flips), 'instances': len(inst_d)})
    assert np.mean(inst) > 0.99 and len(flips) <= max(1, len(inst_d) // 100), (stats, worst, flips)
