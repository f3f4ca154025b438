"""GPU: the BASELINE.json configurations and reference options that round 2 left without an ORACLE comparison
("next round" item 1):

  a) configs[4] as written -- ResNet152, fp16, 512x512 -- eval logits against the fp32 torch-CPU oracle within the derived
     16-bit forward bound, and the TTA x4 aggregate against the ORACLE's aggregate of the ORACLE's predictions
     (src/unet_models.py:349-351, src/loaders.py:415-517);
  b) the `encoder: 'AlbuNet'` key (src/models.py:29-31, src/unet_models.py:153-221) through the transformer on the device;
  c) CATEGORY_LAYERS = [1, 19] (src/pipeline_config.py:18: 20 threshold layers) through the batched chain, including the
     zip quirk of build_score (src/postprocessing.py:230: only the first two layers get scores);
  d) a bf16 TRAINING TRAJECTORY: 30 Adam steps of ResNet101 at 256x256 from trained-ish weights, the engine in bf16 against the
     fp32 oracle with torch.optim.Adam -- per-step loss and the final eval masks -- so that a systematic bias of the 16-bit
     gradients (which the per-tensor rel-L2 band of test_gpu_parity_timed.py could hide) shows up as a diverging loss curve.
"""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import losses_ref, post_ref, tta_ref, unet_ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARCH = {'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
        'loss_weights': {'dice_mask': 0.2, 'bce_mask': 1.0}, 'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
UNIT = {'bf16': 2.0 ** -8, 'fp16': 2.0 ** -11}


def record(key, value):
    path = os.path.join(ROOT, 'gpurun_out', 'parity_configs.json')
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


def stored_tensors_eval(depth):
    """16-bit tensors stored one after the other on the longest path of the eval network (BatchNorm folded: one per conv):
    stem + 3 per bottleneck (2 per basic block) + 13 decoder layers (6 x (ConvRelu, ConvTranspose2d) + dec0)"""
    blocks = {34: [3, 4, 6, 3], 101: [3, 4, 23, 3], 152: [3, 8, 36, 3]}[depth]
    return 1 + sum(blocks) * (2 if depth == 34 else 3) + 13


# ------------------------------------------------------------------------------------------------ a) configs[4]
def test_config5_resnet152_fp16_512_eval_and_tta4_against_the_fp32_oracle():
    from mapping_challenge_amd import tta
    ref, net = build(152, 'fp16')
    ref.eval()
    x = unet_ref.synthetic_batch(2, 512, 512, seed=21)
    with torch.no_grad():
        yr = ref(x)
    yh = net.eval()(x.cuda()).cpu()
    d = stored_tensors_eval(152)
    tol = UNIT['fp16'] * math.sqrt(d)                # independent roundings of d stored tensors: u * sqrt(d) (relative L2)
    e = rel_l2(yh, yr)
    assert torch.isfinite(yh).all()
    # TTA x4 (identity, ud, lr, ud: the reference generator with both flips on, src/loaders.py:415-435 + the elif chain :478-481)
    specs = tta.tta_specs(flip_ud=True, flip_lr=True)
    assert specs == tta_ref.tta_specs(flip_ud=True, flip_lr=True) and len(specs) == 4
    got_d = tta.predict_tta(net, x.cuda(), specs, 'gmean').clone()
    assert torch.equal(got_d, tta.predict_tta(net, x.cuda(), specs, 'gmean'))            # deterministic
    plain = net.predict_proba(x.cuda()).clone()
    assert (tta.predict_tta(net, x.cuda(), tta.tta_specs(), 'mean') - plain).abs().max().item() < 1e-6      # identity variant alone
    got = got_d.cpu().numpy()
    xh = x.numpy()
    with torch.no_grad():
        preds = [torch.softmax(ref(torch.from_numpy(np.ascontiguousarray(tta_ref.transform(xh, sp)))), 1).numpy() for sp in specs]
    exp = np.stack([tta_ref.aggregate([p[i] for p in preds], specs, 'gmean') for i in range(x.shape[0])])
    dp = np.abs(got - exp)
    mask_agree = float(((got[:, 1] > 0.5) == (exp[:, 1] > 0.5)).mean())
    record('fp16_r152_512', {'logits_rel_l2': e, 'tol': tol, 'stored_tensors': d, 'tta_prob_maxabs': float(dp.max()),
                             'tta_prob_meanabs': float(dp.mean()), 'tta_mask_agreement': mask_agree})
    assert e < tol, (e, tol)
    assert got.shape == (2, 2, 512, 512)
    # probabilities: |dp| <= |dlogit| / 4 per variant, the geometric mean of four variants does not amplify it
    assert dp.mean() < tol / 4 and dp.max() < 0.05, (dp.mean(), dp.max())
    assert mask_agree > 0.995, mask_agree


# ------------------------------------------------------------------------------------------------ b) AlbuNet key
def test_albunet_encoder_key_on_the_device_matches_the_oracle():
    from mapping_challenge_amd import models as hip_models
    from mapping_challenge_amd.unet_models import AlbuNet
    arch = dict(ARCH, model_params={'encoder': 'AlbuNet', 'compute_dtype': 'fp32'}, optimizer_params={'lr': 5e-4},
                regularizer_params={'regularize': True, 'weight_decay_conv2d': 1e-4})
    t = hip_models.PyTorchUNetWeighted(arch, {'epochs': 1}, {})
    assert isinstance(t.model, AlbuNet) and t.model.encoder_depth == 34
    ref = unet_ref.UNetResNetRef(34)                 # AlbuNet == UNetResNet(34) without the dropout argument (src/unet_models.py:153-221)
    sd = unet_ref.seeded_state_dict(ref)
    ref.load_state_dict(sd)
    t.model.load_state_dict({'module.' + k: v for k, v in sd.items()})
    x = unet_ref.synthetic_batch(3, 128, 128, seed=5)
    ref.eval()
    with torch.no_grad():
        expect = torch.softmax(ref(x), 1).numpy()
    got = t.transform(([[x[:2]], [x[2:]]], 2))['multichannel_map_prediction']
    assert got.shape == (3, 2, 128, 128) and got.dtype == np.float32
    assert np.abs(got - expect).max() < 1e-5
    # one training step through fit() in the exact-fp32 mode tracks the oracle's loss
    tgt = losses_ref.synthetic_target(2, 128, 128, seed=5)
    t.fit(([[x[:2], tgt]], 1))
    ref.train()
    lref = losses_ref.mixed_dice_ce(ref(x[:2]), tgt).item()
    assert abs(t.epoch_losses[0] - lref) < 1e-3 * max(1.0, abs(lref)), (t.epoch_losses, lref)
    # ... and in the timed bf16 mode the same key predicts within the 16-bit forward band
    arch16 = dict(arch, model_params={'encoder': 'AlbuNet', 'compute_dtype': 'bf16'})
    t16 = hip_models.PyTorchUNet(arch16, {'epochs': 1}, {})
    t16.model.load_state_dict(sd)
    got16 = t16.transform(([[x]], 1))['multichannel_map_prediction']
    assert np.abs(got16 - expect).mean() < UNIT['bf16'] * math.sqrt(stored_tensors_eval(34)) / 4


# ------------------------------------------------------------------------------------------------ c) 20 threshold layers
def test_category_layers_1_19_chain_matches_the_oracle_including_the_score_zip_quirk():
    from mapping_challenge_amd import postprocessing as post
    layers_cfg = [1, 19]
    probs = post_ref.synthetic_probs(3, 256, 256, seed=91)
    cls, thr = post.layer_table(layers_cfg)
    assert len(cls) == 20 and list(cls) == [0] + [1] * 19 and thr.dtype == np.float64 and thr[1] == np.arange(0.05, 1, 0.05)[0] and abs(thr[-1] - 0.95) < 1e-12
    out = post.postprocess_batch(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=layers_cfg)
    dev_lab, dev_scores = post.postprocess_device(torch.from_numpy(probs).cuda(), (300, 300), 0, 2, category_layers=layers_cfg)
    assert len(out) == 3 and tuple(dev_lab.shape) == (3, 20, 300, 300)
    for i, (lab, scores) in enumerate(out):
        # the fully independent oracle chain on its own float64 map; thresholds and comparison are float64 on both sides (round 5)
        r = post_ref.resize_image(probs[i], (300, 300))
        assert np.array_equal(post.resize_image(probs[i], (300, 300)), r)
        lay = post.threshold_batch(torch.from_numpy(r[None]).cuda(), layers_cfg)[0].cpu().numpy().astype(bool)
        lay_ref = post_ref.categorize_multilayer_image(r, layers_cfg)
        assert lay.shape == (20, 300, 300) and (lay == lay_ref).all()
        exp = post_ref.dilate_image(post_ref.label_multilayer_image(lay), 2)
        assert lab.dtype == np.int32 and (lab == exp).all()
        assert (dev_lab[i].cpu().numpy() == exp).all()
        # nested thresholds: a higher layer of the building class is a subset of a lower one (before dilation; after the
        # 2x2 max-dilation of label images the SUPPORT is still nested)
        sup = lab[1:] > 0
        assert all((sup[j + 1] <= sup[j]).all() for j in range(18))
        # build_score zips the 20 layers with the 2 probability channels: only layers 0 and 1 are scored (Appendix A.7)
        _, exp_scores = post_ref.build_score(exp, r)
        assert len(scores) == 2 and len(exp_scores) == 2 and len(dev_scores[i]) == 2
        for got_l, exp_l in zip(scores, exp_scores):
            assert len(got_l) == len(exp_l) and np.allclose(got_l, exp_l, rtol=1e-6)
    # per-image functions with the same configuration
    lay = post.threshold_batch(torch.from_numpy(probs[:1]).cuda(), layers_cfg)[0].cpu().numpy().astype(bool)
    assert (lay == post_ref.categorize_multilayer_image(probs[0], layers_cfg)).all()
    _, sc = post.build_score(post_ref.label_multilayer_image(lay), probs[0])
    assert len(sc) == 2


# ------------------------------------------------------------------------------------------------ d) bf16 training trajectory
def test_bf16_training_trajectory_tracks_the_fp32_oracle_resnet101_256():
    """40 bf16 steps from the seeded weights on inputs that carry the target (past the chaotic first steps, eval masks are
    blobs), then 30 MORE steps twice from that state -- the engine continuing in bf16, the fp32 oracle (reference modules +
    losses + torch.optim.Adam with L2) continuing from the same weights, BatchNorm running statistics and Adam moments"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    pre, steps = 40, 30
    tgt = losses_ref.synthetic_target(4, 256, 256, seed=31)
    x = unet_ref.synthetic_batch(4, 256, 256, seed=31) * 0.5 + 2.0 * tgt[:, :1]
    ref, net = build(101, 'bf16')
    net.train()
    opt = HipAdam(net, lr=5e-4, weight_decay=1e-4)
    step = TrainStep(net, LossSpec.mixed(ARCH), opt, use_graph=True)
    warm = [step(x.cuda(), tgt.cuda()).item() for _ in range(pre)]
    assert warm[-1] < 0.5 * warm[0], warm[::8]
    torch.cuda.synchronize()
    # hand the whole training state to the oracle
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in net.state_dict().items()})
    names = [n for n, _ in net._trainable()]
    pw = dict(ref.named_parameters())
    topt = torch.optim.Adam([pw[n] for n in names], lr=5e-4, weight_decay=1e-4)
    for n, m, v in zip(names, net.flat_views(opt.m), net.flat_views(opt.v)):
        topt.state[pw[n]] = {'step': torch.tensor(float(opt.steps)), 'exp_avg': m.detach().cpu().contiguous().clone(),
                             'exp_avg_sq': v.detach().cpu().contiguous().clone()}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    ref.train()
    ref_losses = []
    for _ in range(steps):
        topt.zero_grad()
        loss = losses_ref.mixed_dice_ce(ref(x), tgt)
        loss.backward()
        topt.step()
        ref_losses.append(loss.item())
    hip_losses = [step(x.cuda(), tgt.cuda()).item() for _ in range(steps)]
    rel = [abs(a - b) / abs(b) for a, b in zip(hip_losses, ref_losses)]
    # final weights, eval mode: foreground masks of the two models on the training inputs
    ref.eval()
    with torch.no_grad():
        pr = torch.softmax(ref(x), 1)[:, 1].numpy()
    ph = net.predict_proba(x.cuda())[:, 1].cpu().numpy()
    mr, mh = pr > 0.5, ph > 0.5
    iou = float((mr & mh).sum() / max(1, (mr | mh).sum()))
    # how far apart the two weight sets are after the 30 updates
    final_w = {n: q.detach().cpu().clone() for n, q in net._trainable()}
    record('bf16_r101_256_trajectory', {'warmup_loss': warm[::4], 'oracle_loss': ref_losses, 'engine_loss': hip_losses, 'rel_max': max(rel),
                                         'rel_mean': float(np.mean(rel)), 'final_mask_iou': iou, 'foreground': float(mr.mean()),
                                         'weight_rel_l2_median': float(np.median([rel_l2(final_w[n], pw[n].detach()) for n in names if final_w[n].dim() == 4]))})
    assert np.isfinite(hip_losses).all()
    # the engine's curve follows the oracle's: 1 % on average, no step beyond 5 %, final masks IoU >= 0.985
    # (measured on MI355X: 0.19 % worst step, 0.10 % mean, IoU 0.9989; a 30 % bias of the bf16 gradients -- which the per-tensor band of
    # test_gpu_parity_timed.py would let through -- moves the loss curve by several per cent within ten steps)
    # Round 6: the worst step of a run is a heavy-tailed number -- the weight gradients add their split-K partials with fp32 atomics in arrival order,
    # and thirty Adam steps amplify that noise differently every run: six runs of unchanged code gave 0.19 / 0.48 / 0.51 / 1.1 / 1.8 / 2.7 % worst step,
    # 0.10-0.38 % mean, IoU 0.9936-0.9989.  The MEAN is what a biased gradient moves; the worst step only has to stay out of the several-per-cent range.
    assert max(rel) < 0.05 and np.mean(rel) < 0.01, (max(rel), np.mean(rel), hip_losses[::5], ref_losses[::5])
    assert hip_losses[-1] < 0.6 * hip_losses[0] and ref_losses[-1] < 0.6 * ref_losses[0]      # and both still learn
    assert 0.02 < mr.mean() < 0.9 and iou >= 0.985, (iou, mr.mean())
