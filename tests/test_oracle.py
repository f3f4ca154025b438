"""CPU: the oracle against (a) the reference's own code run through shims (only where /root/reference
exists), (b) the committed golden vectors (everywhere), (c) the one known-answer test the reference
holds (label_multiclass_image docstring, src/postprocessing.py:96-111)."""
import os
from functools import partial

import numpy as np
import pytest
import torch

from oracle import ref_import, unet_ref, post_ref, losses_ref, crf_ref

needs_ref = pytest.mark.skipif(not ref_import.available(), reason='/root/reference not present')

KAT_IN = np.array([[0, 0, 1, 1], [1, 0, 0, 0], [1, 1, 1, 0], [0, 0, 1, 0]])
KAT_OUT = np.array([[[1, 1, 0, 0], [0, 1, 1, 1], [0, 0, 0, 1], [2, 2, 0, 1]],
                    [[0, 0, 1, 1], [2, 0, 0, 0], [2, 2, 2, 0], [0, 0, 2, 0]]])


def test_label_docstring_known_answer():
    assert (post_ref.label_multiclass_image(KAT_IN) == KAT_OUT).all()
    for c in range(2):
        assert (post_ref.label_unionfind(KAT_IN == c) == KAT_OUT[c]).all()


def test_unionfind_matches_scipy_on_adversarial_masks():
    rng = np.random.default_rng(0)
    masks = [rng.random((23, 31)) > t for t in (0.3, 0.5, 0.7)]
    masks += [np.indices((16, 16)).sum(0) % 2 == 0, np.ones((9, 9), bool), np.zeros((9, 9), bool)]
    for m in masks:
        assert (post_ref.label_unionfind(m) == post_ref.label(m)).all()


def test_resize_oracle_is_scipy_map_coordinates_bit_for_bit():
    """the reference's resize is skimage's n-d branch = scipy.ndimage.map_coordinates(order 1, 'constant') + clip, and its float64 output
    is what gets thresholded: the restatement the GPU tests compare with (and the kernel follows) must equal the INSTALLED scipy --
    the reference's real dependency -- in every bit, also on maps made of values a float32 ulp around the 0.5 threshold"""
    from scipy import ndimage as ndi
    rng = np.random.default_rng(3)
    cases = [((2, 64, 64), (75, 75)), ((2, 256, 256), (300, 300)), ((3, 96, 80), (112, 100)), ((2, 64, 64), (48, 48)), ((2, 5, 7), (300, 251))]
    cases += [((2, int(rng.integers(5, 120)), int(rng.integers(5, 120))), (int(rng.integers(5, 200)), int(rng.integers(5, 200)))) for _ in range(12)]
    vals = np.array([0.5 - 2.0 ** -25, 0.5, 0.5 + 2.0 ** -24, 1.0, 0.0], np.float32)
    for shape, (th, tw) in cases:
        for img in (rng.random(shape).astype(np.float32), vals[rng.integers(0, 5, shape)]):
            factors = [1.0, shape[1] / th, shape[2] / tw]
            coords = [factors[i] * (np.arange(d) + 0.5) - 0.5 for i, d in enumerate((shape[0], th, tw))]
            cm = np.array(np.meshgrid(*coords, sparse=False, indexing='ij'))
            img64 = img.astype(np.float64)
            exp = ndi.map_coordinates(img64, cm, order=1, mode='constant', cval=0)
            exp = np.clip(exp, min(img64.min(), 0), max(img64.max(), 0))
            got = post_ref.resize_image(img, (th, tw))
            assert got.dtype == np.float64 and np.array_equal(got, exp), (shape, th, tw)


def test_post_chain_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'post.npz'))
    assert (post_ref.label_multiclass_image(KAT_IN) == g['kat_multiclass']).all()
    probs = post_ref.synthetic_probs(3, 64, 64, seed=1234, smooth=2.0)
    for i, p in enumerate(probs):
        r = post_ref.resize_image(p, (75, 75))
        assert np.allclose(r, g['resized_%d' % i], atol=1e-6)
        lay = post_ref.categorize_multilayer_image(r)
        assert (lay == g['layers_%d' % i]).all()
        lab = post_ref.label_multilayer_image(lay)
        assert (lab == g['labels_%d' % i]).all() and lab.dtype == np.int32
        assert (post_ref.dilate_image(lab, 2) == g['dilated2_%d' % i]).all()
        assert (post_ref.dilate_image(lab, 3) == g['dilated3_%d' % i]).all()
        assert (post_ref.erode_image(lay[1], 3) == g['eroded3_%d' % i]).all()
        _, sc = post_ref.build_score(post_ref.dilate_image(lab, 2), r)
        assert np.allclose(sc[0], g['scores0_%d' % i], rtol=1e-12) and np.allclose(sc[1], g['scores1_%d' % i], rtol=1e-12)


def test_unet_oracle_matches_golden(golden_dir):
    for depth in (34, 101):
        g = np.load(os.path.join(golden_dir, 'unet_r%d_64.npz' % depth))
        n = g['logits_eval'].shape[0]
        net = unet_ref.UNetResNetRef(depth)
        net.load_state_dict(unet_ref.seeded_state_dict(net))
        x = unet_ref.synthetic_batch(n, 64, 64)
        net.eval()
        with torch.no_grad():
            assert np.allclose(net(x).numpy(), g['logits_eval'], atol=2e-5)
        net.train()
        out = net(x)
        loss = losses_ref.mixed_dice_ce(out, losses_ref.synthetic_target(n, 64, 64))
        loss.backward()
        assert abs(loss.item() - float(g['loss'])) < 1e-5
        assert np.allclose(net.final.weight.grad.numpy(), g['g_final_w'], rtol=1e-4, atol=1e-6)
        assert np.allclose(net.encoder.conv1.weight.grad.numpy()[:8], g['g_conv1'], rtol=1e-3, atol=1e-6)


def test_unet_oracle_matches_reference_digest_of_the_timed_network(golden_dir):
    """ResNet101 at 128x128, batch 4 (what the GPU gradient-parity tests of the timed network compare with): the oracle's
    loss, logits and every parameter gradient against the digest the reference's own backward produced"""
    g = np.load(os.path.join(golden_dir, 'unet_r101_128.npz'))
    net = unet_ref.UNetResNetRef(101)
    net.load_state_dict(unet_ref.seeded_state_dict(net))
    net.train()
    out = net(unet_ref.synthetic_batch(4, 128, 128, seed=12))
    loss = losses_ref.mixed_dice_ce(out, losses_ref.synthetic_target(4, 128, 128, seed=12))
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) < 1e-5 and np.allclose(out.detach().numpy(), g['logits_train'], atol=2e-5)
    checked = 0
    for name, p in net.named_parameters():
        if 'n|' + name not in g.files:
            continue
        flat = p.grad.reshape(-1).double()
        norm = float(g['n|' + name])
        assert abs(flat.norm().item() - norm) <= 1e-4 * norm + 1e-12, name
        head = torch.from_numpy(g['h|' + name]).double()
        assert (flat[:head.numel()] - head).abs().max().item() <= 1e-4 * norm / np.sqrt(flat.numel()) * 8 + 1e-12, name
        checked += 1
    assert checked == len([k for k in g.files if k.startswith('n|')]) and checked > 300


def _corner_cases():
    """masks where erosion leaves (a) only pixel (0,0) of a component, (b) nothing of one, (c) an interior pixel"""
    orig = np.zeros((9, 11), np.uint8)
    orig[0:2, 0:2] = 1           # component touching the corner
    orig[4:7, 4:7] = 1           # survives at its centre
    orig[8, 9:11] = 1            # wiped out
    proc = np.zeros_like(orig)
    proc[0, 0] = 1
    proc[5, 5] = 1
    return orig, proc


@needs_ref
def test_add_dropped_objects_reproduces_the_reference_index_quirk():
    """src/utils.py:337 evaluates np.any(np.where(overlap)): a component whose only surviving pixel is (0,0) is re-added"""
    ref_utils = ref_import.ref('utils')
    orig, proc = _corner_cases()
    for o, p in ((orig, proc), (orig.astype(bool), proc.astype(bool)), (orig[::-1].copy(), proc[::-1].copy())):
        exp = ref_utils.add_dropped_objects(o, p)
        got = post_ref.add_dropped_objects(o, p)
        assert exp.dtype == got.dtype == np.uint8 and (exp == got).all()
    exp = ref_utils.add_dropped_objects(orig, proc)
    assert exp[0, 0] == 2 and exp[1, 1] == 1 and exp[8, 10] == 1 and exp[4, 4] == 0      # integer masks: the corner pixel is summed
    assert ref_utils.add_dropped_objects(orig.astype(bool), proc.astype(bool))[0, 0] == 1     # bool masks: logical or


def test_loss_oracle_matches_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, 'loss.npz'))
    rng = np.random.default_rng(1234)
    logits = torch.from_numpy(rng.standard_normal((2, 2, 64, 64)).astype(np.float32) * 3).requires_grad_(True)
    tgt = losses_ref.synthetic_target(2, 64, 64, seed=7)
    loss = losses_ref.segmentation_ce(logits, tgt[:, :1])
    loss.backward()
    assert abs(loss.item() - float(g['loss_ce'])) < 1e-6 and np.allclose(logits.grad.numpy(), g['dlogits_ce'], atol=1e-8)
    logits.grad = None
    loss = losses_ref.mixed_dice_ce(logits, tgt)
    loss.backward()
    assert abs(loss.item() - float(g['loss_mixed'])) < 1e-5 and np.allclose(logits.grad.numpy(), g['dlogits_mixed'], atol=1e-7)
    # dice_activation = 'sigmoid' (src/models.py:440-441), generated by the reference's multiclass_dice_loss like the rest
    logits.grad = None
    loss = losses_ref.mixed_dice_ce(logits, tgt, dice_activation='sigmoid')
    loss.backward()
    assert abs(loss.item() - float(g['loss_mixed_sigmoid'])) < 1e-5 and np.allclose(logits.grad.numpy(), g['dlogits_mixed_sigmoid'], atol=1e-8, rtol=1e-5)
    assert np.abs(g['dlogits_mixed_sigmoid'] - g['dlogits_mixed']).max() > 1e-6       # the two activations are told apart at this tolerance
    with pytest.raises(NotImplementedError):
        losses_ref.mixed_dice_ce(logits, tgt, dice_activation='tanh')


def test_adam_oracle_matches_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(1000)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=5e-4, weight_decay=1e-4)
    p, m, v = p0.clone(), torch.zeros(1000), torch.zeros(1000)
    for step in range(1, 4):
        g = torch.randn(1000)
        p_ref.grad = g.clone()
        opt.step()
        losses_ref.adam_l2_step(p, g, m, v, step)
        assert torch.allclose(p, p_ref.data, atol=1e-7)


def test_crf_oracle_properties():
    rng = np.random.default_rng(0)
    probs = post_ref.synthetic_probs(1, 24, 24, seed=3, smooth=2.0)[0]
    img = rng.standard_normal((3, 24, 24)).astype(np.float32)
    q = crf_ref.dense_crf(img, probs, post_ref_mean(), post_ref_std(), iterations=2)
    assert q.shape == probs.shape and np.allclose(q.sum(0), 1, atol=1e-5) and (q >= 0).all()
    q0 = crf_ref.dense_crf(img, probs, post_ref_mean(), post_ref_std(), iterations=0)
    assert np.allclose(q0, np.clip(probs, 1e-5, 1) / np.clip(probs, 1e-5, 1).sum(0), atol=1e-5)


def post_ref_mean():
    return [0.485, 0.456, 0.406]


def post_ref_std():
    return [0.229, 0.224, 0.225]


# ---------------------------------------------------------------- against the literal reference code
@needs_ref
def test_oracle_unet_equals_reference_class():
    um = ref_import.ref('unet_models')
    for depth in (34, 101):
        a = um.UNetResNet(depth, 2, num_filters=32, dropout_2d=0.0, pretrained=True, is_deconv=True)
        b = unet_ref.UNetResNetRef(depth)
        assert set(a.state_dict()) == set(b.state_dict())
        sd = unet_ref.seeded_state_dict(a)
        a.load_state_dict(sd)
        b.load_state_dict(sd)
        x = unet_ref.synthetic_batch(1, 64, 64)
        for mode in ('eval', 'train'):
            getattr(a, mode)()
            getattr(b, mode)()
            with torch.no_grad():
                assert torch.equal(a(x), b(x))


@needs_ref
def test_oracle_post_equals_reference_functions():
    pp = ref_import.ref('postprocessing')
    ru = ref_import.ref('utils')
    probs = post_ref.synthetic_probs(2, 96, 96, seed=5, smooth=3.0)
    for p in probs:
        a, b = pp.resize_image(p, (112, 112)), post_ref.resize_image(p, (112, 112))
        assert a.dtype == np.float64 and np.array_equal(a, b)       # bit for bit: the reference thresholds this map (ties at 0.5)
        la, lb = pp.categorize_multilayer_image(a), post_ref.categorize_multilayer_image(b)
        assert (la == lb).all()
        lab = pp.label_multilayer_image(la)
        assert (lab == post_ref.label_multilayer_image(lb)).all()
        for k in (1, 2, 3, 4, 5):
            assert (pp.dilate_image(lab, k) == post_ref.dilate_image(lab, k)).all()
            assert (pp.erode_image(la[1], k) == post_ref.erode_image(lb[1], k)).all()
        sa, sb = pp.build_score(lab, a), post_ref.build_score(lab, b)
        assert all(np.allclose(x, y, rtol=1e-12) for x, y in zip(sa[1], sb[1]))
        assert (pp.crop_image_center_per_class(a, 100, 100) == post_ref.crop_image_center_per_class(a, 100, 100)).all()
        assert np.allclose(ru.softmax(p, axis=0), post_ref.softmax(p, axis=0))
        assert (pp.categorize_image(p) == post_ref.categorize_image(p)).all()


@needs_ref
def test_oracle_losses_equal_reference_functions():
    rm = ref_import.ref('models')
    val = ref_import.ref('steps.pytorch.validation')
    torch.manual_seed(0)
    out = torch.randn(2, 2, 48, 48) * 2
    t = losses_ref.synthetic_target(2, 48, 48, seed=3)
    wf = partial(rm.get_weights, w0=50, sigma=10, imsize=(256, 256))
    assert torch.allclose(rm.multiclass_weighted_cross_entropy(out, t, weights_function=wf),
                          losses_ref.weighted_ce(out, t, 50, 10, (256, 256)), rtol=1e-6)
    mix = rm.mixed_dice_cross_entropy_loss(out, t, dice_weight=0.2, cross_entropy_weight=1.0,
                                           dice_loss=partial(rm.multiclass_dice_loss, excluded_classes=[0]),
                                           cross_entropy_loss=partial(rm.multiclass_weighted_cross_entropy, weights_function=wf),
                                           smooth=1, dice_activation='softmax')
    assert torch.allclose(mix, losses_ref.mixed_dice_ce(out, t), rtol=1e-6)
    mix_s = rm.mixed_dice_cross_entropy_loss(out, t, dice_weight=0.2, cross_entropy_weight=1.0,
                                             dice_loss=partial(rm.multiclass_dice_loss, excluded_classes=[0]),
                                             cross_entropy_loss=partial(rm.multiclass_weighted_cross_entropy, weights_function=wf),
                                             smooth=1, dice_activation='sigmoid')
    assert torch.allclose(mix_s, losses_ref.mixed_dice_ce(out, t, dice_activation='sigmoid'), rtol=1e-6) and not torch.equal(mix, mix_s)
    assert torch.allclose(val.multiclass_segmentation_loss(out, t[:, :1]), losses_ref.segmentation_ce(out, t[:, :1]))


@needs_ref
def test_reference_dense_crf_runs_on_oracle_shim():
    pp = ref_import.ref('postprocessing')
    rng = np.random.default_rng(1)
    probs = post_ref.synthetic_probs(1, 20, 20, seed=2, smooth=2.0)[0]
    img = rng.standard_normal((3, 20, 20)).astype(np.float32)
    a = pp.dense_crf(img, probs, iterations=3)
    b = crf_ref.dense_crf(img, probs, post_ref_mean(), post_ref_std(), iterations=3)
    assert np.allclose(a, b, atol=1e-6)


def test_c_restatement_of_post_chain_matches_numpy_oracle():
    """oracle/post_ref.c (the C port timed as CPU baseline) against oracle/post_ref.py, incl. empty and full masks"""
    from oracle import post_ref_c
    probs = list(post_ref.synthetic_probs(3, 96, 80, seed=11, smooth=3.0))
    probs.append(np.stack([np.ones((96, 80), np.float32), np.zeros((96, 80), np.float32)]))
    for p in probs:
        for k in (0, 2, 3):
            lab, sc = post_ref_c.postprocess(p, (112, 100), k)
            lab2, sc2 = post_ref.postprocess(p, (112, 100), 0, k)
            assert (lab == lab2).all()
            assert all(np.allclose(a, b, rtol=1e-12) for a, b in zip(sc, sc2))


@needs_ref
def test_oracle_tta_equals_reference_functions():
    from oracle import tta_ref
    ld = ref_import.ref('loaders')
    rng = np.random.default_rng(0)
    gen = ld.TestTimeAugmentationGenerator(flip_ud=True, flip_lr=True, rotation=True, color_shift_runs=False)
    _, params, _ = gen._get_tta_data(0, {'x': 1})
    specs = tta_ref.tta_specs(True, True, True)
    assert params == specs
    img = rng.random((32, 32, 3)).astype(np.float32)
    preds = [rng.random((2, 32, 32)).astype(np.float32) for _ in specs]
    for sp, pr in zip(specs, preds):
        assert np.array_equal(ld.test_time_augmentation_transform(img, sp), tta_ref.transform(img.transpose(2, 0, 1), sp).transpose(1, 2, 0))
        assert np.array_equal(ld.test_time_augmentation_inverse_transform(pr, sp), tta_ref.inverse_transform(pr, sp))
    for m in ('mean', 'gmean', 'max', 'min'):
        agg = ld.TestTimeAugmentationAggregator(m, 1)
        a = ld.aggregate_augmentations(0, preds, specs, [0] * len(specs), agg.agg_method)
        assert np.allclose(a, tta_ref.aggregate(preds, specs, m), rtol=1e-6)
