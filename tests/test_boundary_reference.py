"""CPU: the drop-in boundary.  The HIP transformer classes are driven by the REFERENCE's own `Step`
(src/steps/base.py) and the REFERENCE's own `CallbackList` built by `callbacks_unet` (src/models.py:295-307), both imported
unmodified through oracle/ref_import.py; the compute back end is tests/emu.py (the numpy interpreter of the C-ABI
semantics), so what is tested is the host-side contract: callback protocol, callable loss_function entries, optimizer that
torch's ExponentialLR accepts, validation through score_model, early stopping, best-checkpoint format, Step caching.
The same loop runs on the GPU in tests/test_gpu_pipeline.py."""
import logging
import os

import numpy as np
import pytest
import torch

import emu
from mapping_challenge_amd import models as hip_models
from mapping_challenge_amd import unet_models as um
from oracle import losses_ref, ref_import, unet_ref

needs_ref = pytest.mark.skipif(not ref_import.available(), reason='/root/reference not present')


@pytest.fixture()
def interpreted(monkeypatch):
    emu.install(monkeypatch.setattr, models=True)


def configs(tmp_path, encoder='ResNet34', epochs=3, patience=0):
    import pathlib
    tmp_path = pathlib.Path(str(tmp_path))
    arch = {'model_params': {'encoder': encoder, 'compute_dtype': 'fp32'},
            'optimizer_params': {'lr': 5e-4}, 'regularizer_params': {'regularize': True, 'weight_decay_conv2d': 1e-4},
            'weights_init': {'function': 'he'}, 'loss_weights': {'bce_mask': 1.0, 'dice_mask': 0.2},
            'weighted_cross_entropy': {'w0': 50, 'sigma': 10, 'imsize': (256, 256)},
            'dice': {'smooth': 1, 'dice_activation': 'softmax'}}
    cb = {'model_checkpoint': {'filepath': str(tmp_path / 'checkpoints' / 'unet' / 'best.torch'), 'epoch_every': 1, 'minimize': True},
          'exp_lr_scheduler': {'gamma': 0.5, 'epoch_every': 1},
          'plateau_lr_scheduler': {'lr_factor': 0.5, 'lr_patience': 1, 'epoch_every': 1},
          'training_monitor': {'batch_every': 1, 'epoch_every': 1},
          'experiment_timing': {'batch_every': 10, 'epoch_every': 1},
          'validation_monitor': {'epoch_every': 1, 'data_dir': str(tmp_path), 'validate_with_map': False, 'small_annotations_size': 14},
          'neptune_monitor': {'model_name': 'unet', 'image_nr': 1, 'image_resize': 0.2, 'outputs_to_plot': []},
          'early_stopping': {'patience': patience, 'minimize': True}}
    return arch, {'epochs': epochs}, cb


def batches(n_batches, n, hw, weighted):
    out = []
    for b in range(n_batches):
        x = unet_ref.synthetic_batch(n, hw, hw, seed=10 + b)
        t = losses_ref.synthetic_target(n, hw, hw, seed=20 + b)
        out.append([x, t if weighted else t[:, :1].contiguous()])
    return out


class Loader(hip_models.BaseTransformer):
    """stands in for the reference's loader Step: emits the (iterable, steps) pairs the model Step consumes"""

    def __init__(self, train, valid):
        self.train, self.valid = train, valid

    def transform(self, **kwargs):
        return {'datagen': (self.train, len(self.train)), 'validation_datagen': (self.valid, len(self.valid))}

    def save(self, filepath):
        import joblib
        joblib.dump({}, filepath)

    def load(self, filepath):
        return self


@needs_ref
def test_hip_transformer_inside_reference_step_with_reference_callbacks(interpreted, tmp_path, caplog):
    ref_models = ref_import.ref('models')
    ref_base = ref_import.ref('steps.base')
    ref_cb = ref_import.ref('steps.pytorch.callbacks')
    arch, train_cfg, cb_cfg = configs(tmp_path, epochs=4, patience=1)
    t = hip_models.PyTorchUNetWeighted(arch, train_cfg, cb_cfg)
    t.model.load_state_dict(unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34)))
    t.callbacks = ref_models.callbacks_unet(cb_cfg)                 # the reference's own CallbackList
    assert isinstance(t.callbacks, ref_cb.CallbackList) and len(t.callbacks) == 7
    # what the reference's callbacks read from the transformer (src/steps/pytorch/callbacks.py:26-32,58,222)
    assert isinstance(t.optimizer, torch.optim.Optimizer)
    assert t.output_names == ['multichannel_map'] and callable(t.loss_function[0][1]) and t.loss_function[0][2] == 1.0
    train, valid = batches(2, 2, 64, True), batches(1, 2, 64, True)
    loader = ref_base.Step(name='loader', transformer=Loader(train, valid), input_data=['input'], cache_dirpath=str(tmp_path))
    unet = ref_base.Step(name='unet', transformer=t, input_steps=[loader], cache_dirpath=str(tmp_path), is_trainable=True)
    with caplog.at_level(logging.INFO):
        out = unet.fit_transform({'input': {}})
    probs = out['multichannel_map_prediction']
    assert probs.shape == (4, 2, 64, 64) and probs.dtype == np.float32 and np.allclose(probs.sum(1), 1, atol=1e-5)
    # the protocol ran: per-epoch validation through the reference's score_model on the callable loss, early stopping
    # (patience 1: stops once two epochs in a row did not improve, or runs all 4), LR decayed by the reference's scheduler
    epochs_run = len(t.epoch_losses)
    assert 1 <= epochs_run <= 4 and sorted(t.validation_loss) == list(range(epochs_run))
    assert all(v['sum'].numel() == 1 for v in t.validation_loss.values())
    es = [c for c in t.callbacks.callbacks if isinstance(c, ref_cb.EarlyStopping)][0]
    assert epochs_run == 4 or es.training_break()
    assert abs(t.optimizer.param_groups[0]['lr'] - 5e-4 * 0.5 ** epochs_run) < 1e-12
    assert abs(t.optimizer.lr - 5e-4 * 0.5 ** (epochs_run - 1)) < 1e-12          # the rate the last epoch's steps really used
    assert t.optimizer.steps == 2 * epochs_run
    # the checkpoint the reference's ModelCheckpoint wrote has the DataParallel key format and loads back (Model.load, :148-160)
    ckpt = torch.load(cb_cfg['model_checkpoint']['filepath'])
    assert set(ckpt) == {'module.' + k for k in t.model.state_dict()}
    t2 = hip_models.PyTorchUNetWeighted(arch, train_cfg, cb_cfg).load(cb_cfg['model_checkpoint']['filepath'])
    assert all(torch.equal(ckpt['module.' + k], v.cpu()) for k, v in t2.model.state_dict().items())
    # Step caching: the transformer file exists now, a second Step transforms without fitting
    assert os.path.exists(os.path.join(str(tmp_path), 'transformers', 'unet'))
    t3 = hip_models.PyTorchUNetWeighted(arch, train_cfg, cb_cfg)
    unet3 = ref_base.Step(name='unet', transformer=t3, input_steps=[loader], cache_dirpath=str(tmp_path), is_trainable=True)
    out3 = unet3.transform({'input': {}})
    assert out3['multichannel_map_prediction'].shape == (4, 2, 64, 64) and t3.epoch_losses == []


def _synthetic_tiles(root, n_train=4, n_valid=2):
    """300x300 RGB PNG tiles + class-index mask PNGs on disk and the metadata frame the reference's XYSplit / loaders read
    (X_COLUMNS / Y_COLUMNS of src/pipeline_config.py:11-13; masks as overlay_masks writes them, src/preparation.py:84-95)"""
    import pandas as pd
    from PIL import Image
    from scipy import ndimage as ndi
    os.makedirs(os.path.join(root, 'data'), exist_ok=True)
    rng = np.random.default_rng(1234)
    rows = []
    for i in range(n_train + n_valid):
        img = (ndi.gaussian_filter(rng.random((300, 300, 3)), (4, 4, 0)) * 4 % 1 * 255).astype(np.uint8)
        z = ndi.gaussian_filter(rng.standard_normal((300, 300)), 12)
        mask = (z > np.quantile(z, 0.7)).astype(np.uint8)
        fi, fm = os.path.join(root, 'data', 'img_%d.png' % i), os.path.join(root, 'data', 'mask_%d.png' % i)
        Image.fromarray(img).save(fi)
        Image.fromarray(mask).save(fm)
        rows.append({'ImageId': i, 'file_path_image': fi, 'file_path_mask_eroded_0_dilated_0': fm,
                     'is_train': int(i < n_train), 'is_valid': int(i >= n_train)})
    return pd.DataFrame(rows)


def _plain(x):
    return {k: _plain(v) for k, v in x.items()} if isinstance(x, dict) else x


@needs_ref
def test_configs0_the_references_own_unet_pipeline_graph_with_the_hip_transformer_swapped_in(interpreted, tmp_path, monkeypatch):
    """BASELINE.json configs[0] (ResNet34-U-Net, 4 synthetic 300x300 tiles, 1 epoch; plumbing): the graph is the REFERENCE's --
    src.pipelines.PIPELINES['unet']['train'] built from src.pipeline_config.SOLUTION_CONFIG (src/pipelines.py:12-52,395-411), its
    XYSplit, its MetadataImageSegmentationLoaderResize reading PNG tiles from disk (src/loaders.py:287-304), its Step caching, its
    callbacks (neptune stubbed), its six mask_postprocessing Steps -- driven the way src/pipeline_manager.py:126-137 drives it;
    the ONLY change is the one INTEGRATION.md section 3 names: get_step('unet').transformer = the HIP transformer built from the
    same config.unet.  (The reference's own transformer cannot finish an epoch on this torch: `loss.data.cpu().numpy()[0]`,
    src/steps/pytorch/callbacks.py:135, indexes a 0-d array.)  Compute back end: tests/emu.py.  Then the inference graph
    (train_mode False) loads the transformer the train graph persisted and its y_pred equals the oracle chain on the transformer's
    own probabilities."""
    from oracle import post_ref
    pc, pl = ref_import.ref('pipeline_config'), ref_import.ref('pipelines')
    from attrdict import AttrDict                      # the shim ref_import puts on the path
    root = str(tmp_path)
    meta = _synthetic_tiles(root)
    cfg = _plain(pc.SOLUTION_CONFIG)
    cfg['env']['cache_dirpath'] = os.path.join(root, 'exp')
    cfg['execution'].update(num_workers=0, batch_size_train=2, batch_size_inference=2, loader_mode='resize', stream_mode=False)
    for k in ('training', 'inference'):
        cfg['loader']['loader_params'][k].update(batch_size=2, num_workers=0, pin_memory=False)
    cfg['loader']['dataset_params'].update(h=64, w=64)             # network input edge (256 in neptune.yaml): the interpreter is slow
    u = cfg['unet']
    u['architecture_config']['model_params']['encoder'] = 'ResNet34'
    u['training_config']['epochs'] = 1
    u['callbacks_config']['model_checkpoint'].update(filepath=os.path.join(root, 'exp', 'checkpoints', 'unet', 'best.torch'), minimize=True)
    u['callbacks_config']['validation_monitor'].update(validate_with_map=0, data_dir=root)
    u['callbacks_config']['early_stopping']['minimize'] = True
    cfg = AttrDict(cfg)

    def hip_transformer():
        ucfg = _plain(cfg.unet)
        ucfg['architecture_config']['model_params']['compute_dtype'] = 'fp32'
        return hip_models.PyTorchUNet(**ucfg)
    pipe = pl.PIPELINES['unet']['train'](cfg)
    assert type(pipe.get_step('unet').transformer).__module__ == 'src.models'            # the reference's class, about to be swapped
    t = hip_transformer()
    t.model.load_state_dict(unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34)))
    pipe.get_step('unet').transformer = t
    train_meta, valid_meta = meta[meta.is_train == 1], meta[meta.is_valid == 1]
    data = {'input': {'meta': train_meta, 'target_sizes': [(300, 300)] * len(train_meta), 'annotations': None},
            'specs': {'train_mode': True, 'num_threads': 1}, 'callback_input': {'meta_valid': valid_meta}}
    pipe.clean_cache()
    out = pipe.fit_transform(data)
    assert list(out) == ['y_pred'] and len(out['y_pred']) == 4
    for labels, scores in out['y_pred']:
        assert labels.shape == (2, 300, 300) and labels.dtype == np.int32
        assert [len(s) for s in scores] == [int(l.max()) for l in labels]
    # one epoch over 2 batches of 2 tiles ran through the reference's callback list; validation on the 2 valid tiles; checkpoint
    assert len(t.epoch_losses) == 1 and t.optimizer.steps == 2 and sorted(t.validation_loss) == [0]
    assert os.path.exists(cfg.unet.callbacks_config.model_checkpoint.filepath)
    assert os.path.exists(os.path.join(root, 'exp', 'transformers', 'unet'))              # Step persisted the fitted transformer
    # inference graph: loader without shuffling, the Step LOADS the persisted HIP transformer (Model.load, module.-prefixed keys)
    # (inference feeds y = None through `squeeze_inputs` = np.squeeze(None, axis=1), src/utils.py:227-228: the numpy of the reference's era
    # ignored the axis for objects without a squeeze method and returned array(None); numpy 2 raises -- restore that one behaviour)
    ref_squeeze = pl.squeeze_inputs
    monkeypatch.setattr(pl, 'squeeze_inputs', lambda inputs: None if inputs[0] is None else ref_squeeze(inputs))
    inf = pl.PIPELINES['unet']['inference'](cfg)
    t2 = hip_transformer()
    inf.get_step('unet').transformer = t2
    data_inf = {'input': {'meta': train_meta, 'target_sizes': [(300, 300)] * len(train_meta)},
                'specs': {'train_mode': False, 'num_threads': 1}, 'callback_input': {'meta_valid': None}}
    out2 = inf.transform(data_inf)
    assert t2.epoch_losses == [] and all(torch.equal(a.cpu(), b.cpu()) for a, b in zip(t.model.state_dict().values(), t2.model.state_dict().values()))
    loader_out = inf.get_step('loader').transform(data_inf)
    probs = t2.transform(loader_out['datagen'])['multichannel_map_prediction']
    assert probs.shape == (4, 2, 64, 64)
    for p, (labels, scores) in zip(probs, out2['y_pred']):
        r = post_ref.resize_image(p, (300, 300))
        exp = post_ref.dilate_image(post_ref.label_multilayer_image(post_ref.categorize_multilayer_image(r)), cfg.postprocessor.mask_dilation.dilate_selem_size)
        assert (labels == exp).all()
        _, exp_sc = post_ref.build_score(exp, r)
        assert all(np.allclose(a, b, rtol=1e-9) for a, b in zip(scores, exp_sc))


def test_fit_with_own_callbacks_validates_stops_early_and_checkpoints_best(interpreted, tmp_path):
    """the standalone mirror of the protocol (mapping_challenge_amd.callbacks): same observable behaviour"""
    from mapping_challenge_amd import callbacks as cb
    arch, train_cfg, cb_cfg = configs(tmp_path, epochs=4, patience=1)
    t = hip_models.PyTorchUNet(arch, train_cfg, cb_cfg)
    assert [type(c).__name__ for c in t.callbacks.callbacks] == ['ExperimentTiming', 'TrainingMonitor', 'ValidationMonitor',
                                                                 'ModelCheckpoint', 'ExponentialLRScheduler', 'EarlyStopping']
    t.model.load_state_dict(unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34)))
    train, valid = batches(2, 2, 64, False), batches(1, 2, 64, False)
    t.fit((train, len(train)), validation_datagen=(valid, len(valid)))
    n = len(t.epoch_losses)
    assert 2 <= n <= 4 and sorted(t.validation_loss) == list(range(n))
    vals = [float(t.validation_loss[e]['sum']) for e in range(n)]
    if n < 4:                                    # stopped early: the last two epochs did not improve on the best before them
        assert min(vals[-2:]) >= min(vals[:-2] or vals[:1])
    # validation loss == the oracle's loss of the model state after that epoch would need the states; check the value of
    # the LAST epoch against the oracle evaluated on the final weights when the last epoch is the one kept in memory
    ref = unet_ref.UNetResNetRef(34)
    ref.load_state_dict({k: v.cpu() for k, v in t.model.state_dict().items()})
    ref.eval()
    with torch.no_grad():
        expect = losses_ref.segmentation_ce(ref(valid[0][0]), valid[0][1]).item()
    assert abs(vals[-1] - expect) < 1e-4
    best = torch.load(cb_cfg['model_checkpoint']['filepath'])
    assert all(k.startswith('module.') for k in best)
    tm = [c for c in t.callbacks.callbacks if isinstance(c, cb.TrainingMonitor)][0]
    assert len(tm.epoch_means) == n and np.allclose([m['sum'] for m in tm.epoch_means], t.epoch_losses, rtol=1e-6)


def test_hiploss_is_a_differentiable_reference_style_callable(interpreted):
    """loss_function entries work in the reference's own _fit_loop: `loss = fn(outputs, target) * weight; loss.backward()`"""
    from mapping_challenge_amd.trainer import HipLoss, LossSpec
    arch, _, _ = configs('.')
    out = torch.randn(2, 2, 16, 16, requires_grad=True)
    tgt = losses_ref.synthetic_target(2, 16, 16)
    for fn, ref_fn, t in ((HipLoss(LossSpec.mixed(arch)), losses_ref.mixed_dice_ce, tgt),
                          (HipLoss(LossSpec.plain_ce()), losses_ref.segmentation_ce, tgt[:, :1].contiguous())):
        out.grad = None
        loss = fn(out, t) * 0.5
        assert loss.shape == (1,)
        loss.backward()
        g = out.grad.clone()
        out.grad = None
        lr = ref_fn(out, t) * 0.5
        lr.backward()
        assert abs(loss.item() - lr.item()) < 1e-6 and (g - out.grad).abs().max().item() < 1e-8
    with pytest.raises(ValueError, match='3 channels'):
        HipLoss(LossSpec.mixed(arch))(out, tgt[:, :1])


def test_albunet_key_builds_the_resnet34_unet(interpreted, tmp_path):
    arch, train_cfg, cb_cfg = configs(tmp_path, encoder='AlbuNet')
    t = hip_models.PyTorchUNet(arch, train_cfg, cb_cfg)
    assert isinstance(t.model, um.AlbuNet) and t.model.encoder_depth == 34 and t.model.num_classes == 2
    ref = unet_ref.UNetResNetRef(34)
    sd = unet_ref.seeded_state_dict(ref)
    assert set(t.model.state_dict()) == set(sd)
    t.model.load_state_dict(sd)
    ref.load_state_dict(sd)
    x = unet_ref.synthetic_batch(1, 64, 64)
    ref.eval()
    with torch.no_grad():
        expect = torch.softmax(ref(x), 1).numpy()
    got = t.transform(([[x]], 1))['multichannel_map_prediction']
    assert np.abs(got - expect).max() < 1e-5
    if ref_import.available():                   # the literal reference class has the same parameter key set
        ref_albu = ref_import.ref('unet_models').AlbuNet(num_classes=2, pretrained=False, is_deconv=True)
        assert set(ref_albu.state_dict()) == set(sd)


def test_pretrained_request_without_weights_warns_and_encoder_checkpoint_loads(interpreted, tmp_path):
    arch, train_cfg, cb_cfg = configs(tmp_path, epochs=1)
    t = hip_models.PyTorchUNet(arch, train_cfg, {})
    train = batches(1, 1, 64, False)
    with pytest.warns(UserWarning, match='ImageNet-pretrained'):
        t.fit((train, 1))
    from oracle.shims.torchvision import models as tvm
    enc = tvm.resnet34()
    path = str(tmp_path / 'resnet34.pth')
    torch.save(enc.state_dict(), path)
    arch['model_params']['encoder_weights'] = path
    t2 = hip_models.PyTorchUNet(arch, train_cfg, {})
    assert t2.model.weights_loaded and torch.equal(t2.model.encoder.layer2[0].conv1.weight.cpu(), enc.layer2[0].conv1.weight)
    with pytest.raises(KeyError):
        t2.model.load_encoder_state_dict({'bogus.weight': torch.zeros(1)})


def test_train_step_keeps_state_per_batch_shape(interpreted):
    """the reference's DataLoader has no drop_last: alternating batch sizes must each use their own program / buffers"""
    from mapping_challenge_amd.trainer import HipAdam, LossSpec, TrainStep
    net = um.UNetResNet(34, 2, num_filters=32, dropout_2d=0.0, is_deconv=True, compute_dtype='fp32')
    net.load_state_dict(unet_ref.seeded_state_dict(unet_ref.UNetResNetRef(34)))
    net.flatten_parameters('cpu')
    net.train()
    step = TrainStep(net, LossSpec.plain_ce(), HipAdam(net, lr=1e-3))
    xa, xb = unet_ref.synthetic_batch(2, 64, 64), unet_ref.synthetic_batch(1, 64, 64, seed=3)
    ta, tb = losses_ref.synthetic_target(2, 64, 64)[:, :1].contiguous(), losses_ref.synthetic_target(1, 64, 64, seed=3)[:, :1].contiguous()
    ref = unet_ref.UNetResNetRef(34)
    ref.load_state_dict(unet_ref.seeded_state_dict(ref))
    ref.train()
    opt = torch.optim.Adam([p for n_, p in ref.named_parameters() if not n_.startswith('encoder.fc')], lr=1e-3)
    for x, t in ((xa, ta), (xb, tb), (xa, ta), (xb, tb)):
        loss = step(x, t).item()
        opt.zero_grad()
        lr = losses_ref.segmentation_ce(ref(x), t)
        lr.backward()
        opt.step()
        assert abs(loss - lr.item()) < 2e-3 * max(1.0, abs(lr.item())), (loss, lr.item())
        assert step.prog.logits.shape[0] == x.shape[0] and step.x.shape == x.shape
    assert len(step.shapes) == 2
