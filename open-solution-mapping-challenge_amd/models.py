"""Model transformers for the `unet` Step: drop-ins for the reference's `PyTorchUNet`,
`PyTorchUNetWeighted` and their `*Stream` variants (`src/models.py:50-209` on top of
`src/steps/pytorch/models.py:18-171`).

Same constructor `(architecture_config, training_config, callbacks_config)` (config shape:
src/pipeline_config.py:61-120), same `fit(datagen, validation_datagen=None, meta_valid=None)`,
`transform(datagen, validation_datagen=None) -> {'multichannel_map_prediction': f32[N,2,H,W]}` (softmax
probabilities, src/models.py:88-92), `load(filepath)` / `save(filepath)` on the reference's
`module.`-prefixed state_dict format (src/steps/pytorch/models.py:148-171).  `datagen` is the
reference's `(iterable of [X] or [X, target] CPU tensors, steps)` pair (src/loaders.py:203-204).

Everything the reference does on the device between H2D and D2H runs in HIP kernels: the network
(unet_models.UNetResNet), the loss, Adam+L2 (trainer.py) and the channel softmax (fused into the last
conv instead of the host numpy pass of src/utils.py:231-273).
"""
import logging
import os

import numpy as np
import torch

from . import _lib
from .distributed import World
from .postprocessing import _to_host
from .steps import BaseTransformer
from .trainer import HipAdam, LossSpec, TrainStep
from .unet_models import UNetResNet

logger = logging.getLogger('mapping-challenge')

# src/models.py:22-47 (only the UNetResNet encoders are on the hot path)
PRETRAINED_NETWORKS = {
    'ResNet34': {'encoder_depth': 34, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
    'ResNet101': {'encoder_depth': 101, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
    'ResNet152': {'encoder_depth': 152, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
}


class BasePyTorchUNet(BaseTransformer):
    loss_name = 'multichannel_map'

    def __init__(self, architecture_config, training_config, callbacks_config):
        _lib.load()                                  # fail loudly if the HIP library is missing
        self.architecture_config = architecture_config
        self.training_config = training_config
        self.callbacks_config = callbacks_config or {}
        self.validation_loss = {}
        self.set_model()
        opt = dict(architecture_config.get('optimizer_params', {}))
        reg = architecture_config.get('regularizer_params', {})
        wd = reg.get('weight_decay_conv2d', 0.0) if reg.get('regularize', False) else 0.0   # src/models.py:287-292
        self.optimizer = HipAdam(self.model, lr=opt.get('lr', 1e-3), weight_decay=wd)
        self.loss_spec = None
        self.loss_function = None                    # [(name, spec, weight)] set by subclasses
        self.world = World()
        self.epoch_losses = []

    # ---- reference API -------------------------------------------------------------------------
    @property
    def output_names(self):
        return [name for (name, _, _) in self.loss_function]

    def set_model(self):
        params = self.architecture_config['model_params']
        encoder = params['encoder']
        if encoder not in PRETRAINED_NETWORKS:
            raise NotImplementedError('HIP path implements the UNetResNet encoders %s (got %r)'
                                      % (sorted(PRETRAINED_NETWORKS), encoder))
        dtype = params.get('compute_dtype', 'bf16')
        self.model = UNetResNet(compute_dtype=dtype, **PRETRAINED_NETWORKS[encoder])

    def _device(self):
        if not torch.cuda.is_available():
            raise _lib.MscError('HIP transformers need a ROCm GPU: the product has no CPU path')
        return torch.device('cuda', torch.cuda.current_device())

    def fit(self, datagen, validation_datagen=None, meta_valid=None):
        dev = self._device()
        self.model.train()
        self.model.flatten_parameters(dev)
        self.world.sync_model(self.model)
        step = TrainStep(self.model, self.loss_spec, self.optimizer, world=self.world,
                         use_graph=bool(self.training_config.get('use_graph', False)))
        batch_gen, steps = datagen
        gamma = self.callbacks_config.get('exp_lr_scheduler', {}).get('gamma', 1.0)
        for epoch_id in range(self.training_config['epochs']):
            losses = []
            for batch_id, data in enumerate(batch_gen):
                X, target = data[0], data[1]
                loss = step(X.to(dev, non_blocking=True), target.to(dev, non_blocking=True).float())
                losses.append(loss.clone())
                if batch_id == steps:
                    break
            if losses:
                mean = float(torch.stack(losses).mean().item())   # one D2H per epoch, not per batch
                self.epoch_losses.append(mean)
                logger.info('epoch {0} sum: {1:.5f}'.format(epoch_id, mean))
            if gamma != 1.0:                                       # ExponentialLRScheduler, per epoch
                self.optimizer.set_lr(self.optimizer.lr * gamma)
            self._checkpoint()
        self.model.weights_changed()
        return self

    def _checkpoint(self):
        cfg = self.callbacks_config.get('model_checkpoint')
        if cfg and cfg.get('filepath') and self.world.rank == 0:
            os.makedirs(os.path.dirname(cfg['filepath']) or '.', exist_ok=True)
            self._save_state(cfg['filepath'])

    def _forward_probs(self, datagen):
        dev = self._device()
        batch_gen, steps = datagen
        for batch_id, data in enumerate(batch_gen):
            X = data[0] if isinstance(data, (list, tuple)) else data
            yield self.model.predict_proba(X.to(dev, non_blocking=True))
            if batch_id == steps:
                break

    def _transform(self, datagen, validation_datagen=None):
        outs = [_to_host(p) for p in self._forward_probs(datagen)]
        return {'{}_prediction'.format(self.output_names[0]): np.vstack(outs)}

    def transform(self, datagen, validation_datagen=None, *args, **kwargs):
        return self._transform(datagen, validation_datagen)

    def transform_tta(self, datagen, tta_transformations, method='gmean'):
        """`unet_tta` in one transformer (src/pipelines.py:94-143, src/loaders.py:401-517): every batch is expanded to its
        TTA variants on the device, predicted, un-transformed and aggregated; returns the same dict as transform()."""
        from . import tta
        specs = tta.tta_specs(**dict(tta_transformations))
        dev = self._device()
        batch_gen, steps = datagen
        outs = []
        for batch_id, data in enumerate(batch_gen):
            X = data[0] if isinstance(data, (list, tuple)) else data
            outs.append(_to_host(tta.predict_tta(self.model, X.to(dev, non_blocking=True), specs, method)))
            if batch_id == steps:
                break
        return {'{}_prediction'.format(self.output_names[0]): np.vstack(outs)}

    def transform_device(self, datagen):
        """generator of cuda f32 [n,2,H,W] probability batches (stays on the device for postprocess_batch)"""
        for p in self._forward_probs(datagen):
            yield p.clone()

    # ---- persistence: the reference's DataParallel-prefixed state_dict ---------------------------
    def _save_state(self, filepath):
        state = {'module.' + k: v.detach().cpu().contiguous() for k, v in self.model.state_dict().items()}
        torch.save(state, filepath)

    def save(self, filepath):
        ckpt = (self.callbacks_config.get('model_checkpoint') or {}).get('filepath')
        if ckpt and os.path.exists(ckpt) and os.path.abspath(ckpt) != os.path.abspath(filepath):
            import shutil
            shutil.copyfile(ckpt, filepath)
        else:
            self._save_state(filepath)

    def load(self, filepath):
        self.model.eval()
        state = torch.load(filepath, map_location='cpu')
        self.model.load_state_dict(state)
        return self


class PyTorchUNet(BasePyTorchUNet):
    """src/models.py:104-107: plain cross entropy."""

    def __init__(self, architecture_config, training_config, callbacks_config):
        super().__init__(architecture_config, training_config, callbacks_config)
        self.loss_spec = LossSpec.plain_ce()
        self.loss_function = [(self.loss_name, self.loss_spec, 1.0)]


class PyTorchUNetWeighted(BasePyTorchUNet):
    """src/models.py:149-161: 0.2 * soft Dice + 1.0 * distance/size weighted cross entropy."""

    def __init__(self, architecture_config, training_config, callbacks_config):
        super().__init__(architecture_config, training_config, callbacks_config)
        self.loss_spec = LossSpec.mixed(architecture_config)
        self.loss_function = [(self.loss_name, self.loss_spec, 1.0)]


class _StreamMixin:
    """src/models.py:115-146: transform yields per-image softmax maps f32[2,H,W] instead of one array."""

    def transform(self, datagen, validation_datagen=None, *args, **kwargs):
        return {'{}_prediction'.format(self.output_names[0]): self._stream(datagen)}

    def _stream(self, datagen):
        for probs in self._forward_probs(datagen):
            for image in _to_host(probs):
                yield image


class PyTorchUNetStream(_StreamMixin, PyTorchUNet):
    pass


class PyTorchUNetWeightedStream(_StreamMixin, PyTorchUNetWeighted):
    pass
