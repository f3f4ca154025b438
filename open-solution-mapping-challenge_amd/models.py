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
from .callbacks import callbacks_unet
from .distributed import World, wire_for
from .postprocessing import _to_host
from .steps import BaseTransformer
from .trainer import HipAdam, HipLoss, LossSpec, TrainStep
from .unet_models import AlbuNet, UNetResNet

logger = logging.getLogger('mapping-challenge')

# src/models.py:22-47 (the ResNet-encoder U-Nets are the hot path; VGG11 / VGG16 / from_scratch are not)
PRETRAINED_NETWORKS = {
    'AlbuNet': {'model': AlbuNet, 'num_classes': 2, 'pretrained': True, 'is_deconv': True},
    'ResNet34': {'encoder_depth': 34, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
    'ResNet101': {'encoder_depth': 101, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
    'ResNet152': {'encoder_depth': 152, 'num_classes': 2, 'num_filters': 32, 'dropout_2d': 0.0, 'pretrained': True, 'is_deconv': True},
}





def _compute_device():
    """the device the transformers compute on: the current ROCm GPU -- the product has no CPU path"""
    if not torch.cuda.is_available():
        raise _lib.MscError('HIP transformers need a ROCm GPU: the product has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())

class _ModuleView(torch.nn.Module):
    """What the callbacks get as `transformer.model`: the reference hands them the nn.DataParallel wrapper
    (src/models.py:65), i.e. a callable whose state_dict keys carry the `module.` prefix and that save_model
    (src/steps/pytorch/utils.py:67-75) moves to the CPU and back around torch.save.  Here the parameters live in ONE flat
    device buffer that must not be re-homed, so cpu() / cuda() are no-ops and state_dict() returns host copies."""

    def __init__(self, net):
        super().__init__()
        self.module = net

    def forward(self, x):
        return self.module(x)

    def cpu(self):
        return self

    def cuda(self, device=None):
        return self

    def state_dict(self, *args, **kwargs):
        return {k: v.detach().cpu().contiguous() for k, v in super().state_dict(*args, **kwargs).items()}


class _TransformerView:
    """the attributes Callback.set_params reads from the transformer (src/steps/pytorch/callbacks.py:26-32)"""

    def __init__(self, t):
        self.model = _ModuleView(t.model)
        self.optimizer, self.loss_function = t.optimizer, t.loss_function
        self.output_names, self.validation_loss = t.output_names, t.validation_loss
        self.callbacks = t.callbacks
        self.world = t.world          # the product's callbacks average the validation loss over ranks and save on rank 0 only


class BasePyTorchUNet(BaseTransformer):
    loss_name = 'multichannel_map'

    def __init__(self, architecture_config, training_config, callbacks_config):
        _lib.load()                                  # fail loudly if the HIP library is missing
        self.architecture_config = architecture_config
        self.training_config = training_config
        self.callbacks_config = callbacks_config or {}
        self.validation_loss = {}
        self.set_model()
        if (training_config or {}).get('deterministic') is not None:      # fixed-order gradient sums (UNetResNet.deterministic); default: MSC_DETERMINISTIC
            self.model.deterministic = bool(training_config['deterministic'])
        opt = dict(architecture_config.get('optimizer_params', {}))
        reg = architecture_config.get('regularizer_params', {})
        wd = reg.get('weight_decay_conv2d', 0.0) if reg.get('regularize', False) else 0.0   # src/models.py:287-292
        self.optimizer = HipAdam(self.model, lr=opt.get('lr', 1e-3), weight_decay=wd)
        self.loss_spec = None
        self.loss_function = None                    # [(name, callable(output, target) -> loss, weight)] set by subclasses
        # gradient exchange over ranks: fp32 ring all-reduce by default -- the precision of the reference's reduce-add (nn.DataParallel,
        # src/models.py:65).  training_config['grad_wire'] = 'bf16' (or MSC_GRAD_WIRE) opts into the 16-bit all-to-all wire: half the
        # bytes, every rank's partial rounded once before the sum (distributed.World.all_reduce_grad_range; always bf16 -- under fp16
        # compute the gradients still carry the loss scale, an fp16 wire would overflow at |g| > 65504 / scale)
        wire = (training_config or {}).get('grad_wire', 'fp32')
        self.world = World(grad_wire='fp32' if wire == 'fp32' else wire_for(self.model.compute_dtype))
        self.callbacks = callbacks_unet(self.callbacks_config)     # replaceable by the reference's own CallbackList
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
        config = dict(PRETRAINED_NETWORKS[encoder])
        self.model = config.pop('model', UNetResNet)(compute_dtype=dtype, **config)
        if params.get('encoder_weights'):            # local torchvision checkpoint standing in for the model-zoo download
            self.model.load_encoder_state_dict(params['encoder_weights'])

    def _device(self):
        return _compute_device()

    def fit(self, datagen, validation_datagen=None, meta_valid=None):
        """src/models.py:62-86: the callback-driven epoch / batch loop around `_fit_loop`; nn.DataParallel is replaced by
        one process per GPU (distributed.World), the callbacks see the model through a `module.`-prefixing view so the
        checkpoints they write have the reference's format"""
        dev = self._device()
        if self.model.pretrained_requested and not self.model.weights_loaded:
            import warnings
            warnings.warn("encoder '%s' asks for ImageNet-pretrained weights (src/models.py:22-47) but none were loaded: training "
                          "starts from random encoder weights. Pass architecture_config['model_params']['encoder_weights'] (a local "
                          'torchvision ResNet state_dict) or load() a checkpoint first.' % self.architecture_config['model_params']['encoder'])
        self.model.train()
        self.model.flatten_parameters(dev)
        self.world.sync_model(self.model)
        self._step = TrainStep(self.model, self.loss_spec, self.optimizer, world=self.world,
                               use_graph=bool(self.training_config.get('use_graph', False)),
                               loss_scale=self.training_config.get('loss_scale'))
        view = _TransformerView(self)
        self.callbacks.set_params(view, validation_datagen=validation_datagen, meta_valid=meta_valid)
        self.callbacks.on_train_begin()
        batch_gen, steps = datagen
        for epoch_id in range(self.training_config['epochs']):
            self.callbacks.on_epoch_begin()
            losses = []
            for batch_id, data in enumerate(batch_gen):
                self.callbacks.on_batch_begin()
                metrics = self._fit_loop(data)
                losses.append(metrics['sum'])
                self.callbacks.on_batch_end(metrics=metrics)
                if batch_id == steps:
                    break
            if losses:
                self.epoch_losses.append(float(torch.stack(losses).mean().item()))     # one D2H per epoch
            self.callbacks.on_epoch_end()
            if self._training_break():
                break
        self.callbacks.on_train_end()
        self.model.weights_changed()
        return self

    def _training_break(self):
        """`callbacks.training_break()`, agreed between the ranks (max): whatever the callbacks are (the reference's own
        decide from a rank-local validation loss), either every rank leaves the epoch loop or none does"""
        stop = bool(self.callbacks.training_break())
        if self.world.size > 1:
            flag = torch.tensor([1.0 if stop else 0.0], device=self._device())
            self.world.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
            stop = bool(flag.item() > 0)
        return stop

    def _fit_loop(self, data):
        """src/steps/pytorch/models.py:76-113 on the fused step (trainer.TrainStep); returns {'sum': loss[1]}"""
        dev = self._device()
        X, target = data[0], data[1]
        loss = self._step(X.to(dev, non_blocking=True), target.to(dev, non_blocking=True).float())
        return {'sum': loss.clone()}

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
        self.loss_function = [(self.loss_name, HipLoss(self.loss_spec, 'multiclass_segmentation_loss'), 1.0)]


class PyTorchUNetWeighted(BasePyTorchUNet):
    """src/models.py:149-161: 0.2 * soft Dice + 1.0 * distance/size weighted cross entropy."""

    def __init__(self, architecture_config, training_config, callbacks_config):
        super().__init__(architecture_config, training_config, callbacks_config)
        self.loss_spec = LossSpec.mixed(architecture_config)
        self.loss_function = [(self.loss_name, HipLoss(self.loss_spec, 'mixed_dice_cross_entropy_loss'), 1.0)]


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
