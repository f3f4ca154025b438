"""Training callbacks: the protocol of the reference's `src/steps/pytorch/callbacks.py:15-104` (`Callback`,
`CallbackList`) and the callbacks `callbacks_unet` wires up (`src/models.py:295-307`), for running the HIP transformers
standalone -- the reference's own `steps` package is not on the GPU box and its neptune / COCO-evaluation monitors need
services that are out of scope.

Dropped into the reference tree the transformers take the reference's own `CallbackList` instead
(`transformer.callbacks = callbacks_unet(config)`): `fit` only relies on the protocol --
`set_params(transformer, validation_datagen=, meta_valid=)`, `on_{train,epoch,batch}_{begin,end}`, `training_break()` --
and hands the callbacks what they read from the transformer: `model` (callable, `eval()/train()/state_dict()`),
`optimizer` (a `torch.optim.Optimizer`), `loss_function = [(name, callable, weight)]`, `output_names`, `validation_loss`.
"""
import logging
import os
from datetime import datetime, timedelta

import torch

logger = logging.getLogger('mapping-challenge')


def _scalar(loss):
    """python float of a loss tensor of one element (0-d or [1])"""
    return float(loss.detach().reshape(-1)[0].item())


def score_model(model, loss_function, datagen):
    """src/steps/pytorch/validation.py:51-76 for single-output models: average of `loss_fn(model(X), target) * weight`
    over the validation batches -> {'sum': tensor[1]}"""
    batch_gen, steps = datagen
    dev = next((p.device for p in model.parameters()), torch.device('cpu'))
    total, count = None, 0
    with torch.no_grad():
        for batch_id, data in enumerate(batch_gen):
            X, target = data[0].to(dev), data[1].to(dev)
            (name, fn, weight) = loss_function[0]
            loss = fn(model(X), target) * weight
            total = loss.detach().clone() if total is None else total + loss.detach()
            count += 1
            if batch_id == steps:
                break
    if total is None:
        return {'sum': torch.zeros(1)}
    return {'sum': total / max(steps, 1)}       # the reference divides by `steps`, not by the batches seen (:75)


class Callback:
    def __init__(self):
        self.epoch_id = self.batch_id = None
        self.model = self.optimizer = self.loss_function = self.output_names = None
        self.validation_datagen = self.lr_scheduler = None
        self.validation_loss = None
        self.world = None                 # distributed.World of the transformer (one process per GPU); None = single process

    def set_params(self, transformer, validation_datagen, *args, **kwargs):
        self.model, self.optimizer = transformer.model, transformer.optimizer
        self.loss_function, self.output_names = transformer.loss_function, transformer.output_names
        self.validation_datagen = validation_datagen
        self.validation_loss = transformer.validation_loss
        self.world = getattr(transformer, 'world', None)

    def on_train_begin(self, *args, **kwargs):
        self.epoch_id, self.batch_id = 0, 0

    def on_train_end(self, *args, **kwargs):
        pass

    def on_epoch_begin(self, *args, **kwargs):
        pass

    def on_epoch_end(self, *args, **kwargs):
        self.epoch_id += 1

    def training_break(self, *args, **kwargs):
        return False

    def on_batch_begin(self, *args, **kwargs):
        pass

    def on_batch_end(self, *args, **kwargs):
        self.batch_id += 1

    def get_validation_loss(self):
        if self.epoch_id not in self.validation_loss:
            self.model.eval()
            score = score_model(self.model, self.loss_function, self.validation_datagen)
            self.model.train()
            # one process per GPU: every rank validates its own shard with its own BatchNorm running statistics; the
            # decisions taken from the value (best checkpoint, early stopping) must be the same on every rank or a rank
            # that stops alone leaves the others in the next gradient all-reduce -> mean over ranks
            if self.world is not None and self.world.size > 1:
                # the collective runs on the backend's device: RCCL cannot reduce the CPU zeros score_model returns for a rank whose
                # validation shard is empty
                import torch.distributed as dist
                on_gpu = dist.is_initialized() and dist.get_backend(self.world.group) == 'nccl'
                dev = next(self.model.parameters()).device if on_gpu else None
                score = {k: self.world.all_reduce(v.detach().to(dev if on_gpu else v.device, torch.float32).clone()) / self.world.size
                         for k, v in score.items()}
            self.validation_loss[self.epoch_id] = score
        return self.validation_loss[self.epoch_id]


class CallbackList:
    def __init__(self, callbacks=None):
        self.callbacks = [] if callbacks is None else ([callbacks] if isinstance(callbacks, Callback) else list(callbacks))

    def __len__(self):
        return len(self.callbacks)

    def _each(self, name, *args, **kwargs):
        return [getattr(c, name)(*args, **kwargs) for c in self.callbacks]

    def set_params(self, *a, **k): self._each('set_params', *a, **k)
    def on_train_begin(self, *a, **k): self._each('on_train_begin', *a, **k)
    def on_train_end(self, *a, **k): self._each('on_train_end', *a, **k)
    def on_epoch_begin(self, *a, **k): self._each('on_epoch_begin', *a, **k)
    def on_epoch_end(self, *a, **k): self._each('on_epoch_end', *a, **k)
    def on_batch_begin(self, *a, **k): self._each('on_batch_begin', *a, **k)
    def on_batch_end(self, *a, **k): self._each('on_batch_end', *a, **k)

    def training_break(self, *a, **k):
        return any(self._each('training_break', *a, **k))


def _every(v):
    return False if v == 0 else v


class TrainingMonitor(Callback):
    """callbacks.py:107-145, with the per-batch losses kept on the device: one D2H per logged batch / per epoch instead
    of one per batch"""

    def __init__(self, epoch_every=None, batch_every=None):
        super().__init__()
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)
        self.sums, self.epoch_means = {}, []

    def on_train_begin(self, *args, **kwargs):
        self.sums, self.epoch_id, self.batch_id = {}, 0, 0

    def on_batch_end(self, metrics, *args, **kwargs):
        for name, loss in metrics.items():
            tot, n = self.sums.get(name, (None, 0))
            self.sums[name] = (loss.detach().clone() if tot is None else tot + loss.detach(), n + 1)
            if self.batch_every and self.batch_id % self.batch_every == 0:
                logger.info('epoch {0} batch {1} {2}:     {3:.5f}'.format(self.epoch_id, self.batch_id, name, _scalar(loss)))
        self.batch_id += 1

    def on_epoch_end(self, *args, **kwargs):
        means = {name: _scalar(tot) / n for name, (tot, n) in self.sums.items() if n}
        self.sums = {}
        self.epoch_means.append(means)
        if self.epoch_every and self.epoch_id % self.epoch_every == 0:
            for name, v in means.items():
                logger.info('epoch {0} {1}:     {2:.5f}'.format(self.epoch_id, name, v))
        self.epoch_id += 1


class ValidationMonitor(Callback):
    """callbacks.py:148-169 (the reference's ValidationMonitorSegmentation with validate_with_map off,
    src/callbacks.py:106-128; COCO-AP validation needs pycocotools and the annotation files: out of scope)"""

    def __init__(self, epoch_every=None, batch_every=None, validate_with_map=False, **unused):
        super().__init__()
        if validate_with_map:
            raise NotImplementedError('validate_with_map needs the COCO evaluation of the reference pipeline; use the '
                                      "reference's own callbacks for it (INTEGRATION.md)")
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_every and self.epoch_id % self.epoch_every == 0 and self.validation_datagen is not None:
            for name, loss in self.get_validation_loss().items():
                logger.info('epoch {0} validation {1}:     {2:.5f}'.format(self.epoch_id, name, _scalar(loss)))
        self.epoch_id += 1


class EarlyStopping(Callback):
    """callbacks.py:172-202"""

    def __init__(self, patience, minimize=True):
        super().__init__()
        self.patience, self.minimize = patience, minimize
        self.best_score, self.epoch_since_best, self._training_break = None, 0, False

    def on_epoch_end(self, *args, **kwargs):
        if self.validation_datagen is not None:
            loss_sum = _scalar(self.get_validation_loss()['sum'])
            if not self.best_score:
                self.best_score = loss_sum
            if (self.minimize and loss_sum < self.best_score) or (not self.minimize and loss_sum > self.best_score):
                self.best_score, self.epoch_since_best = loss_sum, 0
            else:
                self.epoch_since_best += 1
            if self.epoch_since_best > self.patience:
                self._training_break = True
        self.epoch_id += 1

    def training_break(self, *args, **kwargs):
        return self._training_break


class ExponentialLRScheduler(Callback):
    """callbacks.py:205-244: torch's ExponentialLR on the transformer's optimizer"""

    def __init__(self, gamma, epoch_every=1, batch_every=None):
        super().__init__()
        self.gamma, self.epoch_every, self.batch_every = gamma, _every(epoch_every), _every(batch_every)

    def set_params(self, transformer, validation_datagen, *args, **kwargs):
        super().set_params(transformer, validation_datagen, *args, **kwargs)
        from torch.optim.lr_scheduler import ExponentialLR
        self.lr_scheduler = ExponentialLR(self.optimizer, self.gamma, last_epoch=-1)

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_every and (self.epoch_id + 1) % self.epoch_every == 0:
            self.lr_scheduler.step()
            logger.info('epoch {0} current lr: {1}'.format(self.epoch_id + 1, self.optimizer.param_groups[0]['lr']))
        self.epoch_id += 1

    def on_batch_end(self, *args, **kwargs):
        if self.batch_every and self.batch_id % self.batch_every == 0:
            self.lr_scheduler.step()
        self.batch_id += 1


class ModelCheckpoint(Callback):
    """callbacks.py:247-280: keeps the best (by validation loss; every epoch without validation data) state_dict in
    the reference's DataParallel-prefixed format"""

    def __init__(self, filepath, epoch_every=1, minimize=True):
        super().__init__()
        self.filepath, self.minimize, self.best_score = filepath, minimize, None
        self.epoch_every = _every(epoch_every)

    def on_train_begin(self, *args, **kwargs):
        self.epoch_id, self.batch_id = 0, 0
        os.makedirs(os.path.dirname(self.filepath) or '.', exist_ok=True)

    def on_epoch_end(self, *args, **kwargs):
        if self.epoch_every and self.epoch_id % self.epoch_every == 0:
            save = True
            if self.validation_datagen is not None:
                loss_sum = _scalar(self.get_validation_loss()['sum'])
                if self.best_score is None:
                    self.best_score = loss_sum
                save = ((self.minimize and loss_sum < self.best_score) or (not self.minimize and loss_sum > self.best_score)
                        or self.epoch_id == 0)
                if save:
                    self.best_score = loss_sum
            # one process per GPU: rank 0 writes (its BatchNorm running statistics are the ones kept, as DataParallel keeps
            # replica 0's); the file appears atomically so a reader never sees a half-written archive
            if save and (self.world is None or self.world.rank == 0):
                self.model.eval()
                tmp = '%s.tmp.%d' % (self.filepath, os.getpid())
                torch.save(self.model.state_dict(), tmp)
                os.replace(tmp, self.filepath)
                self.model.train()
                logger.info('epoch {0} model saved to {1}'.format(self.epoch_id, self.filepath))
        self.epoch_id += 1


class ExperimentTiming(Callback):
    """callbacks.py:325-375"""

    def __init__(self, epoch_every=None, batch_every=None):
        super().__init__()
        self.epoch_every, self.batch_every = _every(epoch_every), _every(batch_every)
        self.batch_start = self.epoch_start = None
        self.current_sum = timedelta()

    def on_train_begin(self, *args, **kwargs):
        self.epoch_id, self.batch_id = 0, 0
        logger.info('starting training...')

    def on_train_end(self, *args, **kwargs):
        logger.info('training finished')

    def on_epoch_begin(self, *args, **kwargs):
        if self.epoch_id > 0 and self.epoch_every and self.epoch_id % self.epoch_every == 0:
            logger.info('epoch {0} time {1}'.format(self.epoch_id - 1, str(datetime.now() - self.epoch_start)[:-7]))
        self.epoch_start, self.current_sum = datetime.now(), timedelta()

    def on_batch_begin(self, *args, **kwargs):
        if self.batch_id > 0:
            self.current_sum += datetime.now() - self.batch_start
            if self.batch_every and (self.batch_id - 1) % self.batch_every == 0:
                logger.info('epoch {0} average batch time: {1}'.format(self.epoch_id, str(self.current_sum / self.batch_id)[:-5]))
        self.batch_start = datetime.now()


def callbacks_unet(callbacks_config):
    """src/models.py:295-307 over the keys of src/pipeline_config.py:92-119 that are present (order as in the
    reference); `neptune_monitor` is skipped (no service), `plateau_lr_scheduler` is unused by the reference too"""
    cfg = callbacks_config or {}
    made = []
    if 'experiment_timing' in cfg:
        made.append(ExperimentTiming(**cfg['experiment_timing']))
    if 'training_monitor' in cfg:
        made.append(TrainingMonitor(**cfg['training_monitor']))
    if 'validation_monitor' in cfg:
        made.append(ValidationMonitor(**cfg['validation_monitor']))
    if 'model_checkpoint' in cfg and cfg['model_checkpoint'].get('filepath'):
        made.append(ModelCheckpoint(**cfg['model_checkpoint']))
    if 'exp_lr_scheduler' in cfg:
        made.append(ExponentialLRScheduler(**cfg['exp_lr_scheduler']))
    if 'early_stopping' in cfg:
        made.append(EarlyStopping(**cfg['early_stopping']))
    return CallbackList(made)
