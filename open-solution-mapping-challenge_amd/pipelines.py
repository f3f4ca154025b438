"""Pipeline graphs for the hot path: `unet`, `unet_weighted`, `unet_tta`, `unet_padded` and `mask_postprocessing`
(reference: src/pipelines.py:12-155, 248-304), built from the HIP transformers.

`config` has the shape of the reference's SOLUTION_CONFIG (src/pipeline_config.py:33-166): attribute or
key access to `env.cache_dirpath`, `execution.stream_mode`, `unet`, `postprocessor.mask_erosion /
mask_dilation`.  The loader Step is whatever the caller supplies (the reference's loaders are CPU image
decoding and out of scope); `synthetic_loader` feeds pre-built tensors for tests and benchmarks.

Two post-processing graphs are offered:
  mask_postprocessing        the reference's six Steps, each a per-image apply-transformer over a
                             reference-signature HIP function (exact API compatibility)
  mask_postprocessing_fused  ONE Step that keeps the batch on the device through all six stages
                             (postprocessing.postprocess_batch) and emits the same `images_with_scores`
"""
from functools import partial

import torch

from . import postprocessing as post
from .models import PyTorchUNet, PyTorchUNetStream, PyTorchUNetWeighted, PyTorchUNetWeightedStream
from .steps import BaseTransformer, Dummy, Step, make_apply_transformer


def _get(cfg, key):
    return cfg[key] if isinstance(cfg, dict) else getattr(cfg, key)


class SyntheticLoader(BaseTransformer):
    """Stand-in for the reference's Metadata*Loader.transform (src/loaders.py:192-204): hands back
    `{'datagen': (batches, steps), 'validation_datagen': ...}` built from in-memory tensors."""

    def __init__(self, batch_size):
        self.batch_size = batch_size

    def _gen(self, X, y):
        if X is None:
            return None
        n = X.shape[0]
        batches = []
        for i in range(0, n, self.batch_size):
            batches.append([X[i:i + self.batch_size]] if y is None else [X[i:i + self.batch_size], y[i:i + self.batch_size]])
        return batches, len(batches)

    def transform(self, X, y=None, X_valid=None, y_valid=None, train_mode=True, **kwargs):
        return {'datagen': self._gen(X, y if train_mode else None), 'validation_datagen': self._gen(X_valid, y_valid)}


def synthetic_loader(config, batch_size):
    return Step(name='loader', transformer=SyntheticLoader(batch_size), input_data=['input'],
                adapter={'X': ([('input', 'X')]), 'y': ([('input', 'y')]), 'train_mode': ([('input', 'train_mode')])},
                cache_dirpath=_get(_get(config, 'env'), 'cache_dirpath'))


class MaskPostprocessingHIP(BaseTransformer):
    """All of mask_postprocessing as one transformer; `images` may be a numpy array [N,2,h,w], a list of
    per-image arrays, or an iterable of cuda batches (PyTorchUNet.transform_device)."""

    def __init__(self, erode_selem_size=0, dilate_selem_size=0, batch_size=64, watershed_selem_size=0):
        self.erode, self.dilate, self.batch_size = erode_selem_size, dilate_selem_size, batch_size
        self.watershed = watershed_selem_size         # extension (WATERSHED.md); 0 = the reference's plain labelling

    def transform(self, images, target_sizes):
        out = []
        sizes = list(target_sizes)
        if isinstance(images, torch.Tensor) or hasattr(images, 'shape'):
            images = [images[i:i + self.batch_size] for i in range(0, len(images), self.batch_size)]
        pos = 0
        for batch in images:
            t = batch if isinstance(batch, torch.Tensor) else torch.as_tensor(batch)
            if t.dim() == 3:
                t = t[None]
            n = t.shape[0]
            if not sizes:
                out += post.postprocess_batch(t, None, self.erode, self.dilate, watershed_selem_size=self.watershed)
            else:
                # the reference resizes image by image (mask_resize zips images with target_sizes, src/pipelines.py:249-260):
                # one batched call per distinct target size, results put back in image order
                mine = [tuple(sz) for sz in sizes[pos:pos + n]]
                res = [None] * n
                for size in dict.fromkeys(mine):
                    idx = [i for i, sz in enumerate(mine) if sz == size]
                    sel = t if len(idx) == n else t[torch.as_tensor(idx, device=t.device)]
                    for i, r in zip(idx, post.postprocess_batch(sel, size, self.erode, self.dilate, watershed_selem_size=self.watershed)):
                        res[i] = r
                out += res
            pos += n
        return {'images_with_scores': out}


def mask_postprocessing(model, config, make_transformer=make_apply_transformer, **kwargs):
    """src/pipelines.py:248-304 with the HIP functions of postprocessing.py."""
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    pp = _get(config, 'postprocessor')
    mask_resize = Step(name='mask_resize',
                       transformer=make_transformer(post.resize_image, output_name='resized_images',
                                                    apply_on=['images', 'target_sizes']),
                       input_data=['input'], input_steps=[model],
                       adapter={'images': ([(model.name, 'multichannel_map_prediction')]),
                                'target_sizes': ([('input', 'target_sizes')])},
                       cache_dirpath=cache, cache_output=True, **kwargs)
    category_mapper = Step(name='category_mapper',
                           transformer=make_transformer(post.categorize_multilayer_image, output_name='categorized_images'),
                           input_steps=[mask_resize], adapter={'images': ([('mask_resize', 'resized_images')])},
                           cache_dirpath=cache, **kwargs)
    mask_erosion = Step(name='mask_erosion',
                        transformer=make_transformer(partial(_erode_layers, **dict(_get(pp, 'mask_erosion'))),
                                                     output_name='eroded_images'),
                        input_steps=[category_mapper], adapter={'images': ([(category_mapper.name, 'categorized_images')])},
                        cache_dirpath=cache, **kwargs)
    labeler = Step(name='labeler',
                   transformer=make_transformer(post.label_multilayer_image, output_name='labeled_images'),
                   input_steps=[mask_erosion], adapter={'images': ([(mask_erosion.name, 'eroded_images')])},
                   cache_dirpath=cache, **kwargs)
    mask_dilation = Step(name='mask_dilation',
                         transformer=make_transformer(partial(post.dilate_image, **dict(_get(pp, 'mask_dilation'))),
                                                      output_name='dilated_images'),
                         input_steps=[labeler], adapter={'images': ([(labeler.name, 'labeled_images')])},
                         cache_dirpath=cache, **kwargs)
    score_builder = Step(name='score_builder',
                         transformer=make_transformer(post.build_score, output_name='images_with_scores',
                                                      apply_on=['images', 'probabilities']),
                         input_steps=[mask_dilation, mask_resize],
                         adapter={'images': ([(mask_dilation.name, 'dilated_images')]),
                                  'probabilities': ([(mask_resize.name, 'resized_images')])},
                         cache_dirpath=cache, **kwargs)
    return score_builder


def _erode_layers(mask, erode_selem_size):
    # the shipped configuration has erode_selem_size 0 (neptune.yaml:69): passthrough, like the reference
    return post.erode_image(mask, erode_selem_size)


def mask_postprocessing_fused(model, config, **kwargs):
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    pp = _get(config, 'postprocessor')
    tr = MaskPostprocessingHIP(erode_selem_size=dict(_get(pp, 'mask_erosion')).get('erode_selem_size', 0),
                               dilate_selem_size=dict(_get(pp, 'mask_dilation')).get('dilate_selem_size', 0),
                               # extension, off unless configured (postprocessor.watershed.marker_erosion, WATERSHED.md)
                               watershed_selem_size=dict(pp.get('watershed', {}) if hasattr(pp, 'get') else {}).get('marker_erosion', 0))
    return Step(name='score_builder', transformer=tr, input_data=['input'], input_steps=[model],
                adapter={'images': ([(model.name, 'multichannel_map_prediction')]),
                         'target_sizes': ([('input', 'target_sizes')])},
                cache_dirpath=cache, **kwargs)


def unet(config, train_mode, loader=None, fused_postprocessing=False, weighted=False):
    """src/pipelines.py:12-38 (and :41-52 with weighted=True)."""
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    stream = bool(_get(_get(config, 'execution'), 'stream_mode'))
    unet_cfg = dict(_get(config, 'unet'))
    if loader is None:
        loader = synthetic_loader(config, _get(_get(config, 'execution'), 'batch_size_train'))
    cls = {(False, False): PyTorchUNet, (False, True): PyTorchUNetStream,
           (True, False): PyTorchUNetWeighted, (True, True): PyTorchUNetWeightedStream}[(weighted, stream)]
    unet_step = Step(name='unet', transformer=cls(**unet_cfg), input_data=['callback_input'], input_steps=[loader],
                     cache_dirpath=cache, is_trainable=True)
    post_step = (mask_postprocessing_fused if fused_postprocessing else mask_postprocessing)(unet_step, config)
    return Step(name='output', transformer=Dummy(), input_steps=[post_step],
                adapter={'y_pred': ([(post_step.name, 'images_with_scores')])}, cache_dirpath=cache)


def unet_weighted(config, train_mode, loader=None, fused_postprocessing=False):
    return unet(config, train_mode, loader=loader, fused_postprocessing=fused_postprocessing, weighted=True)


class UNetTTA(BaseTransformer):
    """The three Steps tta_generator -> unet -> tta_aggregator of the reference's `unet_tta` (src/pipelines.py:94-116,
    src/loaders.py:401-517) as ONE transformer around a PyTorchUNet: every batch is expanded to its flip / rot90 variants,
    predicted, un-transformed and aggregated on the device.  Registered under the Step name 'unet', so the weights a
    `unet` training run persisted are the ones it loads."""

    def __init__(self, unet_config, tta_generator, tta_aggregator):
        self.unet = PyTorchUNet(**unet_config)
        self.tta = dict(tta_generator)
        self.method = dict(tta_aggregator).get('method', 'gmean')

    def fit(self, *args, **kwargs):
        raise NotImplementedError('unet_tta is an inference pipeline (src/pipelines.py:395-401)')

    def transform(self, datagen, validation_datagen=None, *args, **kwargs):
        return self.unet.transform_tta(datagen, self.tta, self.method)

    def load(self, filepath):
        self.unet.load(filepath)
        return self

    def save(self, filepath):
        self.unet.save(filepath)


def _prediction_crop(model, config):
    """src/pipelines.py:68-83 / 118-133: centre crop of the predictions (loader_mode 'crop_and_pad') + rename"""
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    crop = Step(name='prediction_crop',
                transformer=make_apply_transformer(partial(post.crop_image_center_per_class,
                                                           **dict(_get(_get(config, 'postprocessor'), 'prediction_crop'))),
                                                   output_name='cropped_images'),
                input_steps=[model], adapter={'images': ([(model.name, 'multichannel_map_prediction')])}, cache_dirpath=cache)
    return Step(name='prediction_renamed', transformer=Dummy(), input_steps=[crop],
                adapter={'multichannel_map_prediction': ([(crop.name, 'cropped_images')])}, cache_dirpath=cache)


def _inference_tail(model, config, fused_postprocessing):
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    mode = _get(_get(config, 'execution'), 'loader_mode') if _has(_get(config, 'execution'), 'loader_mode') else 'resize'
    if mode == 'crop_and_pad':
        model = _prediction_crop(model, config)
    elif mode != 'resize':
        raise NotImplementedError('only crop_and_pad and resize options available')      # src/pipelines.py:145
    post_step = (mask_postprocessing_fused if fused_postprocessing else mask_postprocessing)(model, config)
    return Step(name='output', transformer=Dummy(), input_steps=[post_step],
                adapter={'y_pred': ([(post_step.name, 'images_with_scores')])}, cache_dirpath=cache)


def _has(cfg, key):
    return key in cfg if isinstance(cfg, dict) else hasattr(cfg, key)


def unet_tta(config, loader=None, fused_postprocessing=False):
    """src/pipelines.py:94-155"""
    if bool(_get(_get(config, 'execution'), 'stream_mode')):
        raise Exception('TTA not available in stream mode')                                # src/pipelines.py:95-96
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    if loader is None:
        loader = synthetic_loader(config, _get(_get(config, 'execution'), 'batch_size_inference')
                                  if _has(_get(config, 'execution'), 'batch_size_inference') else 32)
    step = Step(name='unet', transformer=UNetTTA(dict(_get(config, 'unet')), _get(config, 'tta_generator'), _get(config, 'tta_aggregator')),
                input_data=['callback_input'], input_steps=[loader], cache_dirpath=cache, is_trainable=True)
    return _inference_tail(step, config, fused_postprocessing)


def unet_padded(config, loader=None, fused_postprocessing=False):
    """src/pipelines.py:55-91: plain inference whose predictions are centre-cropped (loader_mode 'crop_and_pad')"""
    cache = _get(_get(config, 'env'), 'cache_dirpath')
    stream = bool(_get(_get(config, 'execution'), 'stream_mode'))
    if loader is None:
        loader = synthetic_loader(config, 32)
    step = Step(name='unet', transformer=(PyTorchUNetStream if stream else PyTorchUNet)(**dict(_get(config, 'unet'))),
                input_data=['callback_input'], input_steps=[loader], cache_dirpath=cache, is_trainable=True)
    post_step = (mask_postprocessing_fused if fused_postprocessing else mask_postprocessing)(_prediction_crop(step, config), config)
    return Step(name='output', transformer=Dummy(), input_steps=[post_step],
                adapter={'y_pred': ([(post_step.name, 'images_with_scores')])}, cache_dirpath=cache)


class OverlappedAnnotator:
    """Inference -> post-processing -> annotations with the two halves on two HIP streams (BASELINE.json configs[3] end to end).

    The tail of the reference's inference pipeline (mask_postprocessing, src/pipelines.py:248-304, then create_annotations,
    src/utils.py:76-115) is latency-bound on the device -- ~60 small launches and four host synchronisations per call -- while the
    network keeps every CU busy and needs no host attention once enqueued.  On one stream the two alternate (round 3: 5.9 k img/s end
    to end from a 12.8 k img/s network and a 12.7 k img/s tail).  Here the network batches of group g+1 are ENQUEUED first, on their
    own stream, into the other half of a double-buffered probability buffer; then the host drives the tail of group g on the tail
    stream (whose synchronisations wait for that stream only) while the GPU works through the queued forward passes.  An event
    hands each filled buffer over; a buffer is written again only after the tail that read it has returned (the tail ends with its
    results on the host).

    annotate(groups): `groups` yields (image_ids, [x batches: cuda f32 [N,3,H,W]], rgb or None) -- rgb cuda u8 [sum N,H,W,3] switches the
    dense CRF on; yields one JSON document (bytes; `as_list=True`: the list of annotation dicts) per group, in order."""

    def __init__(self, net, category_ids, category_layers, target_size=None, erode_selem_size=0, dilate_selem_size=0,
                 watershed_selem_size=0, crf_params=None, as_list=False):
        self.net = net
        self.kw = dict(category_ids=category_ids, category_layers=category_layers, target_size=target_size, erode_selem_size=erode_selem_size,
                       dilate_selem_size=dilate_selem_size, watershed_selem_size=watershed_selem_size, crf_params=crf_params)
        self.as_list = as_list
        self._bufs, self._streams = {}, None

    def _buf(self, slot, n, h, w, dev):
        key = (slot, n, h, w)
        if key not in self._bufs:
            self._bufs[key] = torch.empty((n, 2, h, w), dtype=torch.float32, device=dev)
        return self._bufs[key]

    def _tail(self, ids, probs, rgb):
        from . import utils
        fn = utils.annotations_from_probabilities if self.as_list else utils.annotations_json_from_probabilities
        return fn(ids, probs, crf_images=rgb, **self.kw)

    def annotate(self, groups):
        pending = None                    # (image ids, probability buffer, rgb, event: buffer filled)
        for slot, (ids, batches, rgb) in enumerate(groups):
            dev = batches[0].device
            if self._streams is None:
                self._streams = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
            s_net, s_tail = self._streams
            n = sum(int(b.shape[0]) for b in batches)
            buf = self._buf(slot & 1, n, batches[0].shape[2], batches[0].shape[3], dev)
            s_net.wait_stream(torch.cuda.current_stream(dev))          # the caller produced the inputs on its stream
            with torch.cuda.stream(s_net):
                at = 0
                for x in batches:
                    buf[at:at + x.shape[0]].copy_(self.net.predict_proba(x))
                    at += x.shape[0]
                filled = torch.cuda.Event()
                filled.record(s_net)
            if pending is not None:
                yield self._run_tail(pending, s_tail)
            pending = (ids, buf, rgb, filled)
        if pending is not None:
            yield self._run_tail(pending, self._streams[1])

    def _run_tail(self, pending, s_tail):
        ids, buf, rgb, filled = pending
        with torch.cuda.stream(s_tail):
            s_tail.wait_event(filled)
            return self._tail(ids, buf, rgb)


# the scoring-model pipelines (src/pipelines.py:307-392: LightGBM/random-forest second level) are outside the hot path
PIPELINES = {'unet': {'train': partial(unet, train_mode=True), 'inference': partial(unet, train_mode=False)},
             'unet_weighted': {'train': partial(unet_weighted, train_mode=True),
                               'inference': partial(unet_weighted, train_mode=False)},
             'unet_tta': {'inference': unet_tta},
             'unet_padded': {'inference': unet_padded}}
