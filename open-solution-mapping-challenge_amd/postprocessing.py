"""Mask post-processing on MI355X: the functions of the reference's `src/postprocessing.py:48-258`
(+ `label`, `add_dropped_objects`, `softmax` from `src/utils.py`) with the same names, argument
meaning and return types, executed by the HIP kernels behind include/msc.h.

Two levels:
  * per-image functions with the reference signatures (numpy in, numpy out) -- what
    `make_apply_transformer(func, ...)` wraps in `mask_postprocessing` (src/pipelines.py:248-304);
    each call is one H2D copy, a few launches and one D2H copy;
  * `postprocess_batch`: the whole chain (resize -> threshold -> [erode] -> label -> dilate -> score)
    for a batch of probability maps that is ALREADY on the device, one launch per stage for the whole
    batch and a single D2H at the end -- this is what keeps post-processing off the critical path.

Integer outputs are bit-identical to the reference (via oracle/post_ref.py); float outputs
(resize, scores, CRF) agree within the tolerances stated in tests/.  There is no CPU fallback.
"""

import numpy as np
import torch

from . import _lib

CATEGORY_LAYERS = [1, 1]      # src/pipeline_config.py:18
MEAN = [0.485, 0.456, 0.406]  # src/pipeline_config.py:19-20
STD = [0.229, 0.224, 0.225]


def _device():
    if not torch.cuda.is_available():
        raise _lib.MscError('mapping_challenge_amd.postprocessing needs a ROCm GPU: the product has no CPU path')
    return torch.device('cuda')


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(_device(), non_blocking=False)


def layer_table(category_layers=CATEGORY_LAYERS, channels=None):
    """(class index, threshold) per layer: thresholds arange(1/(L+1), 1, 1/(L+1)) for each class (:80-83).  `channels`: the map's channel
    count -- the reference walks the CHANNELS and looks the layer count up (:79-80), so classes beyond the last channel are not cut and a
    table shorter than the map raises its IndexError (the kernels index the map by class: a class >= C would read out of bounds)"""
    if channels is not None:
        if len(category_layers) < channels:
            raise IndexError('list index out of range')        # CATEGORY_LAYERS[category_id], src/postprocessing.py:80
        category_layers = category_layers[:channels]
    cls, thr = [], []
    for cat, n in enumerate(category_layers):
        step = 1. / (n + 1)
        for t in np.arange(step, 1, step):
            cls.append(cat)
            thr.append(t)               # the reference's float64 thresholds; the kernels compare in double (numpy: array > float64 scalar)
    return np.asarray(cls, np.int32), np.asarray(thr, np.float64)


# ------------------------------------------------------------------ device-level (batched) primitives
def resize_batch(probs, target_size, dtype=torch.float32):
    """probs: cuda f32 [B,C,h,w] -> cuda [B,C,H,W]: the double interpolant the reference's skimage resize returns, stored as float64
    (`dtype=torch.float64`) or rounded once to float32"""
    probs = probs.contiguous().float()
    B, Cc, h, w = probs.shape
    H, W = target_size
    out = torch.empty((B, Cc, H, W), dtype=dtype, device=probs.device)
    ws = torch.empty(2 * B, dtype=torch.float32, device=probs.device)
    _lib.call('msc_resize_bilinear', probs.data_ptr(), out.data_ptr(), int(dtype == torch.float64), ws.data_ptr(), B, Cc, h, w, H, W, _stream())
    return out


def resize_threshold_batch(probs, target_size, category_layers=CATEGORY_LAYERS):
    """resize_image + categorize_multilayer_image in one launch: cuda f32 [B,C,h,w] -> (cuda f32 [B,C,H,W] for the scoring, cuda u8
    [B,L,H,W]); the layers are cut from the DOUBLE interpolant, as the reference cuts them from skimage's float64 map
    (src/postprocessing.py:60,83), so they are bit-identical to the reference chain's and not a thresholding of the rounded map"""
    probs = probs.contiguous().float()
    B, Cc, h, w = probs.shape
    H, W = target_size
    cls, thr = layer_table(category_layers, Cc)
    L = len(cls)
    dcls, dthr = torch.from_numpy(cls).to(probs.device), torch.from_numpy(thr).to(probs.device)
    out = torch.empty((B, Cc, H, W), dtype=torch.float32, device=probs.device)
    layers = torch.empty((B, L, H, W), dtype=torch.uint8, device=probs.device)
    ws = torch.empty(2 * B, dtype=torch.float32, device=probs.device)
    _lib.call('msc_resize_threshold', probs.data_ptr(), out.data_ptr(), layers.data_ptr(), ws.data_ptr(), B, Cc, h, w, H, W,
              dcls.data_ptr(), dthr.data_ptr(), L, _stream())
    return out, layers


def crop_batch(images, h_crop, w_crop):
    B, Cc, h, w = images.shape
    out = torch.empty((B, Cc, h_crop, w_crop), dtype=torch.float32, device=images.device)
    _lib.call('msc_crop_center', images.data_ptr(), out.data_ptr(), B, Cc, h, w, h_crop, w_crop, _stream())
    return out


def threshold_batch(probs, category_layers=CATEGORY_LAYERS):
    """cuda f32 or f64 [B,C,H,W] -> cuda u8 [B,L,H,W]"""
    if probs.dtype not in (torch.float32, torch.float64):
        probs = probs.float()
    probs = probs.contiguous()          # a numpy map built by fancy indexing arrives with permuted strides
    B, Cc, H, W = probs.shape
    cls, thr = layer_table(category_layers, Cc)
    L = len(cls)
    dcls, dthr = torch.from_numpy(cls).to(probs.device), torch.from_numpy(thr).to(probs.device)
    out = torch.empty((B, L, H, W), dtype=torch.uint8, device=probs.device)
    _lib.call('msc_threshold_layers', probs.data_ptr(), int(probs.dtype == torch.float64), out.data_ptr(), B, Cc, H, W, dcls.data_ptr(), dthr.data_ptr(), L, _stream())
    return out


def label_batch(masks):
    """cuda u8 [B,H,W] -> (cuda i32 labels [B,H,W], cuda i32 counts [B])"""
    B, H, W = masks.shape
    lib = _lib.load()
    labels = torch.empty((B, H, W), dtype=torch.int32, device=masks.device)
    counts = torch.empty((B,), dtype=torch.int32, device=masks.device)
    ws = torch.empty((max(1, lib.msc_label_workspace_bytes(B, H, W)),), dtype=torch.uint8, device=masks.device)
    _lib.call('msc_label4', masks.data_ptr(), labels.data_ptr(), counts.data_ptr(), ws.data_ptr(), B, H, W, _stream())
    return labels, counts


def erode_batch(masks, k):
    B, H, W = masks.shape
    out = torch.empty_like(masks)
    _lib.call('msc_erode_u8', masks.data_ptr(), out.data_ptr(), B, H, W, int(k), _stream())
    return out


def dilate_batch(labels, k):
    B, H, W = labels.shape
    out = torch.empty_like(labels)
    _lib.call('msc_dilate_i32', labels.data_ptr(), out.data_ptr(), B, H, W, int(k), _stream())
    return out


def add_dropped_batch(original, processed, bool_sum=True):
    """cuda u8 [B,H,W] x2 -> cuda u8 [B,H,W] (src/utils.py:333-339); bool_sum: the masks are bool in the reference (the
    layers of categorize_*), where `reconstructed += component` is a logical or"""
    B, H, W = original.shape
    lab, _ = label_batch(original)
    out = torch.empty_like(processed)
    ws = torch.empty((B * H * W,), dtype=torch.int32, device=original.device)
    _lib.call('msc_add_dropped', processed.data_ptr(), lab.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W, int(bool_sum), _stream())
    return out


def watershed_batch(layers, probs, marker_erosion):
    """EXTENSION (WATERSHED.md; the reference has no watershed).  layers cuda u8 [B,H,W], probs cuda f32 [B,H,W] (the
    channel each layer was thresholded from) -> (cuda i32 labels [B,H,W], cuda i32 counts [B]): markers = components of
    erode_image(layer, k), flooded over the 8-bit relief of 1 - P inside the layer."""
    if not marker_erosion > 0:
        raise ValueError('watershed needs marker_erosion >= 1')
    B, H, W = layers.shape
    lib = _lib.load()
    markers_mask = add_dropped_batch(layers, erode_batch(layers, marker_erosion))
    labels, counts = label_batch(markers_mask)
    ws = torch.empty((max(4, lib.msc_watershed_workspace_bytes(B, H, W)),), dtype=torch.uint8, device=layers.device)
    _lib.call('msc_watershed', probs.contiguous().data_ptr(), layers.data_ptr(), labels.data_ptr(), ws.data_ptr(), B, H, W, _stream())
    return labels, counts


def score_batch(labels, probs, max_labels):
    """labels cuda i32 [B,H,W], probs cuda f32 [B,H,W] -> cuda f64 [B,max_labels] scores"""
    B, H, W = labels.shape
    max_labels = max(1, int(max_labels))
    sums = torch.empty((B, max_labels), dtype=torch.float64, device=labels.device)
    areas = torch.empty((B, max_labels), dtype=torch.int32, device=labels.device)
    score = torch.empty((B, max_labels), dtype=torch.float64, device=labels.device)
    _lib.call('msc_build_score', labels.data_ptr(), probs.data_ptr(), sums.data_ptr(), areas.data_ptr(), score.data_ptr(),
              B, H, W, max_labels, _stream())
    return score


def _to_host(t):
    """D2H through a pinned buffer (torch caches pinned blocks): pageable copies run at a fraction of the PCIe rate and the
    label images are the bulk of what this chain returns (46 MB per 64 images)"""
    if not t.is_cuda:                         # already a host tensor (the CPU interpreter of the tests)
        return t.detach().numpy().copy()
    h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    h.copy_(t, non_blocking=True)
    torch.cuda.current_stream(t.device).synchronize()
    return h.numpy()


def _to_host_many(tensors):
    """several D2H copies through pinned buffers behind one stream synchronisation"""
    if not tensors[0].is_cuda:
        return [t.detach().numpy().copy() for t in tensors]
    hs = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        hs.append(h)
    torch.cuda.current_stream(tensors[0].device).synchronize()
    return [h.numpy() for h in hs]


LABEL_CAPACITY = 512          # instances per image layer the score table is sized for before the counts are known (more: one re-score)


def postprocess_device(probs, target_size=None, erode_selem_size=0, dilate_selem_size=0, category_layers=CATEGORY_LAYERS,
                       watershed_selem_size=0, raw_scores=False):
    """postprocess_batch without the final copy of the label images: returns (labels cuda i32 [B,L,H,W], per-image
    per-layer score lists) -- for consumers that stay on the device (utils.annotations_from_probabilities).
    raw_scores: (labels, counts i32 [B,L], scores f64 [B,n_scored,cap]) as numpy arrays instead of the nested lists."""
    return _postprocess(probs, target_size, erode_selem_size, dilate_selem_size, category_layers, False, watershed_selem_size, raw_scores)


def postprocess_batch(probs, target_size=None, erode_selem_size=0, dilate_selem_size=0, category_layers=CATEGORY_LAYERS,
                      watershed_selem_size=0):
    """The six Steps of `mask_postprocessing` (src/pipelines.py:248-304) for a whole batch on the device.

    probs: cuda f32 [B,2,h,w] softmax maps.  Returns the reference's `images_with_scores` list:
    [(labels i32 [L,H,W], [[score, ...] per layer]), ...] (numpy / python floats, one D2H at the end).
    watershed_selem_size > 0 (extension, WATERSHED.md): the labelling step becomes a marker-controlled watershed.
    """
    return _postprocess(probs, target_size, erode_selem_size, dilate_selem_size, category_layers, True, watershed_selem_size)


def _layer_classes(category_layers):
    return [c for c, n in enumerate(category_layers) for _ in range(n)]


def _postprocess(probs, target_size, erode_selem_size, dilate_selem_size, category_layers, to_host, watershed_selem_size=0, raw_scores=False):
    if not probs.is_cuda:
        probs = probs.to(_device())
    probs = probs.contiguous().float()
    if target_size is not None:
        p, layers = resize_threshold_batch(probs, target_size, category_layers)     # layers [B,L,H,W] u8 from the double interpolant
    else:
        p, layers = probs, threshold_batch(probs, category_layers)
    B, Cc, H, W = p.shape
    L = layers.shape[1]
    flat = layers.view(B * L, H, W)
    if erode_selem_size > 0:
        flat = add_dropped_batch(flat, erode_batch(flat, erode_selem_size))
    if watershed_selem_size > 0:      # extension (WATERSHED.md): instances split along the probability ridges instead of plain labelling
        cls = torch.as_tensor(_layer_classes(category_layers[:Cc]), device=p.device)
        labels, counts = watershed_batch(flat, p.index_select(1, cls).reshape(B * L, H, W), watershed_selem_size)
    else:
        labels, counts = label_batch(flat)
    if dilate_selem_size > 0:
        labels = dilate_batch(labels, dilate_selem_size)
    # build_score zips layer l with probability channel l (src/postprocessing.py:230)
    n_scored = min(L, Cc)
    lab4 = labels.view(B, L, H, W)
    sl = lab4[:, :n_scored].contiguous().view(B * n_scored, H, W)
    sp = p[:, :n_scored].contiguous().view(B * n_scored, H, W)
    # The score table needs the largest label count, which lives on the device: score with a capacity guess straight away and
    # fetch counts, scores (and labels) with ONE synchronisation; only a layer with more components than the guess costs a
    # second scoring pass (the chain used to stop twice for the host: counts, then scores).
    cap = LABEL_CAPACITY
    scores_d = score_batch(sl, sp, cap)
    got = _to_host_many([counts, scores_d] + ([lab4] if to_host else []))
    counts_h = got[0].reshape(B, L)
    max_labels = int(counts_h.max()) if counts_h.size else 0
    scores_h = got[1].reshape(B, n_scored, cap)
    if max_labels > cap:
        scores_h = score_batch(sl, sp, max_labels).cpu().numpy().reshape(B, n_scored, max_labels)
    if raw_scores and not to_host:
        return lab4, counts_h, scores_h
    scores = []
    for b in range(B):
        total = []
        for l in range(n_scored):
            n = int(counts_h[b, l])
            total.append(scores_h[b, l, :n].tolist() if n else [])
        scores.append(total)
    if not to_host:
        return lab4, scores
    labels_h = got[2]
    return [(labels_h[b], scores[b]) for b in range(B)]


# ------------------------------------------------------------------ reference-signature functions
def softmax(X, theta=1.0, axis=None):
    """src/utils.py:231-273.  Host helper kept for API parity; the HIP model path fuses softmax into the last
    conv (msc_final_fwd), so this is only used on arrays that never were on the device."""
    y = np.atleast_2d(X)
    if axis is None:
        axis = next(j[0] for j in enumerate(y.shape) if j[1] > 1)
    y = y * float(theta)
    y = y - np.expand_dims(np.max(y, axis=axis), axis)
    y = np.exp(y)
    p = y / np.expand_dims(np.sum(y, axis=axis), axis)
    if len(X.shape) == 1:
        p = p.flatten()
    return p


def resize_image(image, target_size):
    """src/postprocessing.py:48-61.  image (C x H x W) -> (C x h x w), float64 like the reference's skimage resize (the network's
    probabilities are float32; a float64 input is rounded to float32 first)."""
    d = _dev(image, np.float32)[None]
    return resize_batch(d, tuple(target_size), torch.float64)[0].cpu().numpy()


def categorize_image(image):
    """src/postprocessing.py:64-74: argmax over channels."""
    d = _dev(image, np.float32)
    Cc, H, W = d.shape
    out = torch.empty((H, W), dtype=torch.int32, device=d.device)
    _lib.call('msc_argmax_channels', d.data_ptr(), out.data_ptr(), 1, Cc, H, W, _stream())
    return out.cpu().numpy().astype(np.int64)


def categorize_multilayer_image(image):
    """src/postprocessing.py:77-84 -> bool (L x H x W).  A float64 map (what resize_image returns) is thresholded as float64."""
    image = np.asarray(image)
    d = _dev(image, np.float64 if image.dtype == np.float64 else np.float32)[None]
    return threshold_batch(d)[0].cpu().numpy().astype(bool)


def label(mask):
    """src/utils.py:328-330 (scipy.ndimage.label, 4-connectivity, raster-order numbering), int32."""
    d = _dev(np.asarray(mask) != 0, np.uint8)[None]
    return label_batch(d)[0][0].cpu().numpy()


def label_multilayer_image(mask):
    """src/postprocessing.py:127-132."""
    d = _dev(np.asarray(mask) != 0, np.uint8)
    return label_batch(d)[0].cpu().numpy()


def label_multiclass_image(mask):
    """src/postprocessing.py:87-124."""
    mask = np.asarray(mask)
    planes = np.stack([(mask == c) for c in range(0, int(mask.max()) + 1)])
    return label_multilayer_image(planes)


def watershed_multilayer_image(image, probabilities, marker_erosion, category_layers=CATEGORY_LAYERS):
    """EXTENSION (WATERSHED.md): layers bool [L,H,W] as categorize_multilayer_image orders them + the probability map
    f [C,H,W] they were cut from -> int32 [L,H,W] instance labels split along the probability ridges."""
    image = np.asarray(image)
    pr = np.asarray(probabilities, np.float32)
    cls = _layer_classes(category_layers)[:len(image)]
    layers = _dev(image != 0, np.uint8)
    labels, _ = watershed_batch(layers, _dev(np.stack([pr[c] for c in cls]), np.float32), marker_erosion)
    return labels.cpu().numpy()


def add_dropped_objects(original, processed):
    """src/utils.py:333-339."""
    o = _dev(np.asarray(original) != 0, np.uint8)[None]
    p = _dev(processed, np.uint8)[None]
    return add_dropped_batch(o, p, np.asarray(processed).dtype == bool)[0].cpu().numpy()


def erode_image(mask, erode_selem_size):
    """src/postprocessing.py:135-156."""
    if not erode_selem_size > 0:
        return mask
    mask = np.asarray(mask)
    if mask.ndim != 2:
        # the reference's 3-D branch raises for >= 2 layers (np.stack inside the loop, :153-155)
        raise ValueError('erode_image: only 2-D masks are defined behaviour in the reference (src/postprocessing.py:153-155)')
    d = _dev(mask, np.uint8)[None]
    er = erode_batch(d, erode_selem_size)
    return add_dropped_batch(d, er, mask.dtype == bool)[0].cpu().numpy()


def dilate_image(mask, dilate_selem_size):
    """src/postprocessing.py:159-180: grey dilation of the label image (larger label wins)."""
    if not dilate_selem_size > 0:
        return mask
    mask = np.asarray(mask)
    d = _dev(mask if mask.ndim == 3 else mask[None], np.int32)
    out = dilate_batch(d, dilate_selem_size).cpu().numpy().astype(mask.dtype, copy=False)
    return out if mask.ndim == 3 else out[0]


def build_score(image, probabilities):
    """src/postprocessing.py:228-236."""
    image = np.asarray(image)
    n = min(len(image), len(probabilities))
    lab = _dev(image[:n], np.int32)
    pr = _dev(np.asarray(probabilities)[:n], np.float32)
    mx = int(image[:n].max()) if n else 0
    total = []
    if mx > 0:
        sc = score_batch(lab, pr, mx).cpu().numpy()
        for l in range(n):
            total.append([float(v) for v in sc[l, :int(image[l].max())]])
    else:
        total = [[] for _ in range(n)]
    return image, total


def crop_image_center_per_class(image, h_crop, w_crop):
    """src/postprocessing.py:239-258."""
    d = _dev(image, np.float32)[None]
    return crop_batch(d, h_crop, w_crop)[0].cpu().numpy()


def dense_crf(img, output_probs, compat_gaussian=3, sxy_gaussian=1, compat_bilateral=10, sxy_bilateral=1, srgb=50,
              iterations=5):
    """src/postprocessing.py:183-225 -- exact windowed mean field (see oracle/crf_ref.py; parity with pydensecrf's
    permutohedral approximation is unpinned).  img: normalised RGB (3 x H x W), output_probs (2 x H x W)."""
    probs = np.asarray(output_probs, np.float32)
    if probs.shape[0] != 2:
        raise NotImplementedError('dense_crf: 2 classes (the reference hard-codes DenseCRF2D(width, height, 2))')
    org = np.asarray(img) * np.array(STD).reshape(3, 1, 1) + np.array(MEAN).reshape(3, 1, 1)
    org = np.ascontiguousarray((org * 255.).transpose(1, 2, 0), dtype=np.uint8)
    return dense_crf_batch(_dev(probs, np.float32)[None], _dev(org, np.uint8)[None], compat_gaussian, sxy_gaussian,
                           compat_bilateral, sxy_bilateral, srgb, iterations)[0].cpu().numpy()


def dense_crf_batch(probs, rgb, compat_gaussian=3, sxy_gaussian=1, compat_bilateral=10, sxy_bilateral=1, srgb=50,
                    iterations=5):
    """probs cuda f32 [B,2,H,W], rgb cuda u8 [B,H,W,3] -> cuda f32 [B,2,H,W]"""
    B, _, H, W = probs.shape
    lib = _lib.load()
    out = torch.empty_like(probs)
    ws = torch.empty((lib.msc_crf_workspace_bytes(B, H, W, 0),), dtype=torch.uint8, device=probs.device)
    _lib.call('msc_dense_crf', probs.data_ptr(), rgb.data_ptr(), out.data_ptr(), ws.data_ptr(), B, H, W,
              float(sxy_gaussian), float(compat_gaussian), float(sxy_bilateral), float(srgb), float(compat_bilateral),
              int(iterations), _stream())
    return out
