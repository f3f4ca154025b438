"""Annotation encoding after the hot path: drop-ins for the reference's `rle_from_binary`,
`bounding_box_from_rle` and `create_annotations` (`src/utils.py:61-127`; call sites
`src/pipeline_manager.py:207,226`, `src/callbacks.py:199`).

The reference copies the full label image once per instance (`decompose`) and hands each copy to
pycocotools (`cocomask.encode`, `cocomask.toBbox`; pycocotools==2.0.0, `environment.yml:27`).  Here all
instances of all layers of a chunk of images are encoded by one pass of HIP kernels over the label images
(`csrc/annot.hip`: column-major runs -> stable sort by instance -> COCO count strings + boxes); the host
only slices the resulting byte buffer into the annotation dicts.
"""
import ctypes as C
import json
import os

import numpy as np
import torch

from . import _lib


def _device():
    if not torch.cuda.is_available():
        raise _lib.MscError('annotation encoding needs a ROCm GPU: the product has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def _count_chars(x):
    """one count of a COCO RLE string (maskApi.c rleToString): only used for the single-count string of an EMPTY mask"""
    out = bytearray()
    more = True
    while more:
        c = x & 0x1f
        x >>= 5
        more = (x != -1) if (c & 0x10) else (x != 0)
        out.append((c | 0x20 if more else c) + 48)
    return bytes(out)


def encode_labels(labels, trusted=False, raw=False):
    """labels: int32 [L,H,W] (numpy or cuda tensor), 0 = background.  Returns a list of L dicts
    {instance id: (counts bytes, [x, y, w, h])} holding every id that owns at least one pixel.
    trusted: the label images come from msc_label4 / msc_dilate_i32 / msc_watershed (ids in [0, H*W]): skip the range check
    (two reductions and two host synchronisations).  raw: (table i32 [n,8] = layer, label, string begin / end, xs, ys, xe, ye sorted by
    (layer, label); the concatenated count strings) as the encoder returns them, for msc_annotations_json."""
    lib = _lib.load()
    dev = _device()
    t = torch.as_tensor(labels)
    if t.dim() != 3:
        raise ValueError('labels must be [layers, H, W]')
    t = t.to(device=dev, dtype=torch.int32).contiguous()
    L, H, W = t.shape
    out = [dict() for _ in range(L)]
    if L == 0:
        return (np.zeros((0, 8), np.int32), b'') if raw else out
    if not (trusted and H * W < (1 << 24)):
        lo, hi = int(t.min().item()), int(t.max().item())
        if lo < 0 or hi >= (1 << 24):          # the run sort key packs (layer << 24) | label
            raise ValueError('encode_labels: instance ids must lie in [0, 2^24) (got %d..%d)' % (lo, hi))
    stream = torch.cuda.current_stream(dev).cuda_stream
    b1 = lib.msc_rle_segments_workspace(L, H, W)
    if b1 < 0:
        _lib.check(-1, 'msc_rle_segments_workspace')
    ws1 = torch.empty(b1, dtype=torch.uint8, device=dev)
    nseg = C.c_int32(0)
    _lib.check(lib.msc_rle_segments(t.data_ptr(), L, H, W, ws1.data_ptr(), b1, C.byref(nseg), stream), 'msc_rle_segments')
    if nseg.value == 0:
        return (np.zeros((0, 8), np.int32), b'') if raw else out
    b2 = lib.msc_rle_encode_workspace(nseg.value)
    ws2 = torch.empty(b2, dtype=torch.uint8, device=dev)
    n_inst, n_chars = C.c_int32(0), C.c_int64(0)
    table_p, chars_p = C.c_void_p(), C.c_void_p()
    _lib.check(lib.msc_rle_encode(ws1.data_ptr(), L, H, W, nseg.value, ws2.data_ptr(), b2, C.byref(n_inst), C.byref(n_chars),
                                  C.byref(table_p), C.byref(chars_p), stream), 'msc_rle_encode')
    t0, c0 = table_p.value - ws2.data_ptr(), chars_p.value - ws2.data_ptr()
    table = ws2[t0:t0 + n_inst.value * 32].cpu().numpy().view(np.int32).reshape(-1, 8)
    chars = ws2[c0:c0 + n_chars.value].cpu().numpy().tobytes()
    if raw:
        return table, chars
    for layer, label, s0, s1, xs, ys, xe, ye in table.tolist():
        out[layer][label] = (chars[s0:s1], [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)])
    return out


def _instances(encoded, size):
    """per-instance (counts, bbox) in the order of the reference's decompose(): ids 1..max, ids without pixels as empty
    masks; a single empty mask when the layer holds no instance (src/utils.py:61-73)"""
    empty = (_count_chars(size[0] * size[1]), [0.0, 0.0, 0.0, 0.0])
    if not encoded:
        return [empty]
    return [encoded.get(i, empty) for i in range(1, max(encoded) + 1)]


def decompose_rle(labeled):
    """the RLE of every mask src/utils.py:61-73 (decompose) would build, without the full-image copies:
    [{'size', 'counts'}] for instance ids 1..max"""
    lab = np.asarray(labeled)
    size = [int(lab.shape[0]), int(lab.shape[1])]
    return [{'size': size, 'counts': c} for c, _ in _instances(encode_labels(lab[None].astype(np.int32))[0], size)]


def rle_from_binary(prediction):
    """src/utils.py:118-120: {'size': [h, w], 'counts': bytes} of a binary mask (any non-zero value = foreground;
    the reference only passes 0/255 and 0/1 masks)"""
    m = np.asarray(prediction)
    if m.ndim != 2:
        raise ValueError('rle_from_binary expects one [h, w] mask')
    size = [int(m.shape[0]), int(m.shape[1])]
    counts, _ = _instances(encode_labels((m != 0).astype(np.int32)[None])[0], size)[0]
    return {'size': size, 'counts': counts}


def bounding_box_from_rle(rle):
    """src/utils.py:123-124 for an RLE produced here or by pycocotools: [x, y, w, h].  The string is a few hundred
    bytes; it is parsed on the host (maskApi.c rleFrString + rleToBbox).  create_annotations() does not come through
    here -- its boxes are computed on the device together with the strings."""
    s = rle['counts']
    if isinstance(s, str):
        s = s.encode('ascii')
    h, w = rle['size']
    cnts, p = [], 0
    while p < len(s):
        x, k, more = 0, 0, True
        while more:
            c = s[p] - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x)
    m = (len(cnts) // 2) * 2
    if m == 0:
        return [0.0, 0.0, 0.0, 0.0]
    xs, ys, xe, ye, cc, xp = w, h, 0, 0, 0, 0
    for j in range(m):
        cc += cnts[j]
        t = cc - j % 2
        y = t % h
        x = (t - y) // h
        if j % 2 == 0:
            xp = x
        elif xp < x:
            ys, ye = 0, h - 1
        xs, xe, ys, ye = min(xs, x), max(xe, x), min(ys, y), max(ye, y)
    return [float(xs), float(ys), float(xe - xs + 1), float(ye - ys + 1)]


def annotations_from_probabilities(image_ids, probs, category_ids, category_layers, target_size=None, erode_selem_size=0,
                                   dilate_selem_size=0, watershed_selem_size=0, crf_images=None, crf_params=None):
    """The inference tail without a host round trip: softmax maps (cuda f32 [B,2,h,w]) -> mask_postprocessing
    (src/pipelines.py:248-304) -> create_annotations (src/utils.py:76-115).  The label images stay on the device
    (they are 46 MB per 64 images, the strings a few hundred KB); the result equals
    create_annotations(meta, postprocessing.postprocess_batch(probs, ...), ...).
    crf_images (cuda u8 [B,h,w,3], the de-normalised tiles): the probabilities first go through `dense_crf`
    (src/postprocessing.py:183-225, which no shipped pipeline calls) with `crf_params` (keyword arguments of
    postprocessing.dense_crf_batch); watershed_selem_size > 0: the labelling step is the watershed extension (WATERSHED.md)."""
    from . import postprocessing as post
    if crf_images is not None:
        probs = post.dense_crf_batch(probs.contiguous().float(), crf_images, **(crf_params or {}))
    lab4, counts_h, scores_h = post.postprocess_device(probs, target_size, erode_selem_size, dilate_selem_size, category_layers,
                                                       watershed_selem_size=watershed_selem_size, raw_scores=True)
    B, L, H, W = lab4.shape
    inds = np.cumsum(category_layers)
    n_scored = scores_h.shape[1]
    # only the layers of categories that HAVE an id become annotations (src/utils.py:97-99 drops the background class): the others
    # are neither encoded nor copied
    keep = [l for l in range(min(L, n_scored)) if category_ids[int(np.searchsorted(inds, l, side='right'))] is not None]
    if not keep:
        return []
    sel = lab4 if len(keep) == L else lab4[:, keep].contiguous()
    encoded = encode_labels(sel.view(B * len(keep), H, W), trusted=True)
    size = [int(H), int(W)]
    annotations = []
    for b, image_id in enumerate(image_ids):
        image_id = int(image_id)
        for k, l in enumerate(keep):                  # zip(prediction, image_scores), src/utils.py:96
            category_id = category_ids[int(np.searchsorted(inds, l, side='right'))]
            category_scores = scores_h[b, l, :int(counts_h[b, l])].tolist()
            for (counts, bbox), score in zip(_instances(encoded[b * len(keep) + k], size), category_scores):
                annotations.append({'image_id': image_id, 'category_id': category_id, 'score': score,
                                    'segmentation': {'size': size, 'counts': counts.decode('UTF-8')}, 'bbox': bbox})
    return annotations


def annotations_json(table, chars, image_ids, category_ids, counts, scores, score_off, size):
    """msc_annotations_json: the annotation list of `layers = len(image_ids)` encoded label layers as JSON text (bytes), straight from
    the encoder's table -- json.loads() of it equals the list of dicts create_annotations builds (src/utils.py:76-115)"""
    lib = _lib.load()
    table = np.ascontiguousarray(table, np.int32)
    ids = np.ascontiguousarray(image_ids, np.int64)
    cats = np.ascontiguousarray(category_ids, np.int32)
    cnts = np.ascontiguousarray(counts, np.int32)
    sc = np.ascontiguousarray(scores, np.float64)
    offs = np.ascontiguousarray(score_off, np.int64)
    cap = 64 + 2 * len(chars) + 220 * int(cnts.sum() + len(ids))
    while True:
        buf = C.create_string_buffer(cap)
        need = lib.msc_annotations_json(table.ctypes.data, int(table.shape[0]), chars, len(ids), ids.ctypes.data, cats.ctypes.data, cnts.ctypes.data,
                                        sc.ctypes.data, offs.ctypes.data, int(size[0]), int(size[1]), buf, cap)
        if need < 0:
            _lib.check(int(need), 'msc_annotations_json')
        if need <= cap:
            return buf.raw[:need]
        cap = int(need)


def annotations_json_from_probabilities(image_ids, probs, category_ids, category_layers, target_size=None, erode_selem_size=0,
                                        dilate_selem_size=0, watershed_selem_size=0, crf_images=None, crf_params=None):
    """annotations_from_probabilities with the result as the JSON document `create_annotations(..., save=True)` writes to
    submission.json (src/utils.py:105-110): no Python object per instance -- the device chain, one table copy, one native pass."""
    from . import postprocessing as post
    if crf_images is not None:
        probs = post.dense_crf_batch(probs.contiguous().float(), crf_images, **(crf_params or {}))
    lab4, counts_h, scores_h = post.postprocess_device(probs, target_size, erode_selem_size, dilate_selem_size, category_layers,
                                                       watershed_selem_size=watershed_selem_size, raw_scores=True)
    B, L, H, W = lab4.shape
    inds = np.cumsum(category_layers)
    n_scored, cap = scores_h.shape[1], scores_h.shape[2]
    keep = [l for l in range(min(L, n_scored)) if category_ids[int(np.searchsorted(inds, l, side='right'))] is not None]
    if not keep:
        return b'[]'
    sel = lab4 if len(keep) == L else lab4[:, keep].contiguous()
    table, chars = encode_labels(sel.view(B * len(keep), H, W), trusted=True, raw=True)
    keep_a = np.asarray(keep)
    ids = np.repeat(np.asarray(image_ids, np.int64), len(keep))
    cats = np.tile(np.asarray([category_ids[int(np.searchsorted(inds, l, side='right'))] for l in keep], np.int32), B)
    cnts = counts_h[:, keep_a].reshape(-1)
    offs = ((np.arange(B, dtype=np.int64)[:, None] * n_scored + keep_a[None, :]) * cap).reshape(-1)
    return annotations_json(table, chars, ids, cats, cnts, scores_h, offs, (H, W))


def create_annotations(meta, predictions, logger, category_ids, category_layers, save=False, experiment_dir='./', chunk=64):
    """src/utils.py:76-115.  predictions: iterable of (labelled layers int[L,H,W], per-layer score lists) as produced by
    the `score_builder` Step; images are encoded `chunk` at a time on the device."""
    annotations = []
    logger.info('Creating annotations')
    inds = np.cumsum(category_layers)
    pending = []            # (image_id, category_id, layer array, scores)

    def flush():
        by_shape = {}
        for item in pending:
            by_shape.setdefault(item[2].shape, []).append(item)
        encoded = {}
        for shape, items in by_shape.items():
            enc = encode_labels(np.stack([it[2] for it in items]).astype(np.int32))
            for it, e in zip(items, enc):
                encoded[id(it)] = e
        for it in pending:
            image_id, category_id, layer, scores = it
            size = [int(layer.shape[0]), int(layer.shape[1])]
            for (counts, bbox), score in zip(_instances(encoded[id(it)], size), scores):
                annotations.append({'image_id': int(image_id), 'category_id': category_id, 'score': score,
                                    'segmentation': {'size': size, 'counts': counts.decode('UTF-8')}, 'bbox': bbox})
        del pending[:]

    images = 0
    for image_id, (prediction, image_scores) in zip(meta['ImageId'].values, predictions):
        for category_ind, (category_instances, category_scores) in enumerate(zip(prediction, image_scores)):
            category_nr = int(np.searchsorted(inds, category_ind, side='right'))
            if category_ids[category_nr] is not None:
                pending.append((image_id, category_ids[category_nr], np.asarray(category_instances), category_scores))
        images += 1
        if images % chunk == 0:
            flush()
    flush()
    if save:
        submission_filepath = os.path.join(experiment_dir, 'submission.json')
        with open(submission_filepath, 'w') as fp:
            fp.write(str(json.dumps(annotations)))
            logger.info('Submission saved to {}'.format(submission_filepath))
            if annotations:
                logger.info('submission head \n\n{}'.format(annotations[0]))
        return True
    return annotations
