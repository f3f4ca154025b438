"""TEST INFRASTRUCTURE -- numpy/scipy oracle for the mask post-processing chain.

Each function restates one reference function (file:line given) in window / loop form so that it
does not depend on scikit-image (unpinned in the reference, not installed here).  scipy.ndimage
(installed, and the library the reference itself calls for labelling) is used only for `label`.
Pinned in tests/test_oracle.py against the reference's own functions executed through
the skimage-on-scipy shim, and by the docstring known-answer example of label_multiclass_image
(src/postprocessing.py:96-111).  Unpinned by the reference: skimage's even-kernel origin, resize
edge handling / output dtype.
"""
import numpy as np
from scipy import ndimage as ndi

CATEGORY_LAYERS = [1, 1]      # src/pipeline_config.py:18


def softmax(x, axis):
    # src/utils.py:231-273 (theta=1): max-subtract, exp, divide by the sum along `axis`
    y = x - np.expand_dims(np.max(x, axis=axis), axis)
    y = np.exp(y)
    return y / np.expand_dims(np.sum(y, axis=axis), axis)


def resize_image(image, target_size):
    # src/postprocessing.py:48-61 -> skimage resize(order=1, mode='constant') -> for a (C,H,W)->(C,h,w)
    # request skimage's n-d branch: scipy.ndimage.map_coordinates(order=1, mode='constant', cval=0)
    # sampled at src = in/out*(dst+0.5)-0.5.  scipy's 'constant' mode does NOT interpolate beyond
    # the edges: a sample whose coordinate falls outside [0, n-1] on any axis is cval (0) outright,
    # so when upscaling the first and last output row/column are 0.  float64 result.
    # The ARITHMETIC is scipy's, rounding for rounding (NI_GeometricTransform: per tap the coefficient is
    # multiplied by the weight of each axis in turn, taps summed in raster order; weights 1-a and 1-(1-a)):
    # tests/test_oracle.py pins this function to the installed scipy's map_coordinates with array_equal,
    # because the reference thresholds this float64 map and a tie at 0.5 is decided by the last bit.
    c, h, w = image.shape
    th, tw = target_size
    img = image.astype(np.float64)
    ys = (h / th) * (np.arange(th) + 0.5) - 0.5
    xs = (w / tw) * (np.arange(tw) + 0.5) - 0.5
    oky = (ys >= 0) & (ys <= h - 1)
    okx = (xs >= 0) & (xs <= w - 1)
    y0 = np.clip(np.floor(ys).astype(np.int64), 0, h - 1)
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, w - 1)
    y1 = np.minimum(y0 + 1, h - 1)
    x1 = np.minimum(x0 + 1, w - 1)
    wy0 = 1.0 - (ys - y0)
    wy1 = 1.0 - wy0
    wx0 = 1.0 - (xs - x0)
    wx1 = 1.0 - wx0
    wy0, wy1 = wy0[None, :, None], wy1[None, :, None]
    wx0, wx1 = wx0[None, None, :], wx1[None, None, :]

    def tap(yy, xx):
        return img[:, yy[:, None], xx[None, :]]
    out = (tap(y0, x0) * wy0) * wx0
    out = out + (tap(y0, x1) * wy0) * wx1
    out = out + (tap(y1, x0) * wy1) * wx0
    out = out + (tap(y1, x1) * wy1) * wx1
    out = out * (oky[None, :, None] & okx[None, None, :])
    return np.clip(out, min(img.min(), 0.0), max(img.max(), 0.0))       # skimage: warp(..., clip=True)


def categorize_image(image):
    # src/postprocessing.py:64-74
    return np.argmax(image, axis=0)


def categorize_multilayer_image(image, category_layers=CATEGORY_LAYERS):
    # src/postprocessing.py:77-84: per class c, thresholds arange(1/(L_c+1), 1, 1/(L_c+1)); prob > thr
    layers = []
    for cat, prob in enumerate(image):
        step = 1. / (category_layers[cat] + 1)
        for thr in np.arange(step, 1, step):
            layers.append(prob > thr)
    return np.stack(layers)


def label(mask):
    # src/utils.py:328-330: scipy.ndimage.label, default structure = 4-connectivity, labels numbered
    # in raster order of each component's first pixel, int32
    return ndi.label(mask)[0]


def label_unionfind(mask):
    """Independent pure-numpy restatement of `label` (two-pass union-find, 4-connectivity,
    raster-order numbering) -- cross-checks scipy and documents the semantics the HIP kernel and
    oracle/post_ref.c implement."""
    mask = np.asarray(mask).astype(bool)
    h, w = mask.shape
    parent = np.arange(h * w, dtype=np.int64)

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for y in range(h):
        for x in range(w):
            if not mask[y, x]:
                continue
            i = y * w + x
            for j in ((i - 1) if x > 0 and mask[y, x - 1] else -1, (i - w) if y > 0 and mask[y - 1, x] else -1):
                if j >= 0:
                    ra, rb = find(i), find(j)
                    if ra != rb:
                        parent[max(ra, rb)] = min(ra, rb)
    out = np.zeros((h, w), dtype=np.int32)
    next_label = 0
    ids = {}
    for y in range(h):
        for x in range(w):
            if mask[y, x]:
                r = find(y * w + x)
                if r not in ids:
                    next_label += 1
                    ids[r] = next_label
                out[y, x] = ids[r]
    return out


def label_multilayer_image(mask):
    # src/postprocessing.py:127-132
    return np.stack([label(ch) for ch in mask])


def label_multiclass_image(mask):
    # src/postprocessing.py:87-124
    return np.stack([label(mask == c) for c in range(0, mask.max() + 1)])


def window_offsets(k):
    """Offsets of the k x k rectangle skimage applies for erosion AND dilation: odd k is centred;
    even k is padded to odd with a zero row/col on the top/left (skimage `_shift_selem`), i.e. the
    window covers -(k/2-1) .. +k/2 (k=2: {0,+1}).  Dilation pre-mirrors the footprint to cancel
    scipy.ndimage.grey_dilation's own mirroring, so both use the same offsets."""
    lo = -((k - 1) // 2)
    return lo, lo + k - 1


def _rect_filter(img, k, fn):
    lo, hi = window_offsets(k)
    h, w = img.shape
    # scipy 'reflect' border == clamp for min/max filters whose window stays within one reflection
    p = np.pad(img, ((-lo, hi), (-lo, hi)), mode='symmetric')
    out = None
    for dy in range(lo, hi + 1):
        for dx in range(lo, hi + 1):
            v = p[dy - lo:dy - lo + h, dx - lo:dx - lo + w]
            out = v.copy() if out is None else fn(out, v)
    return out


def add_dropped_objects(original, processed):
    # src/utils.py:333-339
    reconstructed = processed.copy()
    labeled = label(original)
    for i in range(1, labeled.max() + 1):
        # literally the reference's test: np.any over the INDEX arrays np.where returns, so an overlap that consists of
        # pixel (0,0) alone (all indices zero) counts as "no surviving pixel"
        if not np.any(np.where((labeled == i) & (processed != 0))):
            reconstructed += (labeled == i).astype(reconstructed.dtype)
    return reconstructed.astype('uint8')


def erode_image(mask, erode_selem_size):
    # src/postprocessing.py:135-156 (2-D branch; the 3-D branch raises for >= 2 layers in the
    # reference because np.stack sits inside the loop, :153-155 -- so only 2-D is defined behaviour)
    if not erode_selem_size > 0:
        return mask
    if mask.ndim != 2:
        raise ValueError('erode_image: the reference only defines the 2-D case (src/postprocessing.py:153-155)')
    src = mask.astype(np.uint8) if mask.dtype == bool else mask
    eroded = _rect_filter(src, erode_selem_size, np.minimum)
    if mask.dtype == bool:
        eroded = eroded.astype(bool)
    return add_dropped_objects(mask, eroded)


def dilate_image(mask, dilate_selem_size):
    # src/postprocessing.py:159-180: grey max filter over the LABEL image (bigger label id wins)
    if not dilate_selem_size > 0:
        return mask
    if mask.ndim == 2:
        return _rect_filter(mask, dilate_selem_size, np.maximum)
    return np.stack([_rect_filter(m, dilate_selem_size, np.maximum) for m in mask])


def build_score(image, probabilities):
    # src/postprocessing.py:228-236 (zip pairs layer l with probability channel l: Appendix A.7)
    total = []
    for instances, probs in zip(image, probabilities):
        score = []
        for lab in range(1, instances.max() + 1):
            sel = instances == lab
            n = np.count_nonzero(sel)
            mean = probs[sel].mean() if n else np.ma.masked
            score.append(mean * np.sqrt(n))
        total.append(score)
    return image, total


def crop_image_center_per_class(image, h_crop, w_crop):
    # src/postprocessing.py:239-258
    out = []
    for ch in image:
        h, w = ch.shape[:2]
        hs, ws = int((h - h_crop) / 2.), int((w - w_crop) / 2.)
        out.append(ch[hs:-hs, ws:-ws])
    return np.stack(out)


def postprocess(probs, target_size=None, erode=0, dilate=0):
    """The six Steps of mask_postprocessing (src/pipelines.py:248-304) for one image."""
    p = resize_image(probs, target_size) if target_size is not None else probs
    layers = categorize_multilayer_image(p)
    if erode > 0:
        layers = np.stack([erode_image(l, erode) for l in layers])
    labeled = label_multilayer_image(layers)
    dilated = dilate_image(labeled, dilate)
    return build_score(dilated, p)


from synthetic_inputs import synthetic_probs      # noqa: E402,F401  (input generator, shared with bench.py)
