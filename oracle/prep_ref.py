"""TEST INFRASTRUCTURE -- CPU restatement of target preparation (`src/preparation.py:44-198`) for one image,
given the decoded instance masks (what `cocomask.decode(cocomask.frPyObjects(...))` returns; polygon
rasterisation is pycocotools' and out of scope).  Never imported by the product.

`update_distances`, `clean_distances`, `get_size_matrix`, `is_on_border` are pure numpy/scipy functions:
tests/test_oracle_prep.py pins the restatements below against the reference's own functions (where
/root/reference exists) -- and scipy's `distance_transform_edt` is the very function the reference calls.
"""
import numpy as np
from scipy import ndimage as ndi


def is_on_border(mask, border_width):
    """src/preparation.py:197-198"""
    return not np.any(mask[border_width:-border_width, border_width:-border_width])


def update_distances(dist, mask):
    """src/preparation.py:146-151 (note: a stack whose entries are all zero is REPLACED, not extended)"""
    if dist.sum() == 0:
        return ndi.distance_transform_edt(1 - mask)
    return np.dstack([dist, ndi.distance_transform_edt(1 - mask)])


def clean_distances(distances):
    """src/preparation.py:154-163"""
    if distances.ndim < 3:
        distances = np.dstack([distances, distances])
    else:
        distances = np.sort(distances, axis=2)[:, :, :2]
    return np.sum(distances, axis=2).astype(np.float16), distances[:, :, 1]


def get_size_matrix(mask):
    """src/preparation.py:181-187; `label` = scipy.ndimage.label, 4-connectivity (src/utils.py:328-330)"""
    labeled, n = ndi.label(mask)
    sizes = np.ones(mask.shape, np.int64)
    if n:
        areas = np.bincount(labeled.ravel(), minlength=n + 1)
        sizes = np.where(labeled > 0, areas[labeled], 1)
    return sizes


def binary_erosion(mask, k):
    """skimage.morphology.binary_erosion(mask, rectangle(k, k)) (skimage 0.13: scipy's, border_value=True)"""
    return ndi.binary_erosion(mask, structure=np.ones((k, k), np.uint8), border_value=True)


def binary_dilation(mask, k):
    return ndi.binary_dilation(mask, structure=np.ones((k, k), np.uint8))


def get_simple_eroded_mask(mask, selem_size, small_annotations_size):
    """src/preparation.py:166-172"""
    return binary_erosion(mask, selem_size) if mask.sum() > small_annotations_size ** 2 else mask


def get_simple_eroded_dilated_mask(mask, erode_selem_size, dilate_selem_size, small_annotations_size):
    """src/preparation.py:175-182"""
    if mask.sum() > small_annotations_size ** 2:
        return binary_erosion(mask, erode_selem_size)
    return binary_dilation(mask, dilate_selem_size)


def add_dropped_objects(original, processed):
    """src/utils.py:333-339"""
    reconstructed = processed.copy()
    labeled, n = ndi.label(original)
    for i in range(1, n + 1):
        if not np.any(np.where((labeled == i) & processed)):
            reconstructed += (labeled == i)
    return reconstructed.astype('uint8')


def prepare_targets(masks, category_nr=None, border_width=0, erode=0, dilate=0, small_annotations_size=14):
    """overlay_mask_one_image (src/preparation.py:44-84) for instance masks u8 [n,H,W] in annotation order
    (category_nr[i] = index of the instance's category in CATEGORY_IDS, ascending).
    Returns (mask_overlayed u8, distances f16, sizes i64, second_nearest f64, kept i32[n])."""
    masks = np.asarray(masks)
    n = len(masks)
    shape = masks.shape[1:]
    cats = np.ones(n, np.int64) if category_nr is None else np.asarray(category_nr)
    mask_overlayed = np.zeros(shape, np.uint8)
    distances = np.zeros(shape)
    kept = np.zeros(n, np.int32)
    for c in sorted(set(cats.tolist())):
        mask = np.zeros(shape)
        plain = np.zeros(shape)
        for i in np.flatnonzero(cats == c):
            mi = masks[i].reshape(shape)
            if is_on_border(mi, 2):
                continue
            plain += mi
            if erode > 0:          # :61-77: distances and overlay from the eroded (or eroded / dilated) instance
                mi = (get_simple_eroded_mask(mi, erode, small_annotations_size) if dilate == 0 else
                      get_simple_eroded_dilated_mask(mi, erode, dilate, small_annotations_size)).astype(np.uint8)
            kept[i] = 2 if distances.sum() == 0 and mi.all() else 1
            distances = update_distances(distances, mi)
            mask += mi
        mask = np.where(mask > 0, 1, 0).astype('uint8')
        if erode > 0 and dilate == 0:      # :62-71
            mask = add_dropped_objects(np.where(plain > 0, 1, 0).astype('uint8'), mask)
        mask_overlayed = np.where(mask, c, mask_overlayed).astype(np.uint8)
    sizes = get_size_matrix(mask_overlayed)
    dist16, second = clean_distances(distances)
    if border_width > 0:
        borders = (second < border_width) & (~mask_overlayed)
        mask_overlayed = np.where(borders, mask_overlayed.max() + 1, mask_overlayed).astype(np.uint8)
    return mask_overlayed, dist16, sizes, second, kept


def synthetic_instances(n, h, w, seed=0):
    """n rectangular / elliptic 'buildings', some touching the border, some overlapping"""
    rng = np.random.default_rng(seed)
    yy, xx = np.indices((h, w))
    out = np.zeros((n, h, w), np.uint8)
    for i in range(n):
        cy, cx = rng.integers(0, h), rng.integers(0, w)
        ry, rx = rng.integers(1, max(2, h // 6)), rng.integers(1, max(2, w // 6))
        if rng.random() < 0.5:
            out[i] = (np.abs(yy - cy) <= ry) & (np.abs(xx - cx) <= rx)
        else:
            out[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0
    return out
