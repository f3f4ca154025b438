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


def prepare_targets(masks, category_nr=None, border_width=0):
    """overlay_mask_one_image (src/preparation.py:44-84) with erode = dilate = 0, for instance masks u8 [n,H,W] in
    annotation order (category_nr[i] = index of the instance's category in CATEGORY_IDS, ascending).
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
        for i in np.flatnonzero(cats == c):
            mi = masks[i].reshape(shape)
            if is_on_border(mi, 2):
                continue
            kept[i] = 2 if distances.sum() == 0 and mi.all() else 1
            distances = update_distances(distances, mi)
            mask += mi
        mask_overlayed = np.where(mask > 0, c, mask_overlayed).astype(np.uint8)
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
