"""TEST INFRASTRUCTURE -- CPU restatement of the watershed EXTENSION defined in WATERSHED.md (the reference has no
watershed: parity is undefined; this oracle pins the HIP kernel to the written definition).  Never imported by the product.

Markers come from the reference's own erode_image (+ add_dropped_objects) and label as restated in oracle/post_ref.py
(src/postprocessing.py:135-156, src/utils.py:328-339)."""
import numpy as np

from oracle import post_ref


def relief(prob):
    """WATERSHED.md step 2: h = clamp(floor((1 - P) * 255), 0, 255), evaluated in float32"""
    p = np.asarray(prob, np.float32)
    return np.clip(np.floor((np.float32(1.0) - p) * np.float32(255.0)), 0, 255).astype(np.uint8)


def flood(mask, markers, h):
    """WATERSHED.md step 3: synchronous immersion; ties to the smaller label"""
    mask = np.asarray(mask) != 0
    lab = np.where(mask, np.asarray(markers), 0).astype(np.int32)
    big = np.int32(np.iinfo(np.int32).max)
    for level in range(256):
        elig = mask & (h <= level)
        while True:
            todo = elig & (lab == 0)
            if not todo.any():
                break
            src = np.where(lab > 0, lab, big)
            best = np.full(lab.shape, big, np.int32)
            best[1:, :] = np.minimum(best[1:, :], src[:-1, :])
            best[:-1, :] = np.minimum(best[:-1, :], src[1:, :])
            best[:, 1:] = np.minimum(best[:, 1:], src[:, :-1])
            best[:, :-1] = np.minimum(best[:, :-1], src[:, 1:])
            upd = todo & (best < big)
            if not upd.any():
                break
            lab[upd] = best[upd]
    return lab


def watershed_image(mask, prob, marker_erosion):
    """one layer: labels int32 [H,W]"""
    mask = np.asarray(mask)
    markers = post_ref.label(post_ref.erode_image(mask.astype(bool), marker_erosion) != 0)
    return flood(mask, markers, relief(prob))


def watershed_multilayer_image(image, probabilities, marker_erosion, category_layers=post_ref.CATEGORY_LAYERS):
    """layers [L,H,W] (as categorize_multilayer_image orders them) with the probability channel each was cut from"""
    cls = [c for c, n in enumerate(category_layers) for _ in range(n)]
    return np.stack([watershed_image(layer, probabilities[cls[l]], marker_erosion) for l, layer in enumerate(image)])
