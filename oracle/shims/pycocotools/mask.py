"""pycocotools.mask subset used by src/utils.py:118-124 and src/postprocessing.py:318 (encode, toBbox) + decode/area."""
import numpy as np

from oracle import annot_ref


def encode(bimask):
    m = np.asarray(bimask)
    if m.ndim == 3:
        return [annot_ref.encode(m[:, :, i]) for i in range(m.shape[2])]
    return annot_ref.encode(m)


def toBbox(rle):
    if isinstance(rle, (list, tuple)):
        return np.stack([annot_ref.to_bbox(r) for r in rle])
    return annot_ref.to_bbox(rle)


def decode(rle):
    h, w = rle['size']
    return annot_ref.rle_decode(annot_ref.rle_from_string(rle['counts']), h, w)


def area(rle):
    return int(sum(annot_ref.rle_from_string(rle['counts'])[1::2]))
