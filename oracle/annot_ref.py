"""TEST INFRASTRUCTURE -- CPU restatement of the annotation encoding after the hot path
(`src/utils.py:61-127`).  Never imported by the product.

The reference delegates the arithmetic to pycocotools==2.0.0 (`environment.yml:27`), which is NOT in
/root/reference and not installed here: PARITY UNPINNED against the real library.  What is restated is the
published algorithm of its `common/maskApi.c` (cocoapi): `rleEncode`, `rleToString`, `rleFrString`,
`rleToBbox`, `rleDecode`.  The reference's own logic around it (`decompose`, `create_annotations`) IS
pinned: tests/test_oracle.py runs the unmodified `src/utils.py` on top of oracle/shims/pycocotools (which
calls this file) and compares it with the restatement below.
"""
import numpy as np


# ---- maskApi.c -----------------------------------------------------------------------------------
def rle_encode(mask):
    """rleEncode: run lengths of the column-major byte stream, starting with a run of zeros.  A new run starts
    whenever the byte VALUE changes (`if(T[j]!=p)`), so the input must be two-valued (0 and one other value)."""
    m = np.asarray(mask)
    t = np.asfortranarray(m).ravel(order='F').astype(np.uint8)
    cnts, p, c = [], 0, 0
    for v in t.tolist():
        if v != p:
            cnts.append(c)
            c, p = 0, v
        c += 1
    cnts.append(c)
    return cnts


def rle_encode_fast(mask):
    """same as rle_encode, vectorised (for full-size tests)"""
    t = np.asfortranarray(np.asarray(mask)).ravel(order='F').astype(np.uint8)
    if t.size == 0:
        return [0]
    change = np.flatnonzero(t[1:] != t[:-1]) + 1
    bounds = np.concatenate(([0], change, [t.size]))
    cnts = np.diff(bounds).tolist()
    if t[0] != 0:
        cnts = [0] + cnts
    return cnts


def rle_to_string(cnts):
    """rleToString: like LEB128 with 5 data bits per char, continuation bit 0x20, chars 48..111; counts after the
    third are stored as differences to the count two places earlier."""
    out = bytearray()
    for i, c in enumerate(cnts):
        x = int(c)
        if i > 2:
            x -= int(cnts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5                      # arithmetic shift, like `long` in C
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(ch + 48)
    return bytes(out)


def rle_from_string(s):
    """rleFrString"""
    if isinstance(s, str):
        s = s.encode('ascii')
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
    return cnts


def rle_to_bbox(cnts, h, w):
    """rleToBbox -> [x, y, w, h] as floats (pycocotools returns a float64 array)"""
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


def rle_decode(cnts, h, w):
    """rleDecode -> uint8 [h, w] of 0/1"""
    flat = np.zeros(h * w, np.uint8)
    pos, v = 0, 0
    for c in cnts:
        flat[pos:pos + c] = v
        pos += c
        v ^= 1
    return flat.reshape((w, h)).T.copy()


# ---- pycocotools.mask API used by the reference -----------------------------------------------------
def encode(mask):
    """cocomask.encode for one [h, w] Fortran-ordered uint8 mask -> {'size': [h, w], 'counts': bytes}"""
    m = np.asarray(mask)
    if m.ndim != 2:
        raise ValueError('one 2-D mask at a time')
    return {'size': [int(m.shape[0]), int(m.shape[1])], 'counts': rle_to_string(rle_encode_fast(m))}


def to_bbox(rle):
    h, w = rle['size']
    return np.array(rle_to_bbox(rle_from_string(rle['counts']), h, w), dtype=np.float64)


# ---- src/utils.py ------------------------------------------------------------------------------------
def decompose(labeled):
    """src/utils.py:61-73: one 0/255 image per instance id 1..max (also for ids without pixels); the image itself
    when there is no instance."""
    nr_true = int(labeled.max())
    masks = []
    for i in range(1, nr_true + 1):
        msk = labeled.copy()
        msk[msk != i] = 0
        msk[msk == i] = 255
        masks.append(msk)
    return masks if masks else [labeled]


def rle_from_binary(prediction):
    """src/utils.py:118-120"""
    return encode(np.asfortranarray(prediction))


def bounding_box_from_rle(rle):
    """src/utils.py:123-124"""
    return list(to_bbox(rle))


def create_annotations(image_ids, predictions, category_ids, category_layers):
    """src/utils.py:76-115 without logging / saving: predictions = iterable of (labelled layers, per-layer scores)"""
    annotations = []
    inds = np.cumsum(category_layers)
    for image_id, (prediction, image_scores) in zip(image_ids, predictions):
        for category_ind, (instances, scores) in enumerate(zip(prediction, image_scores)):
            category_nr = int(np.searchsorted(inds, category_ind, side='right'))
            if category_ids[category_nr] is not None:
                for mask, score in zip(decompose(instances), scores):
                    rle = rle_from_binary(mask.astype('uint8'))
                    annotations.append({'image_id': int(image_id), 'category_id': category_ids[category_nr], 'score': score,
                                        'segmentation': {'size': rle['size'], 'counts': rle['counts'].decode('UTF-8')},
                                        'bbox': bounding_box_from_rle(rle)})
    return annotations
