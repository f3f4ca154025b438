"""CPU: the oracle of the watershed EXTENSION (oracle/watershed_ref.py) against the definition in WATERSHED.md -- a
hand-computed case, a second (per-pixel, pure Python) statement of the same definition on small cases, and the properties the
definition implies.  The reference has no watershed: there is no reference parity to pin, and none is claimed."""
import numpy as np

from oracle import post_ref, watershed_ref


def flood_loops(mask, markers, h):
    """WATERSHED.md step 3 spelled out pixel by pixel"""
    H, W = mask.shape
    lab = np.where(mask, markers, 0).astype(np.int64)
    for level in range(256):
        while True:
            new = {}
            for y in range(H):
                for x in range(W):
                    if not mask[y, x] or lab[y, x] != 0 or h[y, x] > level:
                        continue
                    nb = [lab[yy, xx] for yy, xx in ((y - 1, x), (y + 1, x), (y, x - 1), (y, x + 1))
                          if 0 <= yy < H and 0 <= xx < W and lab[yy, xx] > 0]
                    if nb:
                        new[(y, x)] = min(nb)
            if not new:
                break
            for (y, x), v in new.items():
                lab[y, x] = v
    return lab.astype(np.int32)


def test_hand_computed_bridge():
    m = np.zeros((5, 11), bool)
    m[:, 0:4] = True; m[:, 7:11] = True; m[2, 4:7] = True
    p = np.where(m, 0.99, 0.0).astype(np.float32)
    p[2, 4:7] = [0.8, 0.6, 0.7]
    assert watershed_ref.relief(p)[2, 4:7].tolist() == [50, 101, 76] and watershed_ref.relief(p)[0, 0] == 2
    exp = np.zeros((5, 11), np.int32)
    exp[:, 0:4] = 1; exp[:, 7:11] = 2; exp[2, 4:7] = [1, 1, 2]
    assert (watershed_ref.watershed_image(m, p, 3) == exp).all()
    # k = 1 erodes nothing: one marker = the whole (connected) mask -> one instance, like label()
    assert (watershed_ref.watershed_image(m, p, 1) == post_ref.label(m)).all()


def test_vectorised_oracle_equals_per_pixel_statement_and_properties():
    probs = post_ref.synthetic_probs(3, 40, 36, seed=5, smooth=2.0)
    for pr in probs:
        for layer, pch in zip(post_ref.categorize_multilayer_image(pr), pr):
            for k in (2, 3, 5):
                markers = post_ref.label(post_ref.erode_image(layer, k) != 0)
                h = watershed_ref.relief(pch)
                got = watershed_ref.flood(layer, markers, h)
                assert (got == flood_loops(layer, markers, h)).all()
                assert ((got > 0) == layer).all() and (got[markers > 0] == markers[markers > 0]).all()
                for i in range(1, int(got.max()) + 1):          # every region is 4-connected and holds its marker
                    comp = post_ref.label(got == i)
                    assert comp.max() == 1 and (markers[got == i] == i).any()
