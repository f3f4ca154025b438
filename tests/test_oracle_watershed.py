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


def test_independent_pin_scipy_watershed_ift_agreement():
    """`scipy.ndimage.watershed_ift` (installed here; Falcao's image foresting transform as published in scipy) on the same 8-bit
    relief + markers is an INDEPENDENT marker-controlled watershed.  It cannot be bit-matched by the immersion of WATERSHED.md:
    its path cost is the largest |h(p) - h(q)| along the path (relief [0, 10, 1, 0] with markers at both ends floods
    [1, 2, 2, 2]: the 10 goes to the right marker through the cheap 1 -> 10 climb), the immersion's is the largest h(p)
    ([1, 1, 2, 2]: the 10 is reached at level 10 from both sides, the smaller label wins).  What is pinned is the AGREEMENT of
    the two on the chain's own inputs, as measured when this was written: 99.1 % of the mask pixels (worst layer 97.8 %),
    area-weighted instance IoU 0.985, mean IoU of the instances of >= 64 pixels 0.957."""
    from scipy import ndimage as ndi
    cross = ndi.generate_binary_structure(2, 1)
    tiny = ndi.watershed_ift(np.array([[0, 10, 1, 0]], np.uint8), np.array([[1, 0, 0, 2]], np.int32), structure=cross)
    ours = watershed_ref.flood(np.ones((1, 4), bool), np.array([[1, 0, 0, 2]]), np.array([[0, 10, 1, 0]], np.uint8))
    assert tiny.tolist() == [[1, 2, 2, 2]] and ours.tolist() == [[1, 1, 2, 2]]       # the two definitions differ, by design
    probs = post_ref.synthetic_probs(4, 256, 256, seed=77)
    agree, ious, areas = [], [], []
    for pr in probs:
        r = post_ref.resize_image(pr, (300, 300)).astype(np.float32)
        for layer, pch in zip(post_ref.categorize_multilayer_image(r), r):
            for k in (3, 5):
                markers = post_ref.label(post_ref.erode_image(layer, k) != 0)
                h = watershed_ref.relief(pch)
                got = watershed_ref.flood(layer, markers, h)
                # outside the layer the relief is the maximum, so no minimax path to a mask pixel prefers to leave the mask
                ift = ndi.watershed_ift(np.where(layer, h, 255).astype(np.uint8), markers.astype(np.int32), structure=cross)
                ift = np.where(layer, ift, 0)
                assert ((ift > 0) == layer).all() and (ift[markers > 0] == markers[markers > 0]).all()
                agree.append((got == ift)[layer].mean())
                for i in range(1, int(markers.max()) + 1):
                    a, b = got == i, ift == i
                    ious.append((a & b).sum() / max(1, (a | b).sum()))
                    areas.append(a.sum())
    ious, areas = np.array(ious), np.array(areas)
    assert np.mean(agree) > 0.985 and np.min(agree) > 0.97, (np.mean(agree), np.min(agree))
    assert (ious * areas).sum() / areas.sum() > 0.975 and ious[areas >= 64].mean() > 0.94, ((ious * areas).sum() / areas.sum(), ious[areas >= 64].mean())
