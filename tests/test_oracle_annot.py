"""CPU: the annotation-encoding oracle (oracle/annot_ref.py) against (a) the known-answer examples of the published
algorithm (maskApi.h documents `M=[0 0 1 1 1 0 1] -> [2 3 1 1]` and `M=[1 1 1 1 1 1 0] -> [0 6 1]`), (b) the
reference's own src/utils.py run through the pycocotools shim, (c) the golden fixture that code generated,
(d) round trips.  pycocotools itself is not installed: parity with the real library is unpinned (annot_ref header)."""
import json
import logging
import os

import numpy as np
import pytest

from oracle import annot_ref, ref_import

needs_ref = pytest.mark.skipif(not ref_import.available(), reason='/root/reference not present')


def golden_predictions():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_annot', os.path.join(os.path.dirname(__file__), 'golden', 'make_golden_annot.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.synthetic_predictions()


def test_known_answers_of_the_published_algorithm():
    # column vectors: h = 7, w = 1, so column-major order is the listed order
    assert annot_ref.rle_encode(np.array([0, 0, 1, 1, 1, 0, 1], np.uint8)[:, None]) == [2, 3, 1, 1]
    assert annot_ref.rle_encode(np.array([1, 1, 1, 1, 1, 1, 0], np.uint8)[:, None]) == [0, 6, 1]
    # column-major: a 2x2 mask [[0,1],[1,1]] reads 0,1,1,1
    assert annot_ref.rle_encode(np.array([[0, 1], [1, 1]], np.uint8)) == [1, 3]
    # string coding by hand: 67 = 0b10_00011 -> chars (3|0x20)+48='S', 2+48='2'; deltas from the 4th count on
    assert annot_ref.rle_to_string([67, 5, 7, 5, 7, 5, 7, 5]) == b'S25700000'
    # a negative delta: 3 - 5 = -2 -> 0b11110 (sign bit 0x10 set, x becomes -1 -> stop) -> 30+48 = 'N'
    assert annot_ref.rle_to_string([1, 5, 1, 3]) == b'151N'
    assert annot_ref.rle_from_string(b'151N') == [1, 5, 1, 3]



# Hand-computed known answers of the published COCO mask coding (cocoapi common/maskApi.c: rleEncode, rleToString, rleToBbox),
# derived on paper from the algorithm's definition -- NOT produced by oracle/annot_ref.py, so they pin it (and the HIP
# encoder) independently.  rleToString per count x (from the 4th count on x -= count two places earlier):
# repeat { c = x & 31; x >>= 5; more = (c & 16) ? x != -1 : x != 0; if (more) c |= 32; emit chr(c + 48) } while (more).
HAND_KATS = [
    # (mask rows, counts string, bbox [x, y, w, h])
    ([[1]], b'01', [0, 0, 1, 1]),                                  # counts [0,1] -> '0','1'
    ([[0]], b'1', [0, 0, 0, 0]),                                   # counts [1]
    ([[1, 1, 1], [1, 1, 1], [1, 1, 1]], b'09', [0, 0, 3, 3]),      # counts [0,9]
    # 2x2 block at (1,1) of a 4x4 mask; column-major stream 0000 0110 0110 0000 -> counts [5,2,2,2,5];
    # 4th count 2-2=0 -> '0', 5th 5-2=3 -> '3'
    ([[0, 0, 0, 0], [0, 1, 1, 0], [0, 1, 1, 0], [0, 0, 0, 0]], b'52203', [1, 1, 2, 2]),
    # one column of 40 pixels, only the last set: counts [39,1]; 39 = 0b1_00111 -> (7|32)+48 = 'W', then 1 -> '1'; then '1'
    ([[0]] * 39 + [[1]], b'W11', [0, 39, 1, 1]),
    # column 0,1,1,1,1,1,0,1,1,1: counts [1,5,1,3]; 4th count 3-5 = -2 -> -2 & 31 = 30, x>>5 = -1, sign bit set -> stop: chr(78)='N'
    ([[0], [1], [1], [1], [1], [1], [0], [1], [1], [1]], b'151N', [0, 1, 1, 9]),
    # two columns of height 3: [[1,0],[0,0],[0,1]] -> stream 1,0,0, 0,0,1 -> counts [0,1,4,1]; 4th: 1-1 = 0 -> '0'
    ([[1, 0], [0, 0], [0, 1]], b'0140', [0, 0, 2, 3]),             # a run crossing a column boundary: bbox spans the full height
]


def test_hand_computed_known_answers_pin_the_oracle():
    for rows, counts, bbox in HAND_KATS:
        m = np.array(rows, np.uint8)
        c = annot_ref.rle_encode(m)
        assert annot_ref.rle_to_string(c) == counts, (rows, annot_ref.rle_to_string(c))
        assert annot_ref.rle_from_string(counts) == c
        assert annot_ref.rle_to_bbox(c, *m.shape) == bbox, (rows, annot_ref.rle_to_bbox(c, *m.shape))
        assert annot_ref.rle_from_binary(m) == {'size': list(m.shape), 'counts': counts.decode()} or \
            annot_ref.rle_from_binary(m)['counts'] in (counts, counts.decode())


def test_round_trips_and_bbox_on_random_masks():
    rng = np.random.default_rng(5)
    for h, w in ((1, 1), (1, 9), (9, 1), (7, 5), (40, 33)):
        for dens in (0.0, 0.1, 0.5, 0.95, 1.0):
            m = (rng.random((h, w)) < dens).astype(np.uint8)
            c = annot_ref.rle_encode(m)
            assert c == annot_ref.rle_encode_fast(m) and sum(c) == h * w
            s = annot_ref.rle_to_string(c)
            assert all(48 <= ch < 112 for ch in s)
            assert annot_ref.rle_from_string(s) == c
            assert (annot_ref.rle_decode(c, h, w) == m).all()
            bb = annot_ref.rle_to_bbox(c, h, w)
            if m.any():
                ys, xs = np.nonzero(m)
                assert bb == [xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1]
            else:
                assert bb == [0, 0, 0, 0]
    # 0/255 masks (what decompose produces) encode like 0/1 masks
    m = (rng.random((12, 12)) < 0.4).astype(np.uint8)
    assert annot_ref.rle_encode(m * 255) == annot_ref.rle_encode(m)


def test_create_annotations_matches_golden(golden_dir):
    gold = json.load(open(os.path.join(golden_dir, 'annot.json')))
    preds = golden_predictions()
    ann = annot_ref.create_annotations(range(100, 100 + len(preds)), preds, [None, 100], [1, 1])
    assert len(ann) == len(gold) == 39
    for a, g in zip(ann, gold):
        assert a['image_id'] == g['image_id'] and a['category_id'] == g['category_id'] and a['segmentation'] == g['segmentation']
        assert [float(v) for v in a['bbox']] == g['bbox'] and float(a['score']) == g['score']
    # every string decodes back to the instance it came from
    k = 0
    for (labels, scores) in preds:
        for i, _ in zip(range(1, int(labels[1].max()) + 1), scores[1]):
            seg = gold[k]['segmentation']
            assert (annot_ref.rle_decode(annot_ref.rle_from_string(seg['counts']), *seg['size']) == (labels[1] == i)).all()
            k += 1
    assert k == len(gold)


@needs_ref
def test_restatement_equals_reference_utils():
    import pandas as pd
    utils = ref_import.ref('utils')
    preds = golden_predictions()
    lab = preds[3][0][1]
    ref_masks, my_masks = utils.decompose(lab), annot_ref.decompose(lab)
    assert len(ref_masks) == len(my_masks) == 5 and all((a == b).all() for a, b in zip(ref_masks, my_masks))
    assert len(utils.decompose(np.zeros((4, 4), np.int32))) == 1
    for m in my_masks:
        r = utils.rle_from_binary(m.astype('uint8'))
        assert r == annot_ref.rle_from_binary(m.astype('uint8'))
        assert list(utils.bounding_box_from_rle(r)) == annot_ref.bounding_box_from_rle(r)
    meta = pd.DataFrame({'ImageId': list(range(100, 100 + len(preds)))})
    ref_ann = utils.create_annotations(meta, preds, logging.getLogger('t'), [None, 100], [1, 1])
    assert ref_ann == annot_ref.create_annotations(meta['ImageId'].values, preds, [None, 100], [1, 1])


def test_string_coding_round_trips_for_arbitrary_counts():
    """the 5-bit / continuation / sign-extension code of rleToString <-> rleFrString over the whole count range, incl. the
    delta coding from the fourth count on (negative and multi-character deltas)"""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=300, deadline=None)
    @given(st.lists(st.integers(min_value=0, max_value=2 ** 31 - 1), min_size=1, max_size=40))
    def check(cnts):
        s = annot_ref.rle_to_string(cnts)
        assert all(48 <= ch < 112 for ch in s) and len(s) <= 7 * len(cnts)
        assert annot_ref.rle_from_string(s) == cnts

    check()
    # boundaries of the character count: 15 / 16 need 1 / 2 chars (bit 4 is the sign bit of the last chunk)
    assert len(annot_ref.rle_to_string([15])) == 1 and len(annot_ref.rle_to_string([16])) == 2
    assert len(annot_ref.rle_to_string([511])) == 2 and len(annot_ref.rle_to_string([512])) == 3
    assert annot_ref.rle_from_string(annot_ref.rle_to_string([0, 90000])) == [0, 90000]


def test_native_json_writer_equals_json_dumps_of_the_reference_style_list():
    """msc_annotations_json (host code of the C-ABI library: no GPU involved): the JSON text it writes parses to exactly the list of
    dicts the Python path builds -- ids without pixels as empty masks, instance-free layers as one empty mask, zip-truncation by
    the number of scores, a count string with a backslash (character 92 is in the COCO alphabet) escaped, floats that read back
    bit-identically"""
    import json
    from mapping_challenge_amd import utils
    H, W = 30, 20
    strings = [b'0a1', b'4\\7' , b'n<0O', b'11']            # the second holds ONE backslash
    chars = b''.join(strings)
    offs = np.cumsum([0] + [len(x) for x in strings])
    # layer 0: ids 1 and 3 present (2 missing); layer 1: no instance; layer 2: id 2 present, but only one score
    table = np.array([[0, 1, offs[0], offs[1], 2, 3, 5, 9], [0, 3, offs[1], offs[2], 0, 0, 19, 29], [2, 2, offs[2], offs[3], 7, 7, 7, 7]], np.int32)
    scores = np.array([0.1, 1.0 / 3.0, 12345.678, 7.5e-06, 2.0, 1e+22], np.float64)
    got = json.loads(utils.annotations_json(table, chars, [11, 12, 13], [100, 100, 7], [3, 1, 1], scores, [0, 3, 4], (H, W)))
    empty = utils._count_chars(H * W).decode()

    def ann(i, c, s, counts, bbox):
        return {'image_id': i, 'category_id': c, 'score': s, 'segmentation': {'size': [H, W], 'counts': counts}, 'bbox': bbox}
    exp = [ann(11, 100, 0.1, '0a1', [2.0, 3.0, 4.0, 7.0]), ann(11, 100, 1.0 / 3.0, empty, [0.0, 0.0, 0.0, 0.0]),
           ann(11, 100, 12345.678, '4\\7', [0.0, 0.0, 20.0, 30.0]), ann(12, 100, 7.5e-06, empty, [0.0, 0.0, 0.0, 0.0]),
           ann(13, 7, 2.0, empty, [0.0, 0.0, 0.0, 0.0])]
    assert got == exp
    assert json.loads(json.dumps(exp)) == got and all(isinstance(a['bbox'][0], float) for a in got)
    assert utils.annotations_json(np.zeros((0, 8), np.int32), b'', [], [], [], np.zeros(0), [], (H, W)) == b'[]'
