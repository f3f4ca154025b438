"""GPU: target preparation (csrc/prep.hip through the C ABI) against oracle/prep_ref.py (scipy's distance_transform_edt,
the function the reference itself calls).  Squared distances are integers, so distances / sizes / masks are bit-exact."""
import numpy as np
import pytest

from oracle import prep_ref

pytestmark = pytest.mark.gpu


def check(masks, category_nr=None, border_width=0, erode=0, dilate=0, small=14):
    from mapping_challenge_amd import preparation
    ov, d16, sizes, second, kept = preparation.prepare_targets(masks, category_nr, border_width, erode, dilate, small, return_details=True)
    eov, ed16, esizes, esecond, ekept = prep_ref.prepare_targets(masks, category_nr, border_width, erode, dilate, small)
    assert ov.dtype == np.uint8 and d16.dtype == np.float16 and sizes.dtype == np.int64
    assert (kept == ekept).all()
    assert (ov == eov).all()
    assert (second == esecond).all()
    assert (d16.view(np.uint16) == ed16.view(np.uint16)).all()
    assert (sizes == esizes).all()


@pytest.mark.parametrize('n,h,w', [(0, 7, 7), (1, 5, 5), (1, 9, 31), (3, 4, 4), (9, 40, 52), (25, 64, 33), (60, 300, 300)])
def test_prepare_targets_matches_oracle(n, h, w):
    check(prep_ref.synthetic_instances(n, h, w, seed=n * 7 + h))


def test_degenerate_stacks_and_categories():
    base = prep_ref.synthetic_instances(6, 32, 40, seed=2)
    full = np.ones((1, 32, 40), np.uint8)
    check(np.concatenate([full, base]))                      # an image-covering instance first: its zero layer is discarded
    check(np.concatenate([full, full, base[:1]]))
    check(np.concatenate([base[:2], full, base[2:]]))        # ... later: it stays as a zero layer
    check(np.concatenate([full, full]))
    frame_only = np.zeros((2, 32, 40), np.uint8)
    frame_only[0, :2] = 1
    frame_only[1, :, -2:] = 1
    check(frame_only)                                        # everything skipped by is_on_border
    check(base, category_nr=[1, 1, 2, 2, 3, 3])              # later categories overwrite earlier ones
    check(base, border_width=3)
    check(base, category_nr=[1, 1, 2, 2, 2, 2], border_width=4)


@pytest.mark.parametrize('erode,dilate,small', [(3, 0, 3), (2, 0, 14), (4, 3, 4), (3, 2, 100), (5, 0, 0)])
def test_eroded_and_dilated_variants(erode, dilate, small):
    """src/preparation.py:61-77: per instance binary erosion (big) / identity or dilation (small), dropped objects restored in
    the erode-only form, distances from the transformed instances, is_on_border from the annotations"""
    for n, h, w, seed in ((9, 40, 52, 4), (25, 64, 33, 7), (40, 300, 300, 11)):
        check(prep_ref.synthetic_instances(n, h, w, seed=seed), erode=erode, dilate=dilate, small=small)
    base = prep_ref.synthetic_instances(8, 48, 40, seed=3)
    check(base, category_nr=[2, 1, 2, 1, 3, 3, 1, 2], erode=erode, dilate=dilate, small=small)          # unsorted categories
    check(base, category_nr=[1, 1, 2, 2, 2, 2, 1, 1], border_width=3, erode=erode, dilate=dilate, small=small)


def test_size_matrix_and_argument_errors():
    from mapping_challenge_amd import preparation
    rng = np.random.default_rng(0)
    for shape in ((1, 1), (3, 50), (64, 64), (300, 300)):
        m = (rng.random(shape) > 0.45).astype(np.uint8)
        got = preparation.get_size_matrix(m)
        assert got.dtype == np.int64 and (got == prep_ref.get_size_matrix(m)).all()
    ov, _, _ = preparation.prepare_targets(np.zeros((1, 8, 8), np.uint8), erode=0, dilate=2)       # plain path (:57-60)
    assert not ov.any()
    with pytest.raises(ValueError):
        preparation.prepare_targets(np.zeros((1, 8, 8), np.uint8), erode=-1)
