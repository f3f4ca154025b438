"""CPU: the target-preparation oracle (oracle/prep_ref.py) against the reference's own functions
(src/preparation.py, run through the shims; only where /root/reference exists) and against hand-checkable cases."""
import numpy as np
import pytest

from oracle import prep_ref, ref_import

needs_ref = pytest.mark.skipif(not ref_import.available(), reason='/root/reference not present')


def test_hand_checkable_cases():
    m = np.zeros((2, 9, 9), np.uint8)
    m[0, 4, 2] = 1                         # one pixel at (y 4, x 2)
    m[1, 4, 6] = 1                         # one pixel at (4, 6)
    ov, d16, sizes, second, kept = prep_ref.prepare_targets(m)
    assert kept.tolist() == [1, 1] and ov.sum() == 2 and ov[4, 2] == 1 and ov[4, 6] == 1
    assert second[4, 2] == 4.0 and second[4, 6] == 4.0 and second[4, 4] == 2.0
    assert d16[4, 4] == np.float16(4.0) and d16[0, 2] == np.float16(4.0 + np.sqrt(32.0))
    assert sizes[4, 2] == 1 and (sizes == 1).all()
    # instances that only touch the 2-pixel frame are skipped; a single instance doubles its own distance
    m2 = np.zeros((2, 9, 9), np.uint8)
    m2[0, 0:2, :] = 1
    m2[1, 3:6, 3:6] = 1
    ov, d16, sizes, second, kept = prep_ref.prepare_targets(m2)
    assert kept.tolist() == [0, 1] and ov[0].sum() == 0 and sizes[4, 4] == 9 and sizes[0, 0] == 1
    assert d16[4, 0] == np.float16(6.0) and second[4, 0] == 3.0
    # no instance at all
    ov, d16, sizes, second, kept = prep_ref.prepare_targets(np.zeros((0, 7, 7), np.uint8))
    assert not ov.any() and not d16.any() and (sizes == 1).all()


@needs_ref
def test_restatement_equals_reference_functions():
    prep = ref_import.ref('preparation')
    rng = np.random.default_rng(1)
    masks = prep_ref.synthetic_instances(9, 40, 52, seed=4)
    for mi in masks:
        assert prep.is_on_border(mi, 2) == prep_ref.is_on_border(mi, 2)
    # the reference's own accumulation loop (overlay_masks_from_annotations without the COCO decoding)
    dist = np.zeros((40, 52))
    mask = np.zeros((40, 52))
    for mi in masks:
        if prep.is_on_border(mi, 2):
            continue
        dist = prep.update_distances(dist, mi)
        mask += mi
    overlay = np.where(np.where(mask > 0, 1, 0).astype('uint8'), 1, np.zeros((40, 52)).astype('uint8'))
    ref_sizes = prep.get_size_matrix(overlay)
    ref_d16, ref_second = prep.clean_distances(dist.copy())
    ov, d16, sizes, second, kept = prep_ref.prepare_targets(masks)
    assert (ov == overlay).all() and (sizes == ref_sizes).all()
    assert (d16.view(np.uint16) == ref_d16.view(np.uint16)).all() and (second == ref_second).all()
    # degenerate stacks: none, one, an image-covering instance first
    for ms in (masks[:0], masks[3:4], np.concatenate([np.ones((1, 40, 52), np.uint8), masks[:4]])):
        dist = np.zeros((40, 52))
        for mi in ms:
            if not prep.is_on_border(mi, 2):
                dist = prep.update_distances(dist, mi)
        ref_d16, ref_second = prep.clean_distances(dist.copy())
        _, d16, _, second, _ = prep_ref.prepare_targets(ms)
        assert (d16.view(np.uint16) == ref_d16.view(np.uint16)).all() and (second == ref_second).all()
    m = rng.random((30, 30)) > 0.5
    assert (prep.get_size_matrix(m.astype(np.uint8)) == prep_ref.get_size_matrix(m.astype(np.uint8))).all()
    # the eroded / eroded+dilated variants: the reference's per-instance functions and its accumulation loops
    utils = ref_import.ref('utils')
    for erode, dilate, small in ((3, 0, 3), (2, 0, 14), (4, 3, 4), (3, 2, 100)):
        for mi in masks:
            assert (prep.get_simple_eroded_mask(mi, erode, small) == prep_ref.get_simple_eroded_mask(mi, erode, small)).all()
            if dilate:
                assert (prep.get_simple_eroded_dilated_mask(mi, erode, dilate, small) ==
                        prep_ref.get_simple_eroded_dilated_mask(mi, erode, dilate, small)).all()
        dist, mask, plain = np.zeros((40, 52)), np.zeros((40, 52)), np.zeros((40, 52))
        for mi in masks:                      # overlay_eroded_masks_from_annotations / ..._dilated_... without the COCO decoding
            if prep.is_on_border(mi, 2):
                continue
            m_ = prep.get_simple_eroded_mask(mi, erode, small) if dilate == 0 else prep.get_simple_eroded_dilated_mask(mi, erode, dilate, small)
            dist = prep.update_distances(dist, m_)
            mask += m_
            plain += mi
        mask = np.where(mask > 0, 1, 0).astype('uint8')
        if dilate == 0:
            mask = utils.add_dropped_objects(np.where(plain > 0, 1, 0).astype('uint8'), mask)
        overlay = np.where(mask, 1, np.zeros((40, 52)).astype('uint8'))
        ref_d16, ref_second = prep.clean_distances(dist.copy())
        ov, d16, sizes, second, kept = prep_ref.prepare_targets(masks, erode=erode, dilate=dilate, small_annotations_size=small)
        assert (ov == overlay).all() and (sizes == prep.get_size_matrix(overlay)).all()
        assert (d16.view(np.uint16) == ref_d16.view(np.uint16)).all() and (second == ref_second).all()
