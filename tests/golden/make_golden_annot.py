"""Generates tests/golden/annot.json BY RUNNING THE REFERENCE's src/utils.py (decompose, create_annotations,
rle_from_binary, bounding_box_from_rle) in the build container, on top of oracle/shims/pycocotools -- the
library itself (pycocotools==2.0.0) is not installed, so the run-length arithmetic underneath is
oracle/annot_ref.py (published maskApi.c algorithm): this fixture pins the reference's own logic around it.

    python tests/golden/make_golden_annot.py
"""
import json
import logging
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_import, post_ref       # noqa: E402


def synthetic_predictions(n=3, hw=(75, 75), seed=77):
    """(labelled layers, scores) per image through the oracle post chain, plus hand-made corner cases"""
    probs = post_ref.synthetic_probs(n, 64, 64, seed=seed, smooth=2.0)
    preds = [post_ref.postprocess(p, hw, 0, 2) for p in probs]
    lab = np.zeros((2,) + hw, np.int32)
    lab[1, 0, 0] = 1                      # first pixel
    lab[1, -1, -1] = 2                    # last pixel: no trailing zero run
    lab[1, 10:20, 30] = 4                 # id 3 has no pixel -> empty mask
    lab[1, :, 50:52] = 5                  # run crossing a column boundary -> full-height box
    preds.append((lab, [[], [0.5, 0.25, 0.125, 0.0625, 1.0]]))
    preds.append((np.zeros((2,) + hw, np.int32), [[], []]))       # nothing found
    return preds


def main():
    import pandas as pd
    ref_import.install()
    utils = ref_import.ref('utils')
    preds = synthetic_predictions()
    meta = pd.DataFrame({'ImageId': list(range(100, 100 + len(preds)))})
    ann = utils.create_annotations(meta, preds, logging.getLogger('golden'), [None, 100], [1, 1])
    for a in ann:
        a['score'] = float(a['score'])
        a['bbox'] = [float(v) for v in a['bbox']]
    with open(os.path.join(HERE, 'annot.json'), 'w') as f:
        json.dump(ann, f)
    print('annot.json: %d annotations' % len(ann))


if __name__ == '__main__':
    main()
