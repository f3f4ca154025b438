import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapping_challenge_amd import postprocessing as post
from oracle import post_ref
cfg = [1, 19]
probs = post_ref.synthetic_probs(1, 256, 256, seed=91)
r = post_ref.resize_image(probs[0], (300, 300))
cls, thr = post.layer_table(cfg)
for dt in (np.float64, np.float32):
    rr = r.astype(dt)
    lay = post.threshold_batch(torch.from_numpy(rr[None]).cuda(), cfg)[0].cpu().numpy().astype(bool)
    ref = post_ref.categorize_multilayer_image(rr, cfg)
    d = lay != ref
    print(dt.__name__, 'diffs', int(d.sum()), 'per layer', d.reshape(20, -1).sum(1).tolist())
    for l, y, x in list(zip(*np.nonzero(d)))[:8]:
        print('  layer', l, 'cls', cls[l], 'thr', repr(thr[l]), 'val', repr(rr[cls[l], y, x]), 'got', lay[l, y, x], 'ref', ref[l, y, x])
