"""dense CRF alone: HIP-event time of msc_dense_crf (5 iterations + normalisers) for 32 and 128 images of 256x256, and the result against
oracle/crf_ref.py on a small case.  MSC_CRF_PK=0 selects the round-4 (unpacked) inner loop for the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mapping_challenge_amd import postprocessing as post
from oracle import post_ref, crf_ref
dev = torch.device('cuda')
rng = np.random.default_rng(0)
p = post_ref.synthetic_probs(2, 64, 64, seed=3)
img = rng.integers(0, 256, (2, 64, 64, 3), dtype=np.uint8)
got = post.dense_crf_batch(torch.from_numpy(p).to(dev), torch.from_numpy(img).to(dev)).cpu().numpy()
exp = np.stack([crf_ref.dense_crf_exact(pi, ii) for pi, ii in zip(p, img)]) if hasattr(crf_ref, 'dense_crf_exact') else None
if exp is not None:
    print('max |dQ| vs oracle', float(np.abs(got - exp).max()))
for nb in (32, 128):
    probs = torch.from_numpy(post_ref.synthetic_probs(nb, 256, 256, seed=1234)).to(dev)
    rgb = torch.from_numpy(rng.integers(0, 256, (nb, 256, 256, 3), dtype=np.uint8)).to(dev)
    for _ in range(3):
        post.dense_crf_batch(probs, rgb)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(10):
        post.dense_crf_batch(probs, rgb)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    flop = nb * 256 * 256 * (121 * 13 + 5 * 121 * 20)
    print('PK=%s  %d images: %.3f ms per call, %.1f us per image, %.1f TFLOP/s algorithmic (%.3f of 157.3)' % (os.environ.get('MSC_CRF_PK', '1'), nb, ms, 1e3 * ms / nb, flop / ms / 1e9, flop / ms / 1e9 / 157.3))
