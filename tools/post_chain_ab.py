"""configs[3] tail on 128 synthetic tiles, as bench.py's north_star block times it: plain chain, full chain (dense CRF + watershed), and the parts.
One process per switch setting (the switches are read once per process):  MSC_CRF_X=0|1  MSC_CCL_STRIP=0|1  MSC_SCORE_LDS=0|1  python tools/post_chain_ab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from mapping_challenge_amd import postprocessing as post
import synthetic_inputs as post_ref


def per_call(fn, iters, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


nb, hw, dev = 128, 256, torch.device('cuda')
probs = torch.from_numpy(post_ref.synthetic_probs(nb, hw, hw, seed=1234)).to(dev)
gen = torch.Generator().manual_seed(1234)
rgb = torch.randint(0, 256, (nb, hw, hw, 3), dtype=torch.uint8, generator=gen).to(dev)
res = {}
res['plain_chain'] = per_call(lambda: post.postprocess_device(probs, (300, 300), 0, 2), 10)
res['full_chain'] = per_call(lambda: post.postprocess_device(post.dense_crf_batch(probs, rgb), (300, 300), 0, 2, watershed_selem_size=5), 5)
res['dense_crf'] = per_call(lambda: post.dense_crf_batch(probs, rgb), 10)
res['dense_crf_32'] = per_call(lambda: post.dense_crf_batch(probs[:32], rgb[:32]), 10)
p300, layers = post.resize_threshold_batch(probs, (300, 300))
flat = layers.view(-1, 300, 300)
res['label'] = per_call(lambda: post.label_batch(flat), 10)
labels, counts = post.label_batch(flat)
mx = int(counts.max().item())
pl = p300.reshape(-1, 300, 300)
res['score'] = per_call(lambda: post.score_batch(labels, pl, mx), 10)
sw = ' '.join('%s=%s' % (k, os.environ[k]) for k in ('MSC_CRF_X', 'MSC_CCL_STRIP', 'MSC_SCORE_LDS') if k in os.environ)
print('[%s] per 128 tiles (ms): ' % sw + '  '.join('%s %.3f' % kv for kv in res.items()) +
      '   | per image: plain %.4f full %.4f' % (res['plain_chain'] / nb, res['full_chain'] / nb), flush=True)
