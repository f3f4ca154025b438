"""cProfile of the inference tail (post-processing on the device -> annotations) on synthetic blob probabilities: where the HOST time
of utils.annotations_from_probabilities goes (syncs, copies, allocations, Python).  Usage: python tools/tail_profile.py [ws] [crf]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mapping_challenge_amd import utils      # noqa: E402
from oracle import post_ref                  # noqa: E402  (synthetic inputs only)

ws = int(sys.argv[1]) if len(sys.argv) > 1 else 0
crf = len(sys.argv) > 2 and sys.argv[2] == '1'
B = 32
probs = torch.from_numpy(post_ref.synthetic_probs(B, 256, 256, seed=1234)).cuda()
rgb = (torch.rand(B, 256, 256, 3, device='cuda') * 255).to(torch.uint8)
ids = list(range(B))


def tail():
    return utils.annotations_from_probabilities(ids, probs, [None, 100], [1, 1], (300, 300), 0, 2, watershed_selem_size=ws,
                                                crf_images=rgb if crf else None)


for _ in range(3):
    ann = tail()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    tail()
torch.cuda.synchronize()
print('tail ws=%d crf=%d: %.3f ms per batch of %d (%d annotations)' % (ws, crf, 1e3 * (time.perf_counter() - t0) / 20, B, len(ann)))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tail()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
