"""the failing test order with switches: MODE=plain | empty (torch.cuda.empty_cache() between the two tests) | eager (TrainStep without hipGraph) |
sync (a device synchronise + gc between the two)"""
import os, sys, gc, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import test_gpu_configs as T
from mapping_challenge_amd import trainer
mode = os.environ.get('MODE', 'plain')
T.test_category_layers_1_19_chain_matches_the_oracle_including_the_score_zip_quirk()
if mode == 'empty':
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
if mode == 'sync':
    gc.collect(); torch.cuda.synchronize()
if mode == 'noempty':          # torch.cuda.graph.__enter__ calls synchronize / gc.collect / empty_cache before capture_begin
    torch.cuda.empty_cache = lambda: None
if mode == 'nogc':
    import torch.cuda.graphs as G
    G.gc = type('nogc', (), {'collect': staticmethod(lambda *a: 0)})
if mode == 'gconly':           # collect the first test's garbage, but no synchronise
    gc.collect()
if mode == 'eager':
    orig = trainer.TrainStep.__init__
    def init(self, *a, **k):
        k['use_graph'] = False
        orig(self, *a, **k)
    trainer.TrainStep.__init__ = init
try:
    T.test_bf16_training_trajectory_tracks_the_fp32_oracle_resnet101_256()
    res = 'PASS'
except AssertionError as e:
    res = 'FAIL'
t = json.load(open(os.path.join(ROOT, 'gpurun_out', 'parity_configs.json')))['bf16_r101_256_trajectory']
print('MODE=%s: %s rel_max %.4f rel_mean %.4f' % (mode, res, t['rel_max'], t['rel_mean']))
