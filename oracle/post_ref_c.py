"""TEST INFRASTRUCTURE -- ctypes loader for oracle/post_ref.c (built by __graft_entry__.build() into
oracle/build/libpost_ref.so): the single-thread C restatement of the default post-processing chain that
bench.py times as the CPU baseline.  Checked against oracle/post_ref.py in tests/test_oracle.py."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'build', 'libpost_ref.so')
_lib = None


def load(build_if_missing=True):
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) and build_if_missing:
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.run(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', _SO, os.path.join(_HERE, 'post_ref.c'), '-lm'], check=True)
        _lib = C.CDLL(_SO)
        _lib.msc_ref_postprocess.restype = C.c_int
        _lib.msc_ref_postprocess.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_int]
    return _lib


def postprocess(probs, target_size, dilate, max_labels=4096):
    """probs f32[2,h,w] -> (labels i32[2,H,W], [[scores...], [scores...]]) like post_ref.postprocess"""
    lib = load()
    probs = np.ascontiguousarray(probs, np.float32)
    _, h, w = probs.shape
    H, W = target_size
    labels = np.empty((2, H, W), np.int32)
    counts = np.zeros(2, np.int32)
    scores = np.zeros((2, max_labels), np.float64)
    rc = lib.msc_ref_postprocess(probs.ctypes.data, h, w, H, W, int(dilate), labels.ctypes.data, counts.ctypes.data,
                                 scores.ctypes.data, max_labels)
    if rc != 0:
        raise RuntimeError('msc_ref_postprocess: more than %d components' % max_labels)
    return labels, [list(scores[c, :counts[c]]) for c in range(2)]
