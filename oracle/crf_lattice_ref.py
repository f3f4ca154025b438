"""TEST INFRASTRUCTURE -- ctypes loader for oracle/crf_lattice_ref.c (built by __graft_entry__.build() into
oracle/build/libcrf_lattice_ref.so): dense-CRF mean field with PERMUTOHEDRAL-LATTICE filtering, restated from the papers the
reference's `dense_crf` docstring cites (src/postprocessing.py:189-192) -- what pydensecrf computes, as far as the published
algorithm determines it.  Used by tests/test_oracle_crf.py to state the distance between lattice filtering and the exact windowed
filtering of oracle/crf_ref.py (= the HIP kernel).  pydensecrf itself is absent: parity with its binary stays unpinned."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'build', 'libcrf_lattice_ref.so')
_lib = None


def load(build_if_missing=True):
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, 'crf_lattice_ref.c')
        stale = os.path.exists(_SO) and os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(_SO)
        if (not os.path.exists(_SO) or stale) and build_if_missing:
            os.makedirs(os.path.dirname(_SO), exist_ok=True)
            subprocess.run(['gcc', '-O2', '-shared', '-fPIC', '-o', _SO, src, '-lm'], check=True)
        _lib = C.CDLL(_SO)
        _lib.msc_ref_lattice_filter.restype = C.c_int
        _lib.msc_ref_lattice_filter.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _lib.msc_ref_dense_crf_lattice.restype = C.c_int
        _lib.msc_ref_dense_crf_lattice.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                                   C.c_float, C.c_float, C.c_int, C.c_void_p]
    return _lib


def lattice_filter(features, values):
    """features f32[n,d] (already divided by the kernel's standard deviations), values f32[n,vs] -> filtered f32[n,vs]:
    splat / blur / slice of Adams et al. 2010; approximates sum_j exp(-|f_i - f_j|^2 / 2) v_j up to the lattice's constant gain"""
    lib = load()
    f = np.ascontiguousarray(features, np.float32)
    v = np.ascontiguousarray(values, np.float32)
    out = np.empty_like(v)
    rc = lib.msc_ref_lattice_filter(f.ctypes.data, f.shape[0], f.shape[1], v.ctypes.data, out.ctypes.data, v.shape[1])
    assert rc == 0
    return out


def mean_field(unary, rgb, compat_gaussian=3, sxy_gaussian=1, compat_bilateral=10, sxy_bilateral=1, srgb=50, iterations=5):
    """unary f32[M,H,W] energies, rgb u8[H,W,3] -> Q f32[M,H,W]"""
    lib = load()
    u = np.ascontiguousarray(unary, np.float32)
    im = np.ascontiguousarray(rgb, np.uint8)
    out = np.empty_like(u)
    rc = lib.msc_ref_dense_crf_lattice(u.ctypes.data, im.ctypes.data, u.shape[0], u.shape[1], u.shape[2], float(compat_gaussian),
                                       float(sxy_gaussian), float(compat_bilateral), float(sxy_bilateral), float(srgb), int(iterations),
                                       out.ctypes.data)
    assert rc == 0
    return out


def dense_crf(img, output_probs, mean, std, compat_gaussian=3, sxy_gaussian=1, compat_bilateral=10, sxy_bilateral=1, srgb=50,
              iterations=5):
    """src/postprocessing.py:183-225 with lattice filtering (img: normalised f[3,H,W], probs f[C,H,W])"""
    probs = np.asarray(output_probs)
    unary = -np.log(np.clip(probs, 1e-5, 1.0)).astype(np.float32)          # unary_from_softmax
    org = np.asarray(img) * np.array(std).reshape(3, 1, 1) + np.array(mean).reshape(3, 1, 1)
    org = np.ascontiguousarray((org * 255.).transpose(1, 2, 0), dtype=np.uint8)
    return mean_field(unary, org, compat_gaussian, sxy_gaussian, compat_bilateral, sxy_bilateral, srgb, iterations).reshape(probs.shape)
