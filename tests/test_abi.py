"""CPU: the C-ABI library loads and exports every symbol include/msc.h declares; host-side argument
validation returns error codes (no kernel is launched without a GPU)."""
import ctypes as C
import os
import re

import pytest

from mapping_challenge_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'msc.h')).read()
    return set(re.findall(r'\b(msc_[a-z0-9_]+)\s*\(', hdr))


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert names == set(_lib.SIGNATURES), (names ^ set(_lib.SIGNATURES))
    for n in names:
        assert getattr(lib, n) is not None
    assert lib.msc_abi_version() == 11


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libmsc_hip.so')
    with pytest.raises(_lib.MscError, match='no CPU fallback'):
        _lib.load()


def _desc(**kw):
    d = _lib.ConvDesc()
    buf = (C.c_char * 4096)()
    base = (C.addressof(buf) + 63) // 64 * 64
    d.in_, d.wt, d.out = base, base + 1024, base + 2048
    d.dtype, d.mode = _lib.BF16, 0
    d.N, d.Hi, d.Wi, d.Cin, d.Ho, d.Wo, d.Cout = 2, 8, 8, 64, 8, 8, 64
    d.in_ld, d.out_ld = 64, 64
    d.KH = d.KW = 3
    d.stride, d.pad = 1, 1
    for k, v in kw.items():
        setattr(d, k, v)
    d._keep = buf
    return d


def test_conv_descriptor_validation_is_host_side():
    lib = _lib.load()
    assert lib.msc_conv_stats_slices(C.byref(_desc())) > 0
    for bad in (dict(Cin=24), dict(Cout=48), dict(in_ld=60), dict(dtype=7), dict(mode=1, stride=1),
                dict(mode=0, flip=1, stride=2), dict(mode=5)):
        assert lib.msc_conv_stats_slices(C.byref(_desc(**bad))) == -1, bad
        assert lib.msc_last_error()
    # stem form: 4-element pixels are fine when every addressed pixel start stays 16-byte aligned
    ok = _desc(in_ld=4, Cin=32, KH=7, KW=1, stride=2, pad=0, Hi=22, Wi=24, Ho=8, Wo=8)
    assert lib.msc_conv_stats_slices(C.byref(ok)) > 0
    assert lib.msc_conv_cfg_ok(C.byref(_desc(in_ld=4, Cin=32, KH=7, KW=1, stride=2, pad=0, Hi=38, Wi=40, Ho=16, Wo=16)), 58)      # the stem's halo kernel
    assert not lib.msc_conv_cfg_ok(C.byref(ok), 58)                                    # ... needs 8 x 16 output patches
    # more than 2^24 output pixels (or 2 GiB of input) per call run as image ranges inside msc_conv_igemm: the descriptor is valid;
    # a single image beyond 2^24 pixels is not
    big = _desc(N=80, Hi=512, Wi=512, Ho=512, Wo=512, KH=1, KW=1, pad=0)
    assert lib.msc_conv_stats_slices(C.byref(big)) > 0 and lib.msc_conv_cfg_ok(C.byref(big), 57)      # 1x1 / stride 1: the streaming kernel applies
    huge = _desc(N=1, Hi=4096, Wi=4100, Ho=4096, Wo=4100, KH=1, KW=1, pad=0)
    assert lib.msc_conv_stats_slices(C.byref(huge)) == -1
    # ABI v6: BatchNorm-backward sums of a residual join -- a residual needs stats_z for the mask, stats_z excludes the coefficients
    import ctypes
    y = (ctypes.c_char * 65536)()
    base = ctypes.addressof(y) + (-ctypes.addressof(y)) % 16
    join = dict(KH=1, KW=1, pad=0, stats=base, stats_kind=1, stats_y=base, stats_y_ld=64)
    assert lib.msc_conv_stats_slices(C.byref(_desc(res=base, res_ld=64, stats_z=base, stats_z_ld=64, **join))) > 0
    assert lib.msc_conv_stats_slices(C.byref(_desc(res=base, res_ld=64, **join))) == -1
    assert lib.msc_conv_stats_slices(C.byref(_desc(stats_z=base, stats_z_ld=64, scale=base, shift=base, **join))) == -1
    assert not lib.msc_conv_cfg_ok(C.byref(_desc(stats_z=base, stats_z_ld=64, Cout=128, **join)), 30)      # 32-fragment wave tiles do not carry it


def test_postprocessing_argument_errors():
    lib = _lib.load()
    assert lib.msc_label4(None, None, None, None, 1, 8, 8, None) < 0
    assert lib.msc_dilate_i32(1, 2, 1, 8, 8, 0, None) < 0          # k <= 0: the reference returns the input unchanged
    assert lib.msc_label_workspace_bytes(2, 16, 16) == 2 * 16 * 16 * 4
    assert lib.msc_crop_center(1, 2, 1, 2, 320, 320, 301, 301, None) < 0   # asymmetric margins: undefined in the reference


def test_capturable_entry_points_issue_kernels_only():
    """round 5: hipMemsetAsync / hipMemcpyAsync inside an entry point of the training step become memset / memcpy NODES of the captured hipGraph, and a
    replayed step holding them ran with garbage gradients.  The translation units whose entry points a captured step may call must
    not contain the runtime calls -- except the two A/B fall-backs of msc_memset_zero / msc_copy behind MSC_MEMOPS_KERNEL=0"""
    import re
    csrc = os.path.join(os.path.dirname(__file__), '..', 'open-solution-mapping-challenge_amd', 'csrc')
    capturable = ['api.hip', 'igemm.hip', 'wgrad.hip', 'conv1x1.hip', 'halo32.hip', 'bottleneck.hip', 'elementwise.hip', 'reduce.hip', 'loss.hip',
                  'common.h', 'conv_common.h', 'dma.h']
    hits = []
    for f in capturable:
        for i, line in enumerate(open(os.path.join(csrc, f)), 1):
            code = line.split('//')[0]
            if re.search(r'hipMem(set|cpy)\w*Async\(', code):      # the synchronous ones build tables (msc_wgrad_group_create), never inside a capture
                hits.append((f, i, code.strip()[:80]))
    assert [h[0] for h in hits] == ['elementwise.hip', 'elementwise.hip'], hits
    assert 'hipMemcpyAsync' in hits[0][2] and 'hipMemsetAsync' in hits[1][2]
