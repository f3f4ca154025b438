"""TEST SUPPORT -- tensor-level wrappers over single C-ABI entry points (include/msc.h), on NHWC cuda tensors.

Used by the kernel parity tests (tests/test_gpu_kernels.py) and the probes; the product drives the C ABI through pre-built
descriptors (unet_models._Builder) and does not go through here (moved out of the package in round 3).
Activations: torch tensors of shape [N,H,W,C] (f32 or bf16), possibly channel slices of a wider buffer
(`x[..., c0:c1]`) -- the channel stride is taken from `x.stride(2)`.
"""
import ctypes as C

import torch

from mapping_challenge_amd import _lib
from mapping_challenge_amd._lib import ConvDesc, WgradDesc, F32, BF16, F16


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError('dtype must be float32, bfloat16 or float16, got %s' % t.dtype)


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _nhwc(t):
    if t.dim() != 4 or t.stride(3) != 1 or t.stride(1) != t.shape[2] * t.stride(2) or t.stride(0) != t.shape[1] * t.stride(1):
        raise ValueError('expected an NHWC tensor or a channel slice of one')
    return t.stride(2)


def conv_igemm(x, w, out, stride=1, pad=0, mode=0, flip=0, relu=False, scale=None, shift=None, res=None, stats=None, cfg=0, in_bn=None):
    """w: [Cout, KH, KW, Cin] in x.dtype.  mode 0: gather conv; mode 1: transposed (stride 2).  in_bn: a _lib.BnInput (ABI v9)"""
    d = ConvDesc()
    if in_bn is not None:
        d.in_bn = C.addressof(in_bn)
    d.in_, d.wt, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.res = res.data_ptr() if res is not None else None
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    d.stats = stats.data_ptr() if stats is not None else None
    d.in_ld, d.out_ld = _nhwc(x), _nhwc(out)
    d.res_ld = _nhwc(res) if res is not None else 0
    d.dtype, d.mode = _dt(x), mode
    d.N, d.Hi, d.Wi, d.Cin = x.shape
    _, d.Ho, d.Wo, d.Cout = out.shape
    d.KH, d.KW = w.shape[1], w.shape[2]
    d.stride, d.pad, d.flip, d.relu, d.cfg = stride, pad, int(flip), int(relu), int(cfg)
    _lib.check(_lib.load().msc_conv_igemm(C.byref(d), _stream(x)), 'msc_conv_igemm')
    return out


def conv_valid_cfgs(x, w, out, stride=1, pad=0, mode=0):
    """explicit kernel configurations usable for this layer (0 = heuristic is always valid)"""
    d = ConvDesc()
    d.in_, d.wt, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.in_ld, d.out_ld, d.dtype, d.mode = _nhwc(x), _nhwc(out), _dt(x), mode
    d.N, d.Hi, d.Wi, d.Cin = x.shape
    _, d.Ho, d.Wo, d.Cout = out.shape
    d.KH, d.KW, d.stride, d.pad = w.shape[1], w.shape[2], stride, pad
    lib = _lib.load()
    return [c for c in range(1, lib.msc_conv_num_cfgs() + 1) if lib.msc_conv_cfg_ok(C.byref(d), c)]


def conv_stats_slices(x, w, out, stride=1, pad=0, cfg=0):
    d = ConvDesc()
    d.in_, d.wt, d.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    d.in_ld, d.out_ld, d.dtype, d.mode = _nhwc(x), _nhwc(out), _dt(x), 0
    d.N, d.Hi, d.Wi, d.Cin = x.shape
    _, d.Ho, d.Wo, d.Cout = out.shape
    d.KH, d.KW, d.stride, d.pad, d.cfg = w.shape[1], w.shape[2], stride, pad, int(cfg)
    return _lib.load().msc_conv_stats_slices(C.byref(d))


def conv_wgrad(p, q, dw, KH, KW, stride=1, pad=0, cfg=0):
    """dw f32 [A, KH, KW, B] += sum_m p[m][a] * q[gather(m)][b]"""
    d = WgradDesc()
    d.p, d.q, d.dw = p.data_ptr(), q.data_ptr(), dw.data_ptr()
    d.p_ld, d.q_ld, d.dtype = _nhwc(p), _nhwc(q), _dt(p)
    d.N, d.Hp, d.Wp, d.A = p.shape
    _, d.Hq, d.Wq, d.B = q.shape
    d.KH, d.KW, d.stride, d.pad, d.cfg = KH, KW, stride, pad, int(cfg)
    _lib.check(_lib.load().msc_conv_wgrad(C.byref(d), _stream(p)), 'msc_conv_wgrad')
    return dw


def _wgrad_desc(p, q, dw, KH, KW, stride, pad, cfg=0):
    d = WgradDesc()
    d.p, d.q, d.dw = p.data_ptr(), q.data_ptr(), dw.data_ptr()
    d.p_ld, d.q_ld, d.dtype = _nhwc(p), _nhwc(q), _dt(p)
    d.N, d.Hp, d.Wp, d.A = p.shape
    _, d.Hq, d.Wq, d.B = q.shape
    d.KH, d.KW, d.stride, d.pad, d.cfg = KH, KW, stride, pad, int(cfg)
    return d


def conv_wgrad_group(problems, steps_per_block=64, tile_cap=128, runs=1, flags=0):
    """problems: [(p, q, dw, KH, KW, stride, pad)]; all weight gradients in one launch per tile shape.
    Returns the number of kernel launches one run takes."""
    lib = _lib.load()
    arr = (WgradDesc * len(problems))(*[_wgrad_desc(*pr) for pr in problems])
    h = C.c_void_p()
    _lib.check(lib.msc_wgrad_group_create(arr, len(problems), steps_per_block, tile_cap, flags, C.byref(h)), 'msc_wgrad_group_create')
    try:
        for _ in range(runs):
            _lib.check(lib.msc_wgrad_group_run(h, _stream(problems[0][0])), 'msc_wgrad_group_run')
        return lib.msc_wgrad_group_launches(h)
    finally:
        torch.cuda.synchronize()
        lib.msc_wgrad_group_destroy(h)


def pack_transpose(src, dtype):
    """f32 [A, T, B] -> dtype [B, T, A]"""
    A, T, B = src.shape
    dst = torch.empty((B, T, A), dtype=dtype, device=src.device)
    _lib.call('msc_pack_transpose', src.data_ptr(), dst.data_ptr(), {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}[dtype], A, T, B, _stream(src))
    return dst


def maxpool2_fwd(x):
    N, H, W, Cc = x.shape
    out = torch.empty((N, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
    _lib.call('msc_maxpool2_fwd', x.data_ptr(), _nhwc(x), out.data_ptr(), Cc, _dt(x), N, H // 2, W // 2, Cc, _stream(x))
    return out


def maxpool2_bwd(dout, x, accumulate_into=None):
    N, H, W, Cc = x.shape
    din = accumulate_into if accumulate_into is not None else torch.empty_like(x)
    _lib.call('msc_maxpool2_bwd', dout.data_ptr(), _nhwc(dout), x.data_ptr(), _nhwc(x), din.data_ptr(), _nhwc(din), _dt(x),
              N, H // 2, W // 2, Cc, int(accumulate_into is not None), _stream(x))
    return din


def adam_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0):
    _lib.call('msc_adam_step', p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
              float(beta2), float(eps), float(weight_decay), int(step), float(grad_scale), None, _stream(p))
    return p
