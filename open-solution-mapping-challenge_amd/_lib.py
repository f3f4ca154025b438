"""ctypes binding of libmsc_hip.so (C ABI: include/msc.h).

There is deliberately NO fallback: if the HIP library has not been built, importing any compute
module of this package raises.  Build it with `python __graft_entry__.py build` (or
`make -C open-solution-mapping-challenge_amd/csrc`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MSC_HIP_LIB', os.path.join(_HERE, 'lib', 'libmsc_hip.so'))

F32, BF16, F16 = 0, 1, 2
BN_SLOTS = 8                      # MSC_BN_SLOTS: per-XCD accumulation slots of the BatchNorm sums
CFG_HALO, CFG_HALO_T = 27, 28     # msc_conv_igemm configurations that are halo-tile kernels, not tiles of the DMA kernel
CFG_STREAM = 57                   # ... the persistent streaming kernel for 1x1 / stride 1 layers
CFG_STEM, CFG_DOWN4 = 58, 59      # ... the stem's 7x7 / stride 2 halo kernel; Conv2d(k4, s2, p1) 32 -> 128 (the data gradient of dec1's ConvTranspose2d, round 6)


class MscError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [('in_', C.c_void_p), ('wt', C.c_void_p), ('out', C.c_void_p), ('res', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('stats', C.c_void_p),
                ('in_ld', C.c_int64), ('out_ld', C.c_int64), ('res_ld', C.c_int64),
                ('dtype', C.c_int32), ('mode', C.c_int32),
                ('N', C.c_int32), ('Hi', C.c_int32), ('Wi', C.c_int32), ('Cin', C.c_int32),
                ('Ho', C.c_int32), ('Wo', C.c_int32), ('Cout', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32),
                ('stride', C.c_int32), ('pad', C.c_int32), ('flip', C.c_int32), ('relu', C.c_int32), ('cfg', C.c_int32),
                ('stats_kind', C.c_int32), ('stats_y', C.c_void_p), ('stats_y_ld', C.c_int64),
                ('final_w', C.c_void_p), ('final_b', C.c_void_p), ('final_logits', C.c_void_p), ('final_probs', C.c_void_p),
                ('final_skip_store', C.c_int32), ('splitk', C.c_int32), ('splitk_ws', C.c_void_p),
                ('stats_z', C.c_void_p), ('stats_z_ld', C.c_int64),        # ABI v6: stats_kind 1 masked by a stored activation, residual allowed
                ('stats_z_bits', C.c_int32), ('reserved0', C.c_int32),     # ABI v7: stats_z is msc_bn_apply's ReLU byte mask
                ('in_bn', C.c_void_p)]                                     # ABI v9: BnInput* -- BatchNorm + ReLU of the input applied on load


class BnInput(C.Structure):
    _fields_ = [('slots', C.c_void_p), ('count', C.c_int64), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('eps', C.c_float), ('momentum', C.c_float), ('running_mean', C.c_void_p), ('running_var', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p), ('save_mean', C.c_void_p), ('save_invstd', C.c_void_p),
                ('out', C.c_void_p), ('out_ld', C.c_int64)]


class WgradDesc(C.Structure):
    _fields_ = [('p', C.c_void_p), ('q', C.c_void_p), ('dw', C.c_void_p),
                ('p_ld', C.c_int64), ('q_ld', C.c_int64), ('dtype', C.c_int32),
                ('N', C.c_int32), ('Hp', C.c_int32), ('Wp', C.c_int32), ('A', C.c_int32),
                ('Hq', C.c_int32), ('Wq', C.c_int32), ('B', C.c_int32), ('KH', C.c_int32), ('KW', C.c_int32),
                ('stride', C.c_int32), ('pad', C.c_int32), ('cfg', C.c_int32)]


class BneckDesc(C.Structure):             # == msc_bneck_desc
    _fields_ = [('x', C.c_void_p), ('out', C.c_void_p), ('wpk', C.c_void_p),
                ('scale1', C.c_void_p), ('shift1', C.c_void_p), ('scale2', C.c_void_p), ('shift2', C.c_void_p),
                ('scale3', C.c_void_p), ('shift3', C.c_void_p),
                ('x_ld', C.c_int64), ('out_ld', C.c_int64),
                ('dtype', C.c_int32), ('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32), ('Cmid', C.c_int32), ('cfg', C.c_int32)]


class BiasSlotsItem(C.Structure):         # == msc_bias_slots_item
    _fields_ = [('slots', C.c_void_p), ('db', C.c_void_p), ('Cs', C.c_int32), ('C', C.c_int32)]


BIAS_SLOTS_MAX = 16

# optimizer state in device memory (include/msc.h, ABI v7): f32[OPT_STATE]
OPT_STEP, OPT_LR, OPT_OVERFLOW, OPT_SKIP, OPT_SCALE, OPT_GOOD, OPT_GROWTH, OPT_SKIPPED, OPT_UNSCALE = range(9)
OPT_STATE = 12
WGRAD_ORDERED = 1            # msc_wgrad_group_create flags
FINAL_BWD_WS_ROWS = 1024     # msc_final_bwd ordered_ws rows


class LossCfg(C.Structure):
    _fields_ = [('w0', C.c_float), ('sigma', C.c_float), ('size_c', C.c_float),
                ('dice_weight', C.c_float), ('ce_weight', C.c_float), ('smooth', C.c_float), ('eps', C.c_float),
                ('weighted', C.c_int32), ('dice_sigmoid', C.c_int32)]


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double

# name -> (restype, argtypes); every symbol declared in include/msc.h
SIGNATURES = {
    'msc_last_error': (C.c_char_p, []),
    'msc_abi_version': (_i, []),
    'msc_conv_igemm': (_i, [C.POINTER(ConvDesc), _vp]),
    'msc_conv_stats_slices': (_i, [C.POINTER(ConvDesc)]),
    'msc_conv_num_cfgs': (_i, []),
    'msc_conv_cfg_ok': (_i, [C.POINTER(ConvDesc), _i]),
    'msc_conv_wgrad': (_i, [C.POINTER(WgradDesc), _vp]),
    'msc_conv_wgrad_num_cfgs': (_i, []),
    'msc_wgrad_group_create': (_i, [C.POINTER(WgradDesc), _i, _i, _i, _i, C.POINTER(_vp)]),
    'msc_wgrad_group_run': (_i, [_vp, _vp]),
    'msc_wgrad_group_launches': (_i, [_vp]),
    'msc_wgrad_group_run_part': (_i, [_vp, _i, _vp]),
    'msc_wgrad_group_destroy': (None, [_vp]),
    'msc_pack_cast': (_i, [_vp, _vp, _i, _i64, _vp]),
    'msc_pack_transpose': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_pack_multi': (_i, [_vp, _vp, _vp, _i, _i, _vp]),
    'msc_stem_pack': (_i, [_vp, _vp, _i, _i, _vp]),
    'msc_grad_reduce': (_i, [_vp, _vp, _i, _i, _i64, _vp]),
    'msc_grad_unpack': (_i, [_vp, _vp, _i, _i64, _vp]),
    'msc_stem_unpack_grad': (_i, [_vp, _vp, _i, _vp]),
    'msc_stem_prepare': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_maxpool2_fwd': (_i, [_vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp]),
    'msc_maxpool2_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_bn_fold': (_i, [_vp, _vp, _vp, _vp, _f, _vp, _vp, _i, _vp]),
    'msc_bottleneck_ok': (_i, [C.POINTER(BneckDesc)]),
    'msc_bottleneck_pack_bytes': (_i64, [_i]),
    'msc_bottleneck_pack': (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp]),
    'msc_bottleneck_fused': (_i, [C.POINTER(BneckDesc), _vp]),
    'msc_annotations_json': (_i64, [_vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i64]),
    'msc_memset_zero': (_i, [_vp, _i64, _vp]),
    'msc_copy': (_i, [_vp, _vp, _i64, _vp]),
    'msc_bn_apply': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i64, _i, _vp]),
    'msc_bn_bwd_reduce': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _i, _i64, _i, _vp]),
    'msc_bn_bwd_apply': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, _i64, _vp, _i, _i64, _i, _vp]),
    'msc_bn_apply_pool': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msc_bn_pool_bwd_reduce': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msc_bn_pool_bwd_apply': (_i, [_vp, _i64, _vp, _i64, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msc_relu_bwd': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i64, _i, _vp]),
    'msc_bias_grad_workspace_bytes': (_i64, [_i64, _i, _i]),
    'msc_bias_grad': (_i, [_vp, _i64, _vp, _vp, _i, _i64, _i, _vp]),
    'msc_relu_bias_grad': (_i, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _i, _i64, _i, _vp]),
    'msc_final_fwd': (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msc_bias_slots_finalize': (_i, [_vp, _i, _vp, _i, _vp]),
    'msc_bias_slots_finalize_multi': (_i, [C.POINTER(BiasSlotsItem), _i, _vp]),
    'msc_final_bwd': (_i, [_vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'msc_loss_sums': (_i, [_vp, _vp, _i, C.POINTER(LossCfg), _vp, _i, _i, _i, _vp]),
    'msc_loss_grad': (_i, [_vp, _vp, _i, C.POINTER(LossCfg), _vp, _d, _f, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'msc_adam_step': (_i, [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    'msc_adam_tick': (_i, [_vp, _vp]),
    'msc_grad_check': (_i, [_vp, _i64, _vp, _vp]),
    'msc_adam_pack': (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    'msc_resize_bilinear': (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_resize_threshold': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'msc_crop_center': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_threshold_layers': (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    'msc_argmax_channels': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_erode_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_dilate_i32': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_label_workspace_bytes': (_i64, [_i, _i, _i]),
    'msc_label4': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'msc_add_dropped': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_watershed_workspace_bytes': (_i64, [_i, _i, _i]),
    'msc_watershed': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'msc_build_score': (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_crf_workspace_bytes': (_i64, [_i, _i, _i, _i]),
    'msc_tta_transform': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_tta_aggregate': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_rle_segments_workspace': (_i64, [_i, _i, _i]),
    'msc_rle_segments': (_i, [_vp, _i, _i, _i, _vp, _i64, C.POINTER(C.c_int32), _vp]),
    'msc_rle_encode_workspace': (_i64, [_i]),
    'msc_prep_workspace_bytes': (_i64, [_i, _i, _i]),
    'msc_prep_targets': (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'msc_prep_morph_workspace_bytes': (_i64, [_i, _i, _i]),
    'msc_prep_morph': (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp]),
    'msc_prep_paint': (_i, [_vp, _vp, _i, _i, _i, _vp]),
    'msc_rect_filter_u8': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'msc_prep_border': (_i, [_vp, _vp, C.c_double, _vp, _i, _i, _vp]),
    'msc_size_matrix': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'msc_rle_encode': (_i, [_vp, _i, _i, _i, _i, _vp, _i64, C.POINTER(C.c_int32), C.POINTER(C.c_int64), C.POINTER(_vp), C.POINTER(_vp), _vp]),
    'msc_dense_crf': (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _f, _f, _f, _f, _f, _i, _vp]),
}

_lib = None


def load():
    """Load libmsc_hip.so and bind every symbol of include/msc.h; raises MscError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise MscError('HIP library not built: %s is missing. Run `python __graft_entry__.py build`. '
                       'There is no CPU fallback for the product path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError -> missing export: fail loudly
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=''):
    if rc != 0:
        msg = load().msc_last_error()
        raise MscError('%s failed (rc=%d): %s' % (what or 'msc call', rc, msg.decode() if msg else '?'))


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
