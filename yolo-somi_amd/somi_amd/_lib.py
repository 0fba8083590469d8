"""ctypes binding of libsomi_hip.so (the C ABI declared in include/somi_hip.h).

The library is the product; this module only loads it, declares argument types and turns return
codes into exceptions.  There is no fallback: if the shared library is missing or an entry point
fails, a RuntimeError is raised (a reference-side `AT_ASSERTM` surfaces as RuntimeError too,
models/ops_dcnv3/src/cuda/dcnv3_cuda.cu:29-53).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SOMI_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libsomi_hip.so')   # override: kernel experiments

ABI_VERSION = 14         # SOMI_ABI_VERSION of include/somi_hip.h this binding was written against
c_f32p = C.c_void_p      # device pointers travel as integers
c_stream = C.c_void_p


class ConvDesc(C.Structure):
    """struct somi_conv_desc (include/somi_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in ('x', 'w', 'bias', 'post_scale', 'post_shift', 'residual', 'a_chan_scale',
                                          'a_pix_scale', 'y')] + \
               [(n, C.c_int32) for n in ('B', 'H', 'W', 'Cin', 'x_cs', 'x_coff', 'Ho', 'Wo', 'Cout', 'y_cs', 'y_coff',
                                         'kh', 'kw', 'stride', 'pad', 'dil', 'res_cs', 'res_coff', 'act', 'per_sample_w')] + \
               [('res2_cs', C.c_int32), ('res2_coff', C.c_int32), ('workspace', C.c_void_p), ('workspace_bytes', C.c_uint64),
                ('residual2', C.c_void_p), ('stat_sum', C.c_void_p), ('stat_sumsq', C.c_void_p), ('stat_pivot', C.c_void_p), ('prec', C.c_int32)]


class LossDesc(C.Structure):
    """struct somi_loss_desc (include/somi_hip.h)."""
    _fields_ = [('p', C.c_void_p * 4), ('grad', C.c_void_p * 4), ('ny', C.c_int32 * 4), ('nx', C.c_int32 * 4),
                ('nl', C.c_int32), ('na', C.c_int32), ('nc', C.c_int32), ('B', C.c_int32), ('nt', C.c_int32),
                ('targets', C.c_void_p), ('anchors', C.c_void_p), ('balance', C.c_float * 4),
                ('box_gain', C.c_float), ('obj_gain', C.c_float), ('cls_gain', C.c_float), ('cls_pw', C.c_float),
                ('obj_pw', C.c_float), ('anchor_t', C.c_float), ('cp', C.c_float), ('cn', C.c_float), ('gr', C.c_float),
                ('fl_gamma', C.c_float), ('slide', C.c_int32), ('nwd_ratio', C.c_float), ('nwd_constant', C.c_float)]


class AugSource(C.Structure):
    """struct somi_aug_source (include/somi_hip.h)."""
    _fields_ = [('pixels', C.c_void_p), ('h', C.c_int32), ('w', C.c_int32), ('x1', C.c_int32), ('y1', C.c_int32),
                ('x2', C.c_int32), ('y2', C.c_int32), ('dx', C.c_int32), ('dy', C.c_int32)]


class AugCanvas(C.Structure):
    """struct somi_aug_canvas."""
    _fields_ = [('src', AugSource * 4), ('nsrc', C.c_int32), ('height', C.c_int32), ('width', C.c_int32), ('warp', C.c_int32),
                ('minv', C.c_double * 6)]


class AugSample(C.Structure):
    """struct somi_aug_sample."""
    _fields_ = [('canvas', AugCanvas * 2), ('mix', C.c_int32), ('hsv', C.c_int32), ('flipud', C.c_int32), ('fliplr', C.c_int32),
                ('mix_r', C.c_double), ('lut', (C.c_uint8 * 256) * 3)]


I, F, P, S, Z, U64 = C.c_int, C.c_float, C.c_void_p, c_stream, C.c_size_t, C.c_uint64

# name -> (restype, argtypes); every symbol include/somi_hip.h declares
SIGNATURES = {
    'somi_abi_version': (I, []),
    'somi_last_error': (C.c_char_p, []),
    'somi_sizeof_desc': (Z, [I]),
    'somi_conv2d_nhwc_f32': (I, [C.POINTER(ConvDesc), S]),
    'somi_conv2d_workspace_bytes': (Z, []),
    'somi_conv2d_stat_rows': (I, [C.POINTER(ConvDesc)]),
    'somi_conv2d_dgrad_nhwc_f32': (I, [C.POINTER(ConvDesc), P, I, I, P, P, I, I, P, I, I, S]),
    'somi_conv2d_wgrad_workspace_bytes': (Z, [C.POINTER(ConvDesc)]),
    'somi_conv2d_wgrad_nhwc_f32': (I, [C.POINTER(ConvDesc), P, I, I, P, I, I, P, P, P, Z, S]),
    'somi_conv2d_kernel_name': (C.c_char_p, [C.POINTER(ConvDesc)]),
    'somi_dcnv3_forward_f32': (I, [P, P, P, P] + [I] * 13 + [F, I, S]),
    'somi_dcnv3_backward_workspace_bytes': (Z, [I] * 13 + [F]),
    'somi_dcnv3_backward_f32': (I, [P, P, P, P, P, P, P] + [I] * 13 + [F, I, P, Z, S]),
    'somi_dcnv3_forward_strided_f32': (I, [P, P, P, C.c_long, C.c_long, P] + [I] * 13 + [F, I, S]),
    'somi_dcnv3_backward_strided_f32': (I, [P, P, P, C.c_long, C.c_long, P, P, P, P] + [I] * 13 + [F, I, P, Z, S]),
    'somi_dcnv3_forward_f16': (I, [P, P, P, P] + [I] * 13 + [F, I, S]),
    'somi_dcnv3_backward_f16': (I, [P, P, P, P, P, P, P] + [I] * 13 + [F, I, P, Z, S]),
    'somi_dcnv3_forward_f64': (I, [P, P, P, P] + [I] * 13 + [F, I, S]),
    'somi_dcnv3_backward_f64': (I, [P, P, P, P, P, P, P] + [I] * 13 + [F, I, P, Z, S]),
    'somi_layernorm_act_nhwc_f32': (I, [P, P, P, F, I, P, C.c_long, I, S]),
    'somi_dwconv3x3_ln_nhwc_f32': (I, [P, P, P, P, P, F, I, P, P, I, I, I, I, S]),
    'somi_group_softmax_f32': (I, [P, P, C.c_long, I, S]),
    'somi_group_softmax_strided_f32': (I, [P, C.c_long, P, C.c_long, C.c_long, I, I, S]),
    'somi_group_softmax_bwd_strided_f32': (I, [P, C.c_long, P, P, C.c_long, C.c_long, I, I, S]),
    'somi_layernorm_act_bwd_workspace_floats': (Z, [C.c_long, I]),
    'somi_layernorm_gelu_bwd_nhwc_f32': (I, [P, P, P, F, P, P, P, P, P, C.c_long, I, S]),
    'somi_group_softmax_bwd_f32': (I, [P, P, P, C.c_long, I, S]),
    'somi_dcnv3_cfs_blend_bwd_f32': (I, [P, P, P, I, P, P, P, P, I, C.c_long, I, I, S]),
    'somi_dcnv3_cfs_blend_f32': (I, [P, P, P, I, P, C.c_long, I, I, S]),
    'somi_image_u8_to_nhwc4': (I, [P, P, I, I, I, I, S]),
    'somi_image_f32_to_nhwc4': (I, [P, P, I, I, I, I, F, S]),
    'somi_dwconv3x3_nhwc_f32': (I, [P, P, P, P, P, P, P, I, I, I, I, I, S]),
    'somi_sppf_pool_nhwc_f32': (I, [P, I, I, I, I, I, I, S]),
    'somi_bifpn_nhwc_f32': (I, [C.POINTER(C.c_void_p), C.POINTER(C.c_int), P, F, I, P, I, I, I, I, S]),
    'somi_pool_nchunk': (I, [I]),
    'somi_global_pool_nhwc_f32': (I, [P, I, I, I, I, I, P, P, P, S]),
    'somi_global_pool_act_nhwc_f32': (I, [P, I, I, I, I, I, I, P, P, P, P, S]),
    'somi_affine_silu_pool_rows': (I, [I, I, I]),
    'somi_affine_silu_pool_nhwc_f32': (I, [P, I, I, P, P, P, I, I, I, I, I, P, P, P, S]),
    'somi_attn_mlp_f32': (I, [I, P, P, P, P, P, P, P, I, I, I, S]),
    'somi_chan_stats_nhwc_f32': (I, [P, I, I, P, P, I, I, I, S]),
    'somi_spatial_attn_f32': (I, [P, P, P, P, I, I, I, I, S]),
    'somi_cbam_apply_nhwc_f32': (I, [P, I, I, P, P, P, P, P, I, I, I, I, I, I, I, S]),
    'somi_scale_channels_nhwc_f32': (I, [P, P, P, P, I, I, I, S]),
    'somi_odconv_weights_f32': (I, [P] * 18 + [I] * 7 + [S]),
    'somi_detect_decode_f32': (I, [P, I, P, I, C.POINTER(C.c_float), F, P, P, I, I, I, I, I, I, I, S]),
    'somi_red_nchunk': (I, [C.c_long]),
    'somi_bn_stats_nhwc_f32': (I, [P, I, I, C.c_long, I, F, F, P, P, P, P, P, P, P, P, P, S]),
    'somi_bn_stats_act_nhwc_f32': (I, [P, I, I, I, C.c_long, I, F, F, P, P, P, P, P, P, P, P, P, S]),
    'somi_bn_stats_partials_f32': (I, [P, P, I, C.c_long, I, F, F, P, P, P, P, P, P, P, P, P, S]),
    'somi_bn_local_sums_f64': (I, [P, I, I, C.c_long, I, P, P, P, I, P, P, S]),
    'somi_bn_stats_from_sums_f64': (I, [P, I, I, F, F, P, P, P, P, P, P, P, P, S]),
    'somi_bn_act_backward_sums_f64': (I, [P, I, I, P, I, I, P, P, P, I, I, C.c_long, I, P, P, S]),
    'somi_bn_act_backward_apply_sync_f32': (I, [P, I, I, P, I, I, P, P, P, P, I, I, P, P, I, P, I, I, P, P, C.c_long, I, P, S]),
    'somi_chan_affine_act_nhwc_f32': (I, [P, I, I, P, P, I, I, P, I, I, C.c_long, I, P, I, I, S]),
    'somi_bn_act_backward_nhwc_f32': (I, [P, I, I, P, I, I, P, P, P, P, I, I, I, P, I, I, P, P, C.c_long, I, P, S]),
    'somi_bn_pooled_rows': (I, [I, I]),
    'somi_bn_act_backward_pooled_nhwc_f32': (I, [P, I, I, P, I, I, P, P, P, P, I, I, P, P, P, P, I, I, P, P, I, I, I, P, S]),
    'somi_add_nhwc_f32': (I, [P, I, I, P, I, I, P, I, I, C.c_long, I, S]),
    'somi_chan_sum_nhwc_f32': (I, [P, I, I, C.c_long, I, P, P, S]),
    'somi_img_nchunk': (I, [I]),
    'somi_cbam_bwd_pixel_f32': (I, [P, I, I, P, I, I, P, P, P, P, I, I, I, S]),
    'somi_cbam_bwd_pixel_argmax_f32': (I, [P, I, I, P, I, I, P, P, P, P, P, P, I, I, I, S]),
    'somi_cbam_bn_bwd_workspace_floats': (C.c_size_t, [I, I, I]),
    'somi_cbam_bn_bwd_reduce_f32': (I, [P, I, I, P, I, I, P, P, P, P, P, P, P, P, P, P, I, I, I, S]),
    'somi_cbam_bn_bwd_apply_f32': (I, [P, I, I, P, I, I, P, P, P, P, P, P, P, P, P, P, P, P, I, I, P, P, P, I, I, I, S]),
    'somi_spatial_attn_bwd_f32': (I, [P, P, P, P, P, P, P, I, I, I, I, I, S]),
    'somi_cbam_bwd_chan_f32': (I, [P, I, I, P, I, I, P, P, P, P, P, P, I, I, I, S]),
    'somi_pool_argmax_nhwc_f32': (I, [P, I, I, I, I, I, P, P, S]),
    'somi_attn_mlp_bwd_workspace_floats': (Z, [I, I, I]),
    'somi_attn_mlp_bwd_f32': (I, [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, S]),
    'somi_pool_bwd_add_nhwc_f32': (I, [P, I, I, P, P, P, I, I, I, S]),
    'somi_detect_raw_bwd_f32': (I, [P, P, I, P, I, I, I, I, I, I, S]),
    'somi_resample_slice_nhwc_f32': (I, [P, I, I, P, I, I, I, I, I, I, I, I, I, S]),
    'somi_space_to_depth_nhwc_f32': (I, [P, I, I, P, I, I, I, I, I, I, I, S]),
    'somi_detect_plain_decode_f32': (I, [P, I, C.POINTER(C.c_float), F, P, P, I, I, I, I, I, I, I, S]),
    'somi_detect_plain_raw_bwd_f32': (I, [P, P, I, I, I, I, I, I, S]),
    'somi_tta_resample_nhwc4_f32': (I, [P, P, I, I, I, I, I, I, I, I, F, I, S]),
    'somi_tta_descale_f32': (I, [P, C.c_long, I, F, I, F, S]),
    'somi_sppf_pool_bwd_nhwc_f32': (I, [P, P, P, I, I, I, I, I, I, S]),
    'somi_sppf_pool_codes_nhwc_f32': (I, [P, P, I, I, I, I, I, I, S]),
    'somi_bifpn_bwd_nhwc_f32': (I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), P, F, I, P, P, P, I, I, I, I, S]),
    'somi_dwconv3x3_bwd_workspace_floats': (Z, [I, I, I]),
    'somi_dwconv3x3_bwd_nhwc_f32': (I, [P, P, P, P, P, P, P, P, I, I, I, I, S]),
    'somi_scale_channels_bwd_nhwc_f32': (I, [P, P, P, P, P, P, I, I, I, S]),
    'somi_linear_f32': (I, [P, I, P, P, I, P, I, I, I, I, I, S]),
    'somi_linear_bwd_f32': (I, [P, I, P, P, P, I, I, I, P, P, P, I, I, P, I, I, I, S]),
    'somi_odconv_synth_f32': (I, [P, P, P, P, P, I, I, I, I, I, I, S]),
    'somi_odconv_synth_bwd_workspace_floats': (Z, [I, I, I, I, I]),
    'somi_odconv_synth_bwd_f32': (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, S]),
    'somi_adam_ema_step_f32': (I, [P, P, P, P, P, C.c_long, F, F, F, F, F, I, F, S]),
    'somi_sgd_ema_step_f32': (I, [P, P, P, P, C.c_long, F, F, F, I, F, S]),
    'somi_pack_dgrad_weights_f32': (I, [P, P, I, I, I, S]),
    'somi_axpby_f32': (I, [P, P, C.c_long, F, F, S]),
    'somi_nms_workspace_bytes': (Z, [I, I, I, I]),
    'somi_nms_f32': (I, [P, I, I, I, F, F, I, I, P, I, P, P, P, Z, S]),
    'somi_loss_workspace_bytes': (Z, [C.POINTER(LossDesc)]),
    'somi_yolo_loss_f32': (I, [C.POINTER(LossDesc), P, P, Z, S]),
    'somi_val_match_f32': (I, [P, P, P, P, P, I, I, I, I, P, S]),
    'somi_confusion_matrix_f32': (I, [P, P, P, P, I, I, I, I, F, F, P, S]),
    'somi_ap_per_class_workspace_bytes': (Z, [C.c_long, I, I]),
    'somi_ap_per_class_f64': (I, [P, P, P, P, C.c_long, C.c_long, I, I, P, P, P, P, P, P, P, Z, S]),
    'somi_repulsion_workspace_bytes': (Z, [I, I]),
    'somi_repulsion_loss_f32': (I, [P, P, P, I, I, F, F, F, F, P, P, Z, S]),
    'somi_wbf_workspace_bytes': (Z, [I]),
    'somi_augment_u8': (I, [P, I, I, I, I, P, S]),
    'somi_wbf_f32': (I, [P, P, P, P, I, I, C.POINTER(C.c_float), F, F, P, P, P, P, P, Z, S]),
    'somi_wbf_batch_workspace_bytes': (Z, [I, I, I]),
    'somi_wbf_batch_f32': (I, [C.POINTER(P), C.POINTER(P), I, I, I, C.POINTER(C.c_float), F, F, F, F, P, P, P, P, P, Z, S]),
}

_lib = None


def lib():
    """The loaded library (loads on first use; raises if it is not built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                               f'(or `make -C yolo-somi_amd/csrc`). There is no CPU fallback.')
        # torch first: its wheel carries its own libamdhip64.so (SONAME libamdhip64.so.7) and asks for it by the name "libamdhip64.so".  Loaded
        # before torch, this library would pull /opt/rocm's copy under its SONAME, torch's request would not match that name and a SECOND HIP
        # runtime would come up in the process - whose launches then fail with "no ROCm-capable device is detected".  With torch's copy mapped
        # first, the SONAME this library asks for resolves to it and there is one runtime.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        lax = bool(os.environ.get('SOMI_HIP_LIB')) and os.environ.get('SOMI_HIP_LIB_LAX') == '1'   # kernel A/B runs against an older build
        for name, (res, args) in SIGNATURES.items():
            if lax and not hasattr(L, name):
                continue
            fn = getattr(L, name)            # AttributeError here = header/library mismatch: fail loudly
            fn.restype, fn.argtypes = res, args
        if L.somi_abi_version() != ABI_VERSION and not lax:
            raise RuntimeError('libsomi_hip.so ABI version mismatch')
        if not lax and (L.somi_sizeof_desc(0) != C.sizeof(ConvDesc) or L.somi_sizeof_desc(1) != C.sizeof(LossDesc)):
            raise RuntimeError('libsomi_hip.so descriptor layout differs from this binding')
        _lib = L
    return _lib


def check(rc, what=''):
    if rc != 0:
        msg = lib().somi_last_error().decode(errors='replace')
        raise RuntimeError(f'{what or "somi"} failed (code {rc}): {msg}')
