"""Torch-tensor wrappers over the C ABI (device pointers + current HIP stream).

PyTorch is plumbing here: it owns the HBM allocations and the stream; all arithmetic happens in
libsomi_hip.so.  Tensors must live on the GPU; anything else raises.
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvDesc, check

STREAMK = os.environ.get('SOMI_CONV_STREAMK', '1') != '0'
FUSED_BN_STATS = os.environ.get('SOMI_FUSED_BN_STATS', '1') != '0'   # batch statistics from the conv epilogue (training forward)
DCN_DIRECT = os.environ.get('SOMI_DCN_DIRECT', '0') == '1'          # 1: DCNv3 backward scatters with fp32 atomics (the reference's form)
PROFILE = None   # bench.py sets this to a list: every conv launch appends (kernel name, algorithmic FLOPs, ev0, ev1)

ACT = {'none': 0, None: 0, 'silu': 1, 'gelu': 2, 'relu': 3, 'sigmoid': 4, 'softmax': 5}


_CONV_WS = {}
# Opt-in reduced precision of the conv family's products (train.py:263 `amp.autocast`): 0 exact fp32 (the default, what every parity
# claim is made on), 1 bf16 operands, 2 bf16x3 split (somi_conv_desc.prec).  train.TrainStep(amp=...) / bench.py --amp set it.
CONV_PREC = 0
PREC = {None: 0, 'f32': 0, 'bf16': 1, 'bf16x3': 2}
FUSE_POOL = os.environ.get('SOMI_FUSE_POOL', '1') != '0'      # 0: the channel attention pools its input in a pass of its own (round-3 form; A/B runs)
AMAX_BY_VALUE = os.environ.get('SOMI_AMAX_BY_VALUE', '1') != '0'   # 0: the max-pool's arg-max from somi_pool_argmax_nhwc_f32's own pass over the tensor
ODCONV_INPLACE = os.environ.get('SOMI_ODCONV_INPLACE', '1') != '0'   # 0: ODConv's squeeze pools its input itself; its input gradient goes through a tensor of its own + add passes (A/B runs)
FUSE_DWLN = os.environ.get('SOMI_FUSE_DWLN', '1') != '0'    # 0: the DCNv3 block's depthwise conv and LayerNorm + GELU as two passes (A/B runs)
CBAM_FUSED_BN = os.environ.get('SOMI_CBAM_FUSED_BN', '1') != '0'   # 0: CBAM's step C as a pass of its own before the BatchNorm backward (A/B runs)
BN_POOLED = os.environ.get('SOMI_BN_POOLED', '1') != '0'    # 0: CBAM's pooled gradients are added by a pass of their own (round-3 form; A/B runs)


def _conv_workspace(d, dev):
    """Attach the stream-K scratch of the current stream to a conv descriptor (launches on one stream are ordered, so they
    can share it).  SOMI_CONV_STREAMK=0 keeps the one-workgroup-per-tile schedule."""
    if not STREAMK:
        return
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    ws = _CONV_WS.get(key)
    if ws is None:
        ws = _CONV_WS[key] = torch.empty(_lib.lib().somi_conv2d_workspace_bytes(), dtype=torch.uint8, device=dev)
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel()


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('somi_amd ops need GPU tensors (no CPU fallback)')
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name='tensor'):
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError(f'{name} tensor has to be contiguous float32')
    return t


def conv_out_size(size, k, s, p, d=1):
    return (size + 2 * p - (d * (k - 1) + 1)) // s + 1


def conv2d_nhwc(x, w, bias=None, *, kh, kw, stride=1, pad=0, dil=1, act='none', cin=None, x_coff=0, out=None,
                cout=None, y_coff=0, post_scale=None, post_shift=None, residual=None, res_coff=0, a_chan_scale=None,
                a_pix_scale=None, per_sample_w=False, alg_cin=None, alg_cout=None, bn_stats=None):
    """x (B,H,W,x_cs) NHWC; w packed [n_sets][Cout][kh*kw*Cin]; returns / fills out (B,Ho,Wo,y_cs)."""
    B, H, W, x_cs = x.shape
    cin = x_cs - x_coff if cin is None else cin
    cout = w.shape[-2] if cout is None else cout
    Ho, Wo = conv_out_size(H, kh, stride, pad, dil), conv_out_size(W, kw, stride, pad, dil)
    if out is None:
        out = torch.empty(B, Ho, Wo, cout, device=x.device, dtype=torch.float32)
    d = ConvDesc()
    d.prec = CONV_PREC
    d.x, d.w, d.bias, d.y = _ptr(_f32c(x, 'input')), _ptr(_f32c(w, 'weight')), _ptr(bias), _ptr(_f32c(out, 'output'))
    d.post_scale, d.post_shift, d.residual = _ptr(post_scale), _ptr(post_shift), _ptr(residual)
    d.a_chan_scale, d.a_pix_scale = _ptr(a_chan_scale), _ptr(a_pix_scale)
    d.B, d.H, d.W, d.Cin, d.x_cs, d.x_coff = B, H, W, cin, x_cs, x_coff
    d.Ho, d.Wo, d.Cout, d.y_cs, d.y_coff = Ho, Wo, cout, out.shape[3], y_coff
    d.kh, d.kw, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
    _conv_workspace(d, x.device)
    d.res_cs, d.res_coff = (residual.shape[3] if residual is not None else 0), res_coff
    d.act, d.per_sample_w = ACT[act], int(per_sample_w)
    if w.numel() != (B if per_sample_w else 1) * cout * kh * kw * cin:
        raise RuntimeError(f'weight has {w.numel()} elements, expected {(B if per_sample_w else 1) * cout * kh * kw * cin}')
    if bn_stats is not None:                                      # dict filled in place: the epilogue leaves partial channel sums
        rows = _lib.lib().somi_conv2d_stat_rows(C.byref(d))
        part = torch.empty(2, rows, cout, device=x.device, dtype=torch.float32)
        d.stat_sum, d.stat_sumsq, d.stat_pivot = part[0].data_ptr(), part[1].data_ptr(), _ptr(bn_stats.get('pivot'))
        bn_stats['part'], bn_stats['rows'] = part, rows
    if PROFILE is None:
        check(_lib.lib().somi_conv2d_nhwc_f32(C.byref(d), _stream()), 'conv2d_nhwc')
        return out
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()                                               # torch's current stream == the stream the kernel runs on
    check(_lib.lib().somi_conv2d_nhwc_f32(C.byref(d), _stream()), 'conv2d_nhwc')
    e1.record()
    name = _lib.lib().somi_conv2d_kernel_name(C.byref(d)).decode()
    flops = 2.0 * B * Ho * Wo * (alg_cout or cout) * (alg_cin or cin) * kh * kw
    PROFILE.append((name, flops, e0, e1, (B, H, W, cin, cout, kh, stride, int(per_sample_w))))
    return out


_DCN_SUFFIX = {torch.float32: 'f32', torch.float16: 'f16', torch.float64: 'f64'}


def dcnv3_forward_raw(input, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale,
                      im2col_step):
    N, H, W, _ = input.shape
    Ho, Wo = conv_out_size(H, kh, sh, ph, dh), conv_out_size(W, kw, sw, pw, dw)
    if input.dtype != torch.float32:                              # half / double: the reference's other two dispatch types
        out = torch.empty(N, Ho, Wo, group * group_channels, device=input.device, dtype=input.dtype)
        fn = getattr(_lib.lib(), 'somi_dcnv3_forward_' + _DCN_SUFFIX[input.dtype])
        check(fn(_ptr(input), _ptr(offset), _ptr(mask), _ptr(out), N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw,
                 float(offset_scale), int(im2col_step), _stream()), 'dcnv3_forward')
        return out
    out = torch.empty(N, Ho, Wo, group * group_channels, device=input.device, dtype=torch.float32)
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().somi_dcnv3_forward_f32(_ptr(input), _ptr(offset), _ptr(mask), _ptr(out), N, H, W, group,
                                            group_channels, kh, kw, sh, sw, ph, pw, dh, dw, float(offset_scale),
                                            int(im2col_step), _stream()), 'dcnv3_forward')
    if prof:                                                      # algorithmic bytes: 4 (2C + 3GK) per output pixel
        e1.record()
        C_ = group * group_channels
        PROFILE.append(('dcnv3_fwd_kernel', 4.0 * N * Ho * Wo * (2 * C_ + 3 * group * kh * kw), e0, e1, (N, H, W, C_, group, kh, sh, 10)))
    return out


DCN_LAST_WORKSPACE = None
_DCN_WS = {}                                                     # device -> the windowed backward's workspace (grow-only: the staging slab, <= SOMI_DCN_SLAB_MB = 1 GiB by default, + near masks)
_DCN_FAR = {}                                                    # device -> int64 device counter: far taps of EVERY windowed backward since the last reset


def dcn_overflow_taps(total=False):
    """FAR sampling taps of the windowed DCNv3 backward - those that went to grad_input as fp32 atomics because they landed more than
    two tiles from their own (0 <=> bit-reproducible).  Default: the last call only.  total=True: summed over every call on every device
    since reset_dcn_overflow_taps() - a graph has several DCNv3 sites sharing one workspace whose counter word each call zeroes again, so
    "no float atomic ran in this step" is a statement about the total.  Host sync; diagnostics / tests."""
    if total:
        return int(sum(int(t.item()) for t in _DCN_FAR.values()))
    ws = DCN_LAST_WORKSPACE
    return None if ws is None else int(ws[-256:-252].view(torch.int32).item())


def reset_dcn_overflow_taps():
    for t in _DCN_FAR.values():
        t.zero_()


def _dcn_count_far(ws):
    """Adds the call's far-tap word to the device's running total (one tiny stream-ordered add, no sync)."""
    tot = _DCN_FAR.get(ws.device)
    if tot is None:
        tot = _DCN_FAR[ws.device] = torch.zeros(1, dtype=torch.int64, device=ws.device)
    tot.add_(ws[-256:-252].view(torch.int32))


def free_workspaces():
    """Releases the per-device DCNv3 backward workspace (up to SOMI_DCN_SLAB_MB + near masks; it is otherwise kept for the life of the
    process and only grows).  The next windowed backward allocates it again."""
    global DCN_LAST_WORKSPACE
    DCN_LAST_WORKSPACE = None
    _DCN_WS.clear()


def _dcn_workspace(nbytes, dev):
    """One buffer per device serves every windowed backward: the library caps the staging slab (SOMI_DCN_SLAB_MB, 1024 by default - a memory
    bound, see dcnv3.hip) and walks the batch through it in chunks of images (round 2 allocated up to 2.95 GB per call and kept it in a global)."""
    ws = _DCN_WS.get(dev)
    if ws is None or ws.numel() < nbytes:
        ws = _DCN_WS[dev] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws[:nbytes]


def dcnv3_backward_raw(input, offset, mask, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels,
                       offset_scale, im2col_step):
    N, H, W, _ = input.shape
    if input.dtype != torch.float32:
        # half: fp32 gradient buffers cast back to half afterwards, exactly as dcnv3_cuda.cu:126-133,168-170; double: double
        gdt = torch.float64 if input.dtype == torch.float64 else torch.float32
        gi = torch.zeros(input.shape, device=input.device, dtype=gdt)
        go = torch.empty(offset.shape, device=input.device, dtype=gdt)
        gm = torch.empty(mask.shape, device=input.device, dtype=gdt)
        fn = getattr(_lib.lib(), 'somi_dcnv3_backward_' + _DCN_SUFFIX[input.dtype])
        check(fn(_ptr(input), _ptr(offset), _ptr(mask), _ptr(grad_output), _ptr(gi), _ptr(go), _ptr(gm), N, H, W, group, group_channels, kh, kw,
                 sh, sw, ph, pw, dh, dw, float(offset_scale), int(im2col_step), None, 0, _stream()), 'dcnv3_backward')
        return gi.to(input.dtype), go.to(input.dtype), gm.to(input.dtype)
    gi = torch.zeros_like(input)
    go = torch.empty_like(offset)
    gm = torch.empty_like(mask)
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L = _lib.lib()
    nbytes = 0 if DCN_DIRECT else L.somi_dcnv3_backward_workspace_bytes(N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw, float(offset_scale))
    ws = _dcn_workspace(nbytes, input.device) if nbytes else None   # staging slab + near masks + overflow word of the windowed form
    global DCN_LAST_WORKSPACE
    DCN_LAST_WORKSPACE = ws                                      # tests read the overflow-tap counter at its end (dcn_overflow_taps)
    check(L.somi_dcnv3_backward_f32(_ptr(input), _ptr(offset), _ptr(mask), _ptr(grad_output), _ptr(gi), _ptr(go),
                                    _ptr(gm), N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw,
                                    float(offset_scale), int(im2col_step), _ptr(ws), nbytes, _stream()), 'dcnv3_backward')
    if prof:                                                      # 4 (4C + 6GK) per output pixel
        e1.record()
        C_ = group * group_channels
        Ho, Wo = grad_output.shape[1], grad_output.shape[2]
        PROFILE.append(('dcnv3_bwd_kernel', 4.0 * N * Ho * Wo * (4 * C_ + 6 * group * kh * kw), e0, e1, (N, H, W, C_, group, kh, sh, 11)))
    if ws is not None:
        _dcn_count_far(ws)
    return gi, go, gm


def dcnv3_forward_merged(input, om, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale, im2col_step):
    """The fp32 operator over ONE tensor om (N,Ho,Wo,R >= 3*G*K) holding [2*G*K offsets | G*K masks (already soft-maxed) | pad] per pixel -
    the output of a single 1x1 GEMM with the two Linear weights stacked (modules/dcnv3.py:330-334).  Same arithmetic as dcnv3_forward_raw."""
    N, H, W, _ = input.shape
    Ho, Wo = conv_out_size(H, kh, sh, ph, dh), conv_out_size(W, kw, sw, pw, dw)
    R, GK = om.shape[-1], group * kh * kw
    if om.shape[:3] != (N, Ho, Wo) or R < 3 * GK or not om.is_contiguous() or om.dtype != torch.float32:
        raise RuntimeError(f'dcnv3 (merged offset/mask): om must be contiguous float32 ({N},{Ho},{Wo},>={3 * GK}), got {tuple(om.shape)}')
    out = torch.empty(N, Ho, Wo, group * group_channels, device=input.device, dtype=torch.float32)
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().somi_dcnv3_forward_strided_f32(_ptr(input), om.data_ptr(), om.data_ptr() + 8 * GK, R, R, _ptr(out), N, H, W, group,
                                                    group_channels, kh, kw, sh, sw, ph, pw, dh, dw, float(offset_scale), int(im2col_step),
                                                    _stream()), 'dcnv3_forward')
    if prof:
        e1.record()
        C_ = group * group_channels
        PROFILE.append(('dcnv3_fwd_kernel', 4.0 * N * Ho * Wo * (2 * C_ + 3 * GK), e0, e1, (N, H, W, C_, group, kh, sh, 10)))
    return out


def dcnv3_backward_merged(input, om, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, group, group_channels, offset_scale, im2col_step):
    """-> (grad_input, d_om): d_om has om's layout - grad_offset in the first 2*G*K columns, grad_mask (w.r.t. the soft-maxed mask) in the
    next G*K, zeros in the pad."""
    N, H, W, _ = input.shape
    R, GK = om.shape[-1], group * kh * kw
    gi = torch.zeros_like(input)
    d_om = torch.empty_like(om)
    if R != 3 * GK:
        d_om[..., 3 * GK:].zero_()                                # the operator writes the first 3GK columns of every row; the pad stays zero
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L = _lib.lib()
    nbytes = 0 if DCN_DIRECT else L.somi_dcnv3_backward_workspace_bytes(N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw, dh, dw, float(offset_scale))
    ws = _dcn_workspace(nbytes, input.device) if nbytes else None
    global DCN_LAST_WORKSPACE
    DCN_LAST_WORKSPACE = ws
    check(L.somi_dcnv3_backward_strided_f32(_ptr(input), om.data_ptr(), om.data_ptr() + 8 * GK, R, R, _ptr(grad_output), _ptr(gi),
                                            d_om.data_ptr(), d_om.data_ptr() + 8 * GK, N, H, W, group, group_channels, kh, kw, sh, sw, ph, pw,
                                            dh, dw, float(offset_scale), int(im2col_step), _ptr(ws), nbytes, _stream()), 'dcnv3_backward')
    if prof:
        e1.record()
        C_ = group * group_channels
        Ho, Wo = grad_output.shape[1], grad_output.shape[2]
        PROFILE.append(('dcnv3_bwd_kernel', 4.0 * N * Ho * Wo * (4 * C_ + 6 * GK), e0, e1, (N, H, W, C_, group, kh, sh, 11)))
    if ws is not None:
        _dcn_count_far(ws)
    return gi, d_om


def group_softmax_cols_(om, G, K, col):
    """In place: softmax over each of the G groups of K columns starting at column `col` of every row of om (..., R)."""
    R = om.shape[-1]
    p = om.data_ptr() + 4 * col
    check(_lib.lib().somi_group_softmax_strided_f32(p, R, p, R, om.numel() // R, G, K, _stream()), 'group_softmax')
    return om


def group_softmax_backward_cols_(om, d_om, G, K, col):
    """In place on d_om's columns [col, col + G*K): dmask -> dlogits, with the soft-maxed masks read from the same columns of om."""
    R = om.shape[-1]
    check(_lib.lib().somi_group_softmax_bwd_strided_f32(om.data_ptr() + 4 * col, R, d_om.data_ptr() + 4 * col, d_om.data_ptr() + 4 * col, R,
                                                        om.numel() // R, G, K, _stream()), 'group_softmax_bwd')
    return d_om


def image_to_nhwc4(img, scale=1.0 / 255.0):
    """(B,C<=4,H,W) uint8 or float32 NCHW -> (B,H,W,4) float32 (train.py:249 `/255`)."""
    B, Cc, H, W = img.shape
    if not img.is_contiguous():
        raise RuntimeError('image batch has to be contiguous NCHW')
    y = torch.empty(B, H, W, 4, device=img.device, dtype=torch.float32)
    L = _lib.lib()
    if img.dtype == torch.uint8:
        if abs(scale - 1.0 / 255.0) > 1e-12:
            raise RuntimeError('uint8 images are always scaled by 1/255')
        check(L.somi_image_u8_to_nhwc4(_ptr(img), _ptr(y), B, Cc, H, W, _stream()), 'image_u8_to_nhwc4')
    elif img.dtype == torch.float32:
        check(L.somi_image_f32_to_nhwc4(_ptr(img), _ptr(y), B, Cc, H, W, float(scale), _stream()), 'image_f32_to_nhwc4')
    else:
        raise RuntimeError('images must be uint8 or float32')
    return y


def dwconv3x3(x, w, bias=None, post_scale=None, post_shift=None, residual=None, act='none', out=None):
    B, H, W, Cc = x.shape
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().somi_dwconv3x3_nhwc_f32(_ptr(_f32c(x)), _ptr(w), _ptr(bias), _ptr(post_scale), _ptr(post_shift),
                                             _ptr(residual), _ptr(out), B, H, W, Cc, ACT[act], _stream()), 'dwconv3x3')
    return out


def dwconv3x3_ln(x, w, bias, gamma, beta, eps, act='gelu'):
    """(u, y): u = depthwise 3x3 conv of x (+bias), y = act(LayerNorm_C(u)) - one pass (256 channels; somi_dwconv3x3_ln_nhwc_f32)."""
    B, H, W, Cc = x.shape
    u, y = torch.empty_like(x), torch.empty_like(x)
    check(_lib.lib().somi_dwconv3x3_ln_nhwc_f32(_ptr(_f32c(x)), _ptr(w), _ptr(bias), _ptr(gamma), _ptr(beta), float(eps), ACT[act], _ptr(u), _ptr(y),
                                                B, H, W, Cc, _stream()), 'dwconv3x3_ln')
    return u, y


def sppf_pool_(buf, c, x_coff=0, codes=False):
    """The three chained 5x5 pools of slice [x_coff, x_coff + c) into the next three slices of `buf`.  codes=True (training): level by level, and
    -> (buf, codes): where each window's first maximum sits, for sppf_pool_backward_(codes=...)."""
    B, H, W, cs = buf.shape
    if codes:
        arg = torch.empty(3 * B * H * W * c, device=buf.device, dtype=torch.uint8)
        check(_lib.lib().somi_sppf_pool_codes_nhwc_f32(_ptr(_f32c(buf)), _ptr(arg), B, H, W, c, cs, x_coff, _stream()), 'sppf_pool_codes')
        return buf, arg
    check(_lib.lib().somi_sppf_pool_nhwc_f32(_ptr(_f32c(buf)), B, H, W, c, cs, x_coff, _stream()), 'sppf_pool')
    return buf


def bifpn(srcs, ups, w_dev, eps=1e-4, out=None):
    """y = sum_i w_i / (sum_j swish(w_j) + eps) * src_i; `w_dev` is the raw fusion parameter on the device."""
    n = len(srcs)
    B, H, W, Cc = srcs[0].shape
    H, W = H << ups[0], W << ups[0]
    out = torch.empty(B, H, W, Cc, device=srcs[0].device, dtype=torch.float32) if out is None else out
    ptrs = (C.c_void_p * n)(*[_ptr(_f32c(s)) for s in srcs])
    check(_lib.lib().somi_bifpn_nhwc_f32(ptrs, (C.c_int * n)(*ups), _ptr(_f32c(w_dev)), float(eps), n, _ptr(out),
                                         B, H, W, Cc, _stream()), 'bifpn')
    return out


def global_pool(x, c=None, x_coff=0, want_max=True):
    B, H, W, cs = x.shape
    c = cs - x_coff if c is None else c
    nchunk = _lib.lib().somi_pool_nchunk(H * W)
    ws = torch.empty(2 * B * nchunk * c, device=x.device, dtype=torch.float32)
    avg = torch.empty(B, c, device=x.device, dtype=torch.float32)
    mx = torch.empty(B, c, device=x.device, dtype=torch.float32) if want_max else None
    check(_lib.lib().somi_global_pool_nhwc_f32(_ptr(_f32c(x)), cs, x_coff, B, H * W, c, _ptr(avg), _ptr(mx), _ptr(ws),
                                               _stream()), 'global_pool')
    return avg, mx


def global_pool_act(x, act, post_scale=None, post_shift=None, c=None, x_coff=0):
    """(B, c): post_scale * mean over pixels of act(x) + post_shift - the global average of an activation + per-channel affine map of x without
    writing that tensor (somi_global_pool_act_nhwc_f32)."""
    B, H, W, cs = x.shape
    c = cs - x_coff if c is None else c
    L = _lib.lib()
    ws = torch.empty(2 * B * L.somi_pool_nchunk(H * W) * c, device=x.device, dtype=torch.float32)
    avg = torch.empty(B, c, device=x.device, dtype=torch.float32)
    check(L.somi_global_pool_act_nhwc_f32(_ptr(_f32c(x)), cs, x_coff, B, H * W, c, ACT[act], _ptr(post_scale), _ptr(post_shift), _ptr(avg), _ptr(ws),
                                          _stream()), 'global_pool_act')
    return avg


def affine_silu_pool(x, c, x_coff, scale, shift, out, out_coff=0):
    """out = silu(x * scale + shift) and the global average / max pools of it per (image, channel) in one pass; -> (avg, max) (B, c), or None when
    the kernel does not cover the width (c / 4 must divide 256 or be a multiple of it) - the caller then runs the two separate passes."""
    B, H, W, _ = x.shape
    L = _lib.lib()
    rows = L.somi_affine_silu_pool_rows(B, H * W, c)
    if rows <= 0:
        return None
    ws = torch.empty(2 * B * rows * c, device=x.device, dtype=torch.float32)
    avg = torch.empty(B, c, device=x.device, dtype=torch.float32)
    mx = torch.empty(B, c, device=x.device, dtype=torch.float32)
    check(L.somi_affine_silu_pool_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, _ptr(scale), _ptr(shift), _ptr(_f32c(out)), out.shape[3], out_coff, B, H * W, c,
                                           _ptr(avg), _ptr(mx), _ptr(ws), _stream()), 'affine_silu_pool')
    return avg, mx


def attn_mlp(mode, avg, mx, W1, b1, W2, b2):
    B, Cc = avg.shape
    out = torch.empty_like(avg)
    check(_lib.lib().somi_attn_mlp_f32(mode, _ptr(avg), _ptr(mx), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(out), B, Cc,
                                       W1.shape[0], _stream()), 'attn_mlp')
    return out


def chan_stats(x, ca, c=None, x_coff=0):
    B, H, W, cs = x.shape
    c = cs - x_coff if c is None else c
    stats = torch.empty(B, H, W, 2, device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_chan_stats_nhwc_f32(_ptr(_f32c(x)), cs, x_coff, _ptr(ca), _ptr(stats), B, H * W, c, _stream()),
          'chan_stats')
    return stats


def spatial_attn(stats, w, bias, k):
    B, H, W, _ = stats.shape
    sa = torch.empty(B, H, W, device=stats.device, dtype=torch.float32)
    check(_lib.lib().somi_spatial_attn_f32(_ptr(stats), _ptr(w), _ptr(_f32c(bias)), _ptr(sa), B, H, W, k, _stream()), 'spatial_attn')
    return sa


def cbam_apply_(x, ca, stats, w, bias, k, c=None, x_coff=0):
    """In place: x[..., slice] *= ca[b,c] * sigmoid(conv_kxk(stats)+bias)  (CBAM, models/common.py:686-688)."""
    B, H, W, cs = x.shape
    c = cs - x_coff if c is None else c
    check(_lib.lib().somi_cbam_apply_nhwc_f32(_ptr(_f32c(x)), cs, x_coff, _ptr(ca), _ptr(stats), _ptr(w), _ptr(_f32c(bias)), _ptr(x), cs,
                                              x_coff, B, H, W, c, k, _stream()), 'cbam_apply')
    return x


def scale_channels(x, s=None, pix=None, out=None):
    B, H, W, Cc = x.shape
    out = torch.empty_like(x) if out is None else out
    check(_lib.lib().somi_scale_channels_nhwc_f32(_ptr(_f32c(x)), _ptr(s), _ptr(pix), _ptr(out), B, H * W, Cc, _stream()),
          'scale_channels')
    return out


def detect_decode(box, cls, anchors_px, stride, na, nc, raw=None, z=None, total=0, row_off=0):
    B, ny, nx, box_cs = box.shape
    a = (C.c_float * (na * 2))(*[float(v) for v in anchors_px])
    check(_lib.lib().somi_detect_decode_f32(_ptr(_f32c(box)), box_cs, _ptr(_f32c(cls)), cls.shape[3], a, float(stride),
                                            _ptr(raw), _ptr(z), B, ny, nx, na, nc, total, row_off, _stream()), 'detect_decode')


def detect_plain_decode(t, anchors_px, stride, na, nc, raw=None, z=None, total=0, row_off=0):
    """Plain `Detect` (models/yolo.py:66-98): t (B,ny,nx,>=na*no) -> raw (B,na,ny,nx,no) and rows of z."""
    B, ny, nx, t_cs = t.shape
    a = (C.c_float * (na * 2))(*[float(v) for v in anchors_px])
    check(_lib.lib().somi_detect_plain_decode_f32(_ptr(_f32c(t)), t_cs, a, float(stride), _ptr(raw), _ptr(z), B, ny, nx, na, nc, total,
                                                  row_off, _stream()), 'detect_plain_decode')


def detect_plain_raw_backward(draw, t_cs, na, nc):
    B, _, ny, nx, _ = draw.shape
    dt = torch.empty(B, ny, nx, t_cs, device=draw.device, dtype=torch.float32)
    check(_lib.lib().somi_detect_plain_raw_bwd_f32(_ptr(_f32c(draw)), _ptr(dt), t_cs, B, ny, nx, na, nc, _stream()), 'detect_plain_raw_bwd')
    return dt


def resample_slice(src, src_coff, dst, dst_coff, c, up=0, reduce=False, accumulate=False):
    """reduce=False: dst[b,h,w,dst_coff+c] = src[b,h>>up,w>>up,src_coff+c] (Concat copy with nn.Upsample folded in);
    reduce=True: the adjoint, dst (low resolution) (+)= block sums of src."""
    lo = dst if reduce else src
    B, Hs, Ws, _ = lo.shape
    hi = src if reduce else dst
    if hi.shape[1] != Hs << up or hi.shape[2] != Ws << up or hi.shape[0] != B:
        raise RuntimeError(f'resample_slice: shapes {tuple(src.shape)} / {tuple(dst.shape)} do not differ by 2^{up}')
    check(_lib.lib().somi_resample_slice_nhwc_f32(_ptr(_f32c(src)), src.shape[3], src_coff, _ptr(_f32c(dst)), dst.shape[3], dst_coff, B, Hs, Ws,
                                                  c, up, int(reduce), int(accumulate), _stream()), 'resample_slice')
    return dst


def space_to_depth(x, x_coff, c, out=None, inverse=False):
    """Focus (models/common.py:1996): (B,2Ho,2Wo,.) slice of c channels -> (B,Ho,Wo,4c); inverse=True maps a gradient in the deep
    layout back to the image layout (out must then be given or is allocated with pad4(c) channels)."""
    if not inverse:
        B, H, W, _ = x.shape
        out = torch.empty(B, H // 2, W // 2, (4 * c + 3) // 4 * 4, device=x.device, dtype=torch.float32) if out is None else out
        if 4 * c < out.shape[3]:
            out[..., 4 * c:].zero_()
        check(_lib.lib().somi_space_to_depth_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, _ptr(_f32c(out)), out.shape[3], 0, B, H // 2, W // 2, c, 0,
                                                      _stream()), 'space_to_depth')
        return out
    B, Ho, Wo, _ = x.shape
    out = torch.zeros(B, 2 * Ho, 2 * Wo, (c + 3) // 4 * 4, device=x.device, dtype=torch.float32) if out is None else out
    check(_lib.lib().somi_space_to_depth_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, _ptr(_f32c(out)), out.shape[3], 0, B, Ho, Wo, c, 1, _stream()),
          'space_to_depth (inverse)')
    return out


def tta_resample(x4, ratio, gs, flip_lr, channels=3, pad=0.447):
    """scale_img(x.flip(3) if flip_lr else x, ratio, gs=gs) (utils/torch_utils.py:270-282) on an ingested (B,H,W,4) image."""
    import math
    B, H, W, _ = x4.shape
    Hs, Ws = (H, W) if ratio == 1.0 else (int(H * ratio), int(W * ratio))
    Hp, Wp = (H, W) if ratio == 1.0 else (math.ceil(H * ratio / gs) * gs, math.ceil(W * ratio / gs) * gs)
    y = torch.empty(B, Hp, Wp, 4, device=x4.device, dtype=torch.float32)
    check(_lib.lib().somi_tta_resample_nhwc4_f32(_ptr(_f32c(x4)), _ptr(y), B, H, W, Hs, Ws, Hp, Wp, int(bool(flip_lr)), float(pad), channels,
                                                 _stream()), 'tta_resample')
    return y


def resize_bilinear_nhwc4(x4, Hn, Wn, channels=3):
    """F.interpolate(x, size=(Hn, Wn), mode='bilinear', align_corners=False) on an ingested (B,H,W,4) image (`--multi-scale`, train.py:257-262)."""
    B, H, W, _ = x4.shape
    y = torch.empty(B, Hn, Wn, 4, device=x4.device, dtype=torch.float32)
    check(_lib.lib().somi_tta_resample_nhwc4_f32(_ptr(_f32c(x4)), _ptr(y), B, H, W, Hn, Wn, Hn, Wn, 0, 0.0, channels, _stream()), 'resize_bilinear')
    return y


def tta_descale_(z, scale, flip_lr, img_w):
    """_descale_pred (models/yolo.py:1292-1308) in place on z (B, n, no)."""
    check(_lib.lib().somi_tta_descale_f32(_ptr(_f32c(z)), z.shape[0] * z.shape[1], z.shape[2], float(scale), int(bool(flip_lr)), float(img_w),
                                          _stream()), 'tta_descale')
    return z


def odconv_weights(gap, fc_w, fc_b, pk, wout, bout, cin, cin_pad, cout, kk, K):
    """Attention heads + per-sample weight synthesis of ODConv (models/common.py:4557-4590), outer BN folded in."""
    B = gap.shape[0]
    hid = fc_w.shape[0]
    ws = torch.empty(B * (cout + kk + cin + K), device=gap.device, dtype=torch.float32)
    check(_lib.lib().somi_odconv_weights_f32(_ptr(gap), _ptr(fc_w), _ptr(fc_b), _ptr(pk['Wf']), _ptr(pk['bf']), _ptr(pk['Ws']),
                                             _ptr(pk['bs']), _ptr(pk['Wc']), _ptr(pk['bc']), _ptr(pk['Ww']), _ptr(pk['bw']),
                                             _ptr(pk['Wk']), _ptr(pk['biask']), _ptr(pk['bn_s']), _ptr(pk['bn_t']),
                                             _ptr(wout), _ptr(bout), _ptr(ws), B, cin, cin_pad, cout, kk, K, hid, _stream()),
          'odconv_weights')
    return wout, bout


def layernorm_act(x, gamma, beta, eps, act='none'):
    """LayerNorm over the last dim of contiguous NHWC + activation (DCNv3 module, modules/dcnv3.py:283-291)."""
    out = torch.empty_like(x)
    C_ = x.shape[-1]
    check(_lib.lib().somi_layernorm_act_nhwc_f32(_ptr(_f32c(x)), _ptr(gamma), _ptr(beta), float(eps), ACT[act], _ptr(out),
                                                 x.numel() // C_, C_, _stream()), 'layernorm_act')
    return out


def group_softmax(x, K):
    """softmax over trailing groups of K (mask logits (N,H,W,G*K) -> softmax per (pixel, group), modules/dcnv3.py:334)."""
    out = torch.empty_like(x)
    check(_lib.lib().somi_group_softmax_f32(_ptr(_f32c(x)), _ptr(out), x.numel() // K, K, _stream()), 'group_softmax')
    return out


def layernorm_gelu_backward(u, gamma, beta, eps, dz, dgamma, dbeta):
    """z = gelu(LayerNorm(u)) -> du; dgamma / dbeta are accumulated."""
    C_ = u.shape[-1]
    n = u.numel() // C_
    du = torch.empty_like(u)
    L = _lib.lib()
    ws = torch.empty(L.somi_layernorm_act_bwd_workspace_floats(n, C_), device=u.device, dtype=torch.float32)
    check(L.somi_layernorm_gelu_bwd_nhwc_f32(_ptr(_f32c(u)), _ptr(gamma), _ptr(beta), float(eps), _ptr(_f32c(dz)), _ptr(du), _ptr(dgamma),
                                             _ptr(dbeta), _ptr(ws), n, C_, _stream()), 'layernorm_gelu_bwd')
    return du


def group_softmax_backward(y, dy, K):
    dx = torch.empty_like(y)
    check(_lib.lib().somi_group_softmax_bwd_f32(_ptr(_f32c(y)), _ptr(_f32c(dy)), _ptr(dx), y.numel() // K, K, _stream()), 'group_softmax_bwd')
    return dx


def cfs_blend_backward(x, xproj, logit, dout, G, Gc):
    """-> (dx, dxproj, dlogit) of out = x*(1-s) + xproj*s, s = sigmoid(logit) broadcast over a group's channels."""
    dx, dxp = torch.empty_like(x), torch.empty_like(x)
    dlogit = torch.zeros_like(logit)                              # pad columns (if any) stay zero
    check(_lib.lib().somi_dcnv3_cfs_blend_bwd_f32(_ptr(_f32c(x)), _ptr(_f32c(xproj)), _ptr(_f32c(logit)), logit.shape[-1], _ptr(_f32c(dout)),
                                                  _ptr(dx), _ptr(dxp), _ptr(dlogit), dlogit.shape[-1], x.numel() // (G * Gc), G, Gc,
                                                  _stream()), 'cfs_blend_bwd')
    return dx, dxp, dlogit


def cfs_blend(x, xproj, logit, G, Gc):
    out = torch.empty_like(x)
    check(_lib.lib().somi_dcnv3_cfs_blend_f32(_ptr(_f32c(x)), _ptr(_f32c(xproj)), _ptr(_f32c(logit)), logit.shape[-1], _ptr(out),
                                              x.numel() // (G * Gc), G, Gc, _stream()), 'cfs_blend')
    return out


def conv2d_dgrad_nhwc(dy, w_dgrad, *, B, H, W, cin, kh, kw, stride=1, pad=0, cout=None, dy_coff=0, out=None, dx_coff=0,
                      accumulate=None, acc_coff=0, per_sample_w=False, accumulate2=None, acc2_coff=0):
    """dx of the conv x (B,H,W,cin) -> y (B,Ho,Wo,cout); dy is (B,Ho,Wo,dy_cs); w_dgrad packed [cin][kh*kw*cout]."""
    _, Ho, Wo, dy_cs = dy.shape
    cout = dy_cs - dy_coff if cout is None else cout
    if out is None:
        out = torch.empty(B, H, W, cin, device=dy.device, dtype=torch.float32)
    d = ConvDesc()
    d.prec = CONV_PREC
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, cin, Ho, Wo, cout
    d.kh, d.kw, d.stride, d.pad, d.dil, d.per_sample_w = kh, kw, stride, pad, 1, int(per_sample_w)
    _conv_workspace(d, dy.device)
    if accumulate2 is not None:                                   # second added tensor (e.g. the shortcut branch's gradient)
        d.residual2, d.res2_cs, d.res2_coff = _ptr(_f32c(accumulate2)), accumulate2.shape[3], acc2_coff
    if w_dgrad.numel() != (B if per_sample_w else 1) * cin * kh * kw * cout:
        raise RuntimeError('dgrad weight has the wrong number of elements')
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(_lib.lib().somi_conv2d_dgrad_nhwc_f32(C.byref(d), _ptr(_f32c(dy)), dy_cs, dy_coff, _ptr(_f32c(w_dgrad)), _ptr(_f32c(out)),
                                                out.shape[3], dx_coff, _ptr(accumulate),
                                                accumulate.shape[3] if accumulate is not None else 0, acc_coff, _stream()),
          'conv2d_dgrad_nhwc')
    if prof:
        e1.record()
        g = ConvDesc()                                           # the geometry the kernel runs in: rows = forward-input pixels
        g.B, g.H, g.W, g.Cin, g.Ho, g.Wo, g.Cout = B, Ho, Wo, cout, H, W, cin
        g.kh, g.kw, g.stride, g.per_sample_w = kh, kw, stride, int(per_sample_w)
        if stride == 1:
            g.workspace, g.workspace_bytes = d.workspace, d.workspace_bytes
        name = _lib.lib().somi_conv2d_kernel_name(C.byref(g)).decode()
        PROFILE.append((name, 2.0 * B * Ho * Wo * cout * cin * kh * kw, e0, e1, (B, H, W, cin, cout, kh, stride, 2)))
    return out


def conv2d_wgrad_nhwc(x, dy, *, kh, kw, stride=1, pad=0, cin=None, x_coff=0, cout=None, dy_coff=0, out=None, accumulate=None,
                      per_sample_w=False):
    """dW [Cout][kh*kw*Cin] (forward packing) of the conv x (B,H,W,cin) -> y (B,Ho,Wo,cout) from x and dy."""
    B, H, W, x_cs = x.shape
    _, Ho, Wo, dy_cs = dy.shape
    cin = x_cs - x_coff if cin is None else cin
    cout = dy_cs - dy_coff if cout is None else cout
    d = ConvDesc()
    d.prec = CONV_PREC
    d.B, d.H, d.W, d.Cin, d.Ho, d.Wo, d.Cout = B, H, W, cin, Ho, Wo, cout
    d.kh, d.kw, d.stride, d.pad, d.dil, d.per_sample_w = kh, kw, stride, pad, 1, int(per_sample_w)
    shape = ((B,) if per_sample_w else ()) + (cout, kh * kw * cin)
    if out is None:
        out = torch.empty(*shape, device=x.device, dtype=torch.float32)
    L = _lib.lib()
    nbytes = L.somi_conv2d_wgrad_workspace_bytes(C.byref(d))
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=x.device)
    prof = PROFILE is not None
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(L.somi_conv2d_wgrad_nhwc_f32(C.byref(d), _ptr(_f32c(x)), x_cs, x_coff, _ptr(_f32c(dy)), dy_cs, dy_coff, _ptr(_f32c(out)),
                                       _ptr(accumulate), _ptr(ws), nbytes, _stream()), 'conv2d_wgrad_nhwc')
    if prof:
        e1.record()
        bm = 64 if -(-cout // 64) * 64 < -(-cout // 128) * 128 else 128            # conv_wgrad.hip plan()
        K = kh * kw * cin
        bn = 128 if K > 64 else (64 if (K > 32 or bm == 64) else 32)
        eight = int(os.environ.get('SOMI_WGRAD_8WAVE', '2'))
        nw = 8 if ((bm, bn) == (128, 128) and eight >= 1) or ((bm, bn) == (64, 128) and eight >= 2) else 4
        PROFILE.append((f'conv_wgrad_f32_kernel<{bm},{bn},{nw}>', 2.0 * B * Ho * Wo * cout * cin * kh * kw, e0, e1,
                        (B, H, W, cin, cout, kh, stride, 3)))
    return out


# ---------------------------------------------------------------------------------------------- training-mode helpers
def _npix(t):
    return t.shape[0] * t.shape[1] * t.shape[2]


# --sync-bn (reference train.py:165-167, torch.nn.SyncBatchNorm): set to the torch.distributed module (train.TrainStep(sync_bn=True) does
# it around its step) and every BatchNorm statistic below spans the batches of all ranks: one small all-gather per layer and direction.
SYNC_BN = None
SYNC_BN_GROUP = None     # the process group of these all-gathers.  TrainStep gives sync-BN a group of its OWN: on the default group every
#                          blocking all-gather would queue behind the asynchronous gradient buckets (collectives of one RCCL communicator run
#                          in issue order) and stall the backward pass until they finish - the overlap of the exchange would be lost.


def _gather_records(rec):
    """all-gather of one [2C + 1] double record per rank -> (flat [world * (2C + 1)] tensor, world)."""
    world = SYNC_BN.get_world_size(SYNC_BN_GROUP)
    out = torch.empty(world * rec.numel(), dtype=torch.float64, device=rec.device)
    SYNC_BN.all_gather_into_tensor(out, rec, group=SYNC_BN_GROUP)
    return out, world


def _bn_stats_sync(x, x_coff, n, c, part, rows, gamma, beta, eps, momentum, running_mean, running_var):
    L, dev = _lib.lib(), (x if x is not None else part).device
    mean, rstd, scale, shift = (torch.empty(c, device=dev, dtype=torch.float32) for _ in range(4))
    rec = torch.empty(2 * c + 1, device=dev, dtype=torch.float64)
    ws = torch.empty(2 * max(1024, L.somi_red_nchunk(n)) * c, device=dev, dtype=torch.float32)
    if part is not None:
        check(L.somi_bn_local_sums_f64(None, 0, 0, n, c, _ptr(running_mean), part[0].data_ptr(), part[1].data_ptr(), rows, _ptr(rec), _ptr(ws),
                                       _stream()), 'bn_local_sums')
    else:
        check(L.somi_bn_local_sums_f64(_ptr(_f32c(x)), x.shape[3], x_coff, n, c, _ptr(running_mean), None, None, 0, _ptr(rec), _ptr(ws),
                                       _stream()), 'bn_local_sums')
    allrec, world = _gather_records(rec)
    check(L.somi_bn_stats_from_sums_f64(_ptr(allrec), world, c, float(eps), float(momentum), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd),
                                        _ptr(scale), _ptr(shift), _ptr(running_mean), _ptr(running_var), _stream()), 'bn_stats_from_sums')
    return mean, rstd, scale, shift


def bn_stats(x, c, x_coff, gamma, beta, eps, momentum, running_mean=None, running_var=None, act='none'):
    """Batch statistics of a channel slice - of act(x) when `act` is given, without storing act(x) - -> (mean, rstd, scale, shift);
    optionally updates the running statistics."""
    dev = x.device
    n = _npix(x)
    if SYNC_BN is not None:
        if act != 'none':
            raise RuntimeError('bn_stats(act=...) has no sync-BN form: materialise act(x) first (blocks.SEAM does)')
        return _bn_stats_sync(x, x_coff, n, c, None, 0, gamma, beta, eps, momentum, running_mean, running_var)
    mean, rstd, scale, shift = (torch.empty(c, device=dev, dtype=torch.float32) for _ in range(4))
    ws = torch.empty(2 * _lib.lib().somi_red_nchunk(n) * c, device=dev, dtype=torch.float32)
    check(_lib.lib().somi_bn_stats_act_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, ACT[act], n, c, float(eps), float(momentum), _ptr(gamma),
                                                _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(shift), _ptr(running_mean),
                                                _ptr(running_var), _ptr(ws), _stream()), 'bn_stats')
    return mean, rstd, scale, shift


def bn_stats_from_partials(part, rows, npix, c, gamma, beta, eps, momentum, running_mean=None, running_var=None):
    """Batch statistics from the partial sums a convolution's epilogue left (conv2d_nhwc(..., bn_stats={'pivot': running_mean}));
    `running_mean` must be that same pivot.  -> (mean, rstd, scale, shift); updates the running statistics."""
    dev = part.device
    if SYNC_BN is not None:
        return _bn_stats_sync(None, 0, npix, c, part, rows, gamma, beta, eps, momentum, running_mean, running_var)
    mean, rstd, scale, shift = (torch.empty(c, device=dev, dtype=torch.float32) for _ in range(4))
    ws = torch.empty(2 * 1024 * c, device=dev, dtype=torch.float32)
    check(_lib.lib().somi_bn_stats_partials_f32(part[0].data_ptr(), part[1].data_ptr(), rows, npix, c, float(eps), float(momentum), _ptr(gamma),
                                                _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(scale), _ptr(shift), _ptr(running_mean),
                                                _ptr(running_var), _ptr(ws), _stream()), 'bn_stats_partials')
    return mean, rstd, scale, shift


def chan_affine_act(x, c, x_coff, scale, shift, act, order, out, out_coff=0, residual=None, res_coff=0):
    check(_lib.lib().somi_chan_affine_act_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, _ptr(scale), _ptr(shift), ACT[act], order,
                                                   _ptr(_f32c(out)), out.shape[3], out_coff, _npix(x), c,
                                                   _ptr(residual), residual.shape[3] if residual is not None else 0, res_coff,
                                                   _stream()), 'chan_affine_act')
    return out


def pack_dgrad_weights(w_packed, cout, taps, cin):
    """[cout][taps][cin] (forward packing) -> [cin][taps][cout] (operand of conv2d_dgrad_nhwc)."""
    out = torch.empty(cin, taps * cout, device=w_packed.device, dtype=torch.float32)
    check(_lib.lib().somi_pack_dgrad_weights_f32(_ptr(_f32c(w_packed)), _ptr(out), cout, taps, cin, _stream()), 'pack_dgrad_weights')
    return out


def bn_act_backward(dz, dz_coff, x, x_coff, c, mean, rstd, scale, shift, act, order, batch_stats, dx, dx_coff=0, dgamma=None,
                    dbeta=None, pooled=None):
    """pooled = (davg, dmax, amaxp), each (B, c): the gradient of a channel attention's global average / max pools over this tensor, i.e.
    dz_eff = dz + davg / HW + [pixel == amaxp] * dmax.  The image-aligned kernels take it on the fly (dz is only read); under sync-BN - whose
    reduce / apply entries have no such form - it is added to dz in place by a pass of its own first.  dmax / amaxp may be None (average pool
    only); dz may be None when nothing else reads the tensor (the pooled part is the whole gradient)."""
    n = _npix(x)
    if pooled is not None:
        davg, dmax, amaxp = pooled
        if BN_POOLED and SYNC_BN is None and batch_stats and davg.shape[1] == c and x.dim() == 4:
            B, HW = x.shape[0], x.shape[1] * x.shape[2]
            L = _lib.lib()
            ws = torch.empty(2 * L.somi_bn_pooled_rows(B, HW) * c + 3 * ((c + 3) // 4 * 4), device=x.device, dtype=torch.float32)
            check(L.somi_bn_act_backward_pooled_nhwc_f32(_ptr(None if dz is None else _f32c(dz)), 0 if dz is None else dz.shape[3], dz_coff,
                                                         _ptr(_f32c(x)), x.shape[3], x_coff, _ptr(mean),
                                                         _ptr(rstd), _ptr(scale), _ptr(shift), ACT[act], order, _ptr(davg), _ptr(dmax),
                                                         _ptr(amaxp), _ptr(_f32c(dx)), dx.shape[3], dx_coff, _ptr(dgamma), _ptr(dbeta), B, HW, c,
                                                         _ptr(ws), _stream()), 'bn_act_backward_pooled')
            return dx
        if dz is None:                                            # dz = None: the pooled part is the whole incoming gradient
            dz, dz_coff = torch.zeros_like(x), x_coff
        pool_backward_add_(dz, dz_coff, davg.shape[1], davg, dmax, amaxp)
    if SYNC_BN is not None and batch_stats:
        L = _lib.lib()
        dzc, xc, dxc = _f32c(dz), _f32c(x), _f32c(dx)
        rec = torch.empty(2 * c + 1, device=x.device, dtype=torch.float64)
        ws = torch.empty(2 * L.somi_red_nchunk(n) * c + 3 * ((c + 3) // 4 * 4), device=x.device, dtype=torch.float32)
        check(L.somi_bn_act_backward_sums_f64(_ptr(dzc), dzc.shape[3], dz_coff, _ptr(xc), xc.shape[3], x_coff, _ptr(mean), _ptr(scale),
                                              _ptr(shift), ACT[act], order, n, c, _ptr(rec), _ptr(ws), _stream()), 'bn_act_backward_sums')
        allrec, world = _gather_records(rec)
        check(L.somi_bn_act_backward_apply_sync_f32(_ptr(dzc), dzc.shape[3], dz_coff, _ptr(xc), xc.shape[3], x_coff, _ptr(mean), _ptr(rstd),
                                                    _ptr(scale), _ptr(shift), ACT[act], order, _ptr(rec), _ptr(allrec), world, _ptr(dxc),
                                                    dxc.shape[3], dx_coff, _ptr(dgamma), _ptr(dbeta), n, c, _ptr(ws), _stream()),
              'bn_act_backward_apply_sync')
        return dx
    ws = torch.empty(2 * _lib.lib().somi_red_nchunk(n) * c + 3 * ((c + 3) // 4 * 4), device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_bn_act_backward_nhwc_f32(_ptr(_f32c(dz)), dz.shape[3], dz_coff, _ptr(_f32c(x)), x.shape[3], x_coff, _ptr(mean),
                                                   _ptr(rstd), _ptr(scale), _ptr(shift), ACT[act], order, int(batch_stats), _ptr(_f32c(dx)),
                                                   dx.shape[3], dx_coff, _ptr(dgamma), _ptr(dbeta), n, c, _ptr(ws), _stream()),
          'bn_act_backward')
    return dx


def chan_sum_(x, c, x_coff, out_accumulate):
    n = _npix(x)
    ws = torch.empty(2 * _lib.lib().somi_red_nchunk(n) * c, device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_chan_sum_nhwc_f32(_ptr(_f32c(x)), x.shape[3], x_coff, n, c, _ptr(out_accumulate), _ptr(ws), _stream()), 'chan_sum')
    return out_accumulate


def add_(a, a_coff, b, b_coff, c, out=None, out_coff=None):
    """out[slice] = a[slice] + b[slice]; defaults to in place on a."""
    out = a if out is None else out
    out_coff = a_coff if out_coff is None else out_coff
    check(_lib.lib().somi_add_nhwc_f32(_ptr(_f32c(a)), a.shape[3], a_coff, _ptr(_f32c(b)), b.shape[3], b_coff, _ptr(_f32c(out)),
                                       out.shape[3], out_coff, _npix(a), c, _stream()), 'add')
    return out


def cbam_backward(dt2, t, t_coff, c, ca, sa, stats, w7, k, dw7, db7, t_max=None, dw_chw=False, bn=None):
    """Steps A-C of train_blocks.hip: turns d(t*ca*sa) (dt2, whole tensor, modified in place) into the part of dt that flows
    through the two multiplications and the spatial branch; returns dca (B,C) and, when t_max (B,C) - the spatial maximum of t the forward pooled -
    is given, amaxp (B,C) int32: the first pixel that holds each channel's maximum (step D, found by value inside step A's pass), else None.
    dw7 / db7 are accumulated; dw7 is [k][k][2] like w7, or (dw_chw) laid out (2,k,k) like nn.Conv2d's weight.
    bn = (y, scale, shift, mean) of the Conv -> BatchNorm -> SiLU that produced t (needs t_max): step C is NOT run as a pass of its own - dt2 is left
    alone, dca comes from the reduction half of the fused BatchNorm backward, and a third value is returned: the state cbam_bn_backward_apply needs."""
    B, H, W, _ = t.shape
    dev = t.device
    L = _lib.lib()
    dlogit = torch.empty(B, H, W, device=dev, dtype=torch.float32)
    amaxc = torch.empty(B, H, W, device=dev, dtype=torch.int32)
    amaxp = None
    if t_max is not None:
        amaxp = torch.empty(B, c, device=dev, dtype=torch.int32)
        check(L.somi_cbam_bwd_pixel_argmax_f32(_ptr(_f32c(dt2)), dt2.shape[3], 0, _ptr(_f32c(t)), t.shape[3], t_coff, _ptr(ca), _ptr(sa), _ptr(t_max),
                                               _ptr(dlogit), _ptr(amaxc), _ptr(amaxp), B, H * W, c, _stream()), 'cbam_bwd_pixel')
    else:
        check(L.somi_cbam_bwd_pixel_f32(_ptr(_f32c(dt2)), dt2.shape[3], 0, _ptr(_f32c(t)), t.shape[3], t_coff, _ptr(ca), _ptr(sa), _ptr(dlogit),
                                        _ptr(amaxc), B, H * W, c, _stream()), 'cbam_bwd_pixel')
    dstats = torch.empty(B, H, W, 2, device=dev, dtype=torch.float32)
    ws = torch.empty(((B * H * W + 511) // 512) * (2 * k * k + 1), device=dev, dtype=torch.float32)
    check(L.somi_spatial_attn_bwd_f32(_ptr(dlogit), _ptr(stats), _ptr(w7), _ptr(dstats), _ptr(dw7), _ptr(db7), _ptr(ws), B, H, W, k,
                                      int(dw_chw), _stream()), 'spatial_attn_bwd')
    dca = torch.empty(B, c, device=dev, dtype=torch.float32)
    if bn is not None:
        y, scale, shift, mean = bn
        ws3 = torch.empty(L.somi_cbam_bn_bwd_workspace_floats(B, H * W, c), device=dev, dtype=torch.float32)
        head = (_ptr(_f32c(dt2)), dt2.shape[3], 0, _ptr(_f32c(y)), y.shape[3], 0, _ptr(scale), _ptr(shift), _ptr(mean))
        check(L.somi_cbam_bn_bwd_reduce_f32(*head, _ptr(ca), _ptr(sa), _ptr(dstats), _ptr(amaxc), _ptr(amaxp), _ptr(dca), _ptr(ws3), B, H * W, c,
                                            _stream()), 'cbam_bn_bwd_reduce')
        return dca, amaxp, (dt2, y, scale, shift, mean, ca, sa, dstats, amaxc, amaxp, ws3, c)
    ws2 = torch.empty(B * L.somi_img_nchunk(H * W) * c, device=dev, dtype=torch.float32)
    check(L.somi_cbam_bwd_chan_f32(_ptr(dt2), dt2.shape[3], 0, _ptr(t), t.shape[3], t_coff, _ptr(ca), _ptr(sa), _ptr(dstats), _ptr(amaxc),
                                   _ptr(dca), _ptr(ws2), B, H * W, c, _stream()), 'cbam_bwd_chan')
    return dca, amaxp


def cbam_bn_backward_apply(state, rstd, davg, dmax, dx, dgamma, dbeta):
    """Second half of the fused CBAM + BatchNorm backward (somi_cbam_bn_bwd_apply_f32): state from cbam_backward(bn=...), davg / dmax (B,C) from the
    channel attention's MLP backward on the dca it returned; writes the gradient w.r.t. the convolution output into dx, accumulates dgamma / dbeta."""
    d, y, scale, shift, mean, ca, sa, dstats, amaxc, amaxp, ws, c = state
    B, H, W, _ = y.shape
    check(_lib.lib().somi_cbam_bn_bwd_apply_f32(_ptr(d), d.shape[3], 0, _ptr(y), y.shape[3], 0, _ptr(scale), _ptr(shift), _ptr(mean), _ptr(rstd), _ptr(ca),
                                                _ptr(sa), _ptr(dstats), _ptr(amaxc), _ptr(amaxp), _ptr(davg), _ptr(dmax), _ptr(_f32c(dx)), dx.shape[3], 0,
                                                _ptr(dgamma), _ptr(dbeta), _ptr(ws), B, H * W, c, _stream()), 'cbam_bn_bwd_apply')
    return dx


def pool_argmax(x, c, x_coff=0):
    B, H, W, cs = x.shape
    L = _lib.lib()
    out = torch.empty(B, c, device=x.device, dtype=torch.int32)
    ws = torch.empty(2 * B * L.somi_img_nchunk(H * W) * c, device=x.device, dtype=torch.float32)
    check(L.somi_pool_argmax_nhwc_f32(_ptr(_f32c(x)), cs, x_coff, B, H * W, c, _ptr(out), _ptr(ws), _stream()), 'pool_argmax')
    return out


def attn_mlp_backward(mode, dout, out, avg, mx, W1, b1, W2, dW1, db1, dW2, db2):
    B, Cc = avg.shape
    davg = torch.empty_like(avg)
    dmax = torch.empty_like(avg) if mode == 0 else None
    L = _lib.lib()
    ws = torch.empty(L.somi_attn_mlp_bwd_workspace_floats(B, Cc, W1.shape[0]), device=avg.device, dtype=torch.float32)
    check(L.somi_attn_mlp_bwd_f32(mode, _ptr(dout), _ptr(out), _ptr(avg), _ptr(mx), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(dW1),
                                  _ptr(db1), _ptr(dW2), _ptr(db2), _ptr(davg), _ptr(dmax), _ptr(ws), B, Cc, W1.shape[0], _stream()),
          'attn_mlp_bwd')
    return davg, dmax


def pool_backward_add_(dt, d_coff, c, davg, dmax=None, amaxp=None):
    B, H, W, cs = dt.shape
    check(_lib.lib().somi_pool_bwd_add_nhwc_f32(_ptr(_f32c(dt)), cs, d_coff, _ptr(davg), _ptr(dmax), _ptr(amaxp), B, H * W, c, _stream()),
          'pool_bwd_add')
    return dt


def detect_raw_backward(draw, box_cs, cls_cs, na, nc):
    B, _, ny, nx, _ = draw.shape
    dbox = torch.empty(B, ny, nx, box_cs, device=draw.device, dtype=torch.float32)
    dcls = torch.empty(B, ny, nx, cls_cs, device=draw.device, dtype=torch.float32)
    check(_lib.lib().somi_detect_raw_bwd_f32(_ptr(_f32c(draw)), _ptr(dbox), box_cs, _ptr(dcls), cls_cs, B, ny, nx, na, nc, _stream()),
          'detect_raw_bwd')
    return dbox, dcls


def sppf_pool_backward_(buf, dbuf, c, x_coff=0, codes=None):
    """codes: what sppf_pool_(codes=True) returned (then `buf` is not read); else the windows are searched again."""
    B, H, W, cs = dbuf.shape
    ws = torch.empty(3 * B * H * W * c, device=dbuf.device, dtype=torch.uint8) if codes is None else codes
    check(_lib.lib().somi_sppf_pool_bwd_nhwc_f32(_ptr(_f32c(buf)) if codes is None else None, _ptr(_f32c(dbuf)), _ptr(ws), B, H, W, c, cs, x_coff,
                                                 _stream()), 'sppf_pool_bwd')
    return dbuf


def bifpn_backward(srcs, ups, w_dev, dout, dw, eps=1e-4):
    n = len(srcs)
    B, H, W, Cc = dout.shape
    dsrcs = [torch.empty_like(s) for s in srcs]
    sp = (C.c_void_p * n)(*[_ptr(_f32c(s)) for s in srcs])
    dp = (C.c_void_p * n)(*[_ptr(d) for d in dsrcs])
    ws = torch.empty(3 * 2048, device=dout.device, dtype=torch.float32)
    check(_lib.lib().somi_bifpn_bwd_nhwc_f32(sp, dp, (C.c_int * n)(*ups), _ptr(_f32c(w_dev)), float(eps), n,
                                             _ptr(_f32c(dout)), _ptr(dw), _ptr(ws), B, H, W, Cc, _stream()), 'bifpn_bwd')
    return dsrcs


def dwconv3x3_backward(dy, x, w, dw, dbias, dx_accumulate=None):
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    ws = torch.empty(max(_lib.lib().somi_dwconv3x3_bwd_workspace_floats(B, W, Cc), 1), device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_dwconv3x3_bwd_nhwc_f32(_ptr(_f32c(dy)), _ptr(_f32c(x)), _ptr(w), _ptr(dx), _ptr(dx_accumulate), _ptr(dw), _ptr(dbias),
                                                 _ptr(ws), B, H, W, Cc, _stream()), 'dwconv3x3_bwd')
    return dx


def scale_channels_backward(dout, x, s):
    B, H, W, Cc = x.shape
    dx = torch.empty_like(x)
    ds = torch.empty(B, Cc, device=x.device, dtype=torch.float32)
    ws = torch.empty(B * _lib.lib().somi_img_nchunk(H * W) * Cc, device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_scale_channels_bwd_nhwc_f32(_ptr(_f32c(dout)), _ptr(_f32c(x)), _ptr(s), _ptr(dx), _ptr(ds), _ptr(ws), B, H * W, Cc,
                                                      _stream()), 'scale_channels_bwd')
    return dx, ds


def linear(x, W, bias, act, out=None, out_off=0):
    """Small dense layer on (B, nin) rows (ODConv attention heads): out[:, out_off:out_off+nout] = act(x W^T + bias)."""
    B, nin = x.shape
    nout = W.shape[0]
    out = torch.empty(B, nout, device=x.device, dtype=torch.float32) if out is None else out
    check(_lib.lib().somi_linear_f32(_ptr(_f32c(x)), x.shape[1], _ptr(_f32c(W)), _ptr(bias), ACT[act], _ptr(out), out.shape[1], out_off, B, nin,
                                     nout, _stream()), 'linear')
    return out


def linear_backward(x, W, dy, y, off, act, dW, db, dx=None, accumulate=False):
    B, nin = x.shape
    nout = W.shape[0]
    ws = torch.empty(B * nout, device=x.device, dtype=torch.float32)
    check(_lib.lib().somi_linear_bwd_f32(_ptr(_f32c(x)), nin, _ptr(_f32c(W)), _ptr(dy), _ptr(y), y.shape[1], off, ACT[act], _ptr(dW), _ptr(db),
                                         _ptr(dx), dx.shape[1] if dx is not None else 0, int(accumulate), _ptr(ws), B, nin, nout, _stream()),
          'linear_bwd')
    return dx


def odconv_synth(attn, Wk, biask, cin, cin_pad, cout, kk, K):
    B = attn.shape[0]
    wout = torch.empty(B, cout, kk * cin_pad, device=attn.device, dtype=torch.float32)
    bout = torch.empty(B, cout, device=attn.device, dtype=torch.float32)
    check(_lib.lib().somi_odconv_synth_f32(_ptr(attn), _ptr(Wk), _ptr(biask), _ptr(wout), _ptr(bout), B, cin, cin_pad, cout, kk, K, _stream()),
          'odconv_synth')
    return wout, bout


def odconv_synth_backward(dWb, attn, Wk, biask, dbias_b, dWk, dbiask, cin, cin_pad, cout, kk, K):
    B = attn.shape[0]
    dattn = torch.empty_like(attn)
    L = _lib.lib()
    ws = torch.empty(L.somi_odconv_synth_bwd_workspace_floats(B, cin, cout, kk, K), device=attn.device, dtype=torch.float32)
    check(L.somi_odconv_synth_bwd_f32(_ptr(_f32c(dWb)), _ptr(attn), _ptr(Wk), _ptr(biask), _ptr(dbias_b), _ptr(dWk), _ptr(dbiask),
                                      _ptr(dattn), _ptr(ws), B, cin, cin_pad, cout, kk, K, _stream()), 'odconv_synth_bwd')
    return dattn
