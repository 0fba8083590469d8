"""PyTorch-ROCm custom-op registration of the hot path (`torch.ops.somi.*`, north_star: "surfaced to Python through PyTorch-ROCm
custom ops").  Importing this module registers, for the device ("cuda" = HIP on ROCm) only:

  somi::dcnv3_forward / somi::dcnv3_backward   the reference extension's two entry points (models/ops_dcnv3/src/dcnv3.h:20-59), with the
                                               autograd formula of functions/dcnv3_func.py:19-61 attached to the forward op
  somi::conv2d_nhwc                            the fused NHWC implicit-GEMM convolution (Conv+BN(folded)+act epilogue, models/common.py:64-70)
  somi::nms                                    utils/general.non_max_suppression as one batched op -> (det (B,max_det,6), count (B))
  somi::yolo_loss                              ComputeLoss.__call__ (utils/loss.py:142-208) -> (out[4], grads per level)

Each op is a thin shim over the C ABI (libsomi_hip.so).  There is deliberately NO CPU kernel: a CPU tensor raises PyTorch's own
"could not run 'somi::...' with arguments from the 'CPU' backend" (the reference's extension answers "Not implemented on the CPU",
src/dcnv3.h:37).  Fake (meta) kernels give shapes to tracing tools.
"""
from typing import List, Tuple

import torch

from . import dcnv3 as _dcn
from . import ops as _ops

_DEV = 'cuda'


@torch.library.custom_op('somi::dcnv3_forward', mutates_args=(), device_types=_DEV)
def dcnv3_forward(input: torch.Tensor, offset: torch.Tensor, mask: torch.Tensor, kernel_h: int, kernel_w: int, stride_h: int, stride_w: int,
                  pad_h: int, pad_w: int, dilation_h: int, dilation_w: int, group: int, group_channels: int, offset_scale: float,
                  im2col_step: int) -> torch.Tensor:
    return _dcn.dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                              group_channels, offset_scale, im2col_step)


@dcnv3_forward.register_fake
def _(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale,
      im2col_step):
    N, H, W, _ = input.shape
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    return input.new_empty(N, Ho, Wo, group * group_channels)


@torch.library.custom_op('somi::dcnv3_backward', mutates_args=(), device_types=_DEV)
def dcnv3_backward(input: torch.Tensor, offset: torch.Tensor, mask: torch.Tensor, kernel_h: int, kernel_w: int, stride_h: int, stride_w: int,
                   pad_h: int, pad_w: int, dilation_h: int, dilation_w: int, group: int, group_channels: int, offset_scale: float,
                   grad_output: torch.Tensor, im2col_step: int) -> List[torch.Tensor]:
    return _dcn.dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                               group_channels, offset_scale, grad_output, im2col_step)


@dcnv3_backward.register_fake
def _(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels, offset_scale,
      grad_output, im2col_step):
    return [torch.empty_like(input), torch.empty_like(offset), torch.empty_like(mask)]


def _dcn_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[:3])
    ctx.cfg = inputs[3:14]
    ctx.im2col_step = inputs[14]


def _dcn_backward(ctx, grad_output):                             # functions/dcnv3_func.py:49-61: three gradients + 12 None
    input, offset, mask = ctx.saved_tensors
    gi, go, gm = torch.ops.somi.dcnv3_backward(input, offset, mask, *ctx.cfg, grad_output.contiguous(), ctx.im2col_step)
    return (gi, go, gm) + (None,) * 12


dcnv3_forward.register_autograd(_dcn_backward, setup_context=_dcn_setup)


@torch.library.custom_op('somi::conv2d_nhwc', mutates_args=(), device_types=_DEV)
def conv2d_nhwc(x: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, kh: int, kw: int, stride: int, pad: int, act: str) -> torch.Tensor:
    """x (B,H,W,Cin) NHWC fp32, Cin % 4 == 0; w_packed [Cout][kh*kw*Cin] (BN folded by the caller); act in none/silu/gelu/relu/sigmoid."""
    return _ops.conv2d_nhwc(x, w_packed, bias, kh=kh, kw=kw, stride=stride, pad=pad, act=act)


@conv2d_nhwc.register_fake
def _(x, w_packed, bias, kh, kw, stride, pad, act):
    B, H, W, _ = x.shape
    return x.new_empty(B, (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1, w_packed.shape[0])


@torch.library.custom_op('somi::nms', mutates_args=(), device_types=_DEV)
def nms(prediction: torch.Tensor, conf_thres: float, iou_thres: float, multi_label: bool, agnostic: bool,
        max_det: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """-> det (B,max_det,6) [x1,y1,x2,y2,conf,cls], count (B) int32: image b keeps det[b, :count[b]] (utils/general.py:629-711)."""
    from . import _lib
    from ._lib import check
    if prediction.dtype != torch.float32 or not prediction.is_contiguous():
        raise RuntimeError('prediction tensor has to be contiguous float32')
    B, n, no = prediction.shape
    nc = no - 5
    ml = bool(multi_label) and nc > 1
    L = _lib.lib()
    nbytes = L.somi_nms_workspace_bytes(B, n, nc, int(ml))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=prediction.device)
    det = torch.zeros(B, max_det, 6, dtype=torch.float32, device=prediction.device)
    count = torch.empty(B, dtype=torch.int32, device=prediction.device)
    check(L.somi_nms_f32(_ops._ptr(prediction), B, n, nc, float(conf_thres), float(iou_thres), int(ml), int(bool(agnostic)), None, int(max_det),
                         _ops._ptr(det), _ops._ptr(count), _ops._ptr(ws), nbytes, _ops._stream()), 'somi::nms')
    return det, count


@nms.register_fake
def _(prediction, conf_thres, iou_thres, multi_label, agnostic, max_det):
    B = prediction.shape[0]
    return prediction.new_empty(B, max_det, 6), prediction.new_empty(B, dtype=torch.int32)


@torch.library.custom_op('somi::yolo_loss', mutates_args=(), device_types=_DEV)
def yolo_loss(p: List[torch.Tensor], targets: torch.Tensor, anchors: torch.Tensor, balance: List[float], gains: List[float],
              need_grad: bool) -> List[torch.Tensor]:
    """ComputeLoss.__call__ (utils/loss.py:142-208).  gains = [box, obj, cls, cls_pw, obj_pw, anchor_t, cp, cn, gr] (hyp after the scaling of
    train.py:211-214).  -> [out4 = (total*bs, lbox, lobj, lcls)] + (need_grad: d out4[0] / d p[l] for every level)."""
    from .loss import ComputeLoss

    class _M:                                                    # the attributes ComputeLoss reads from `model`
        pass
    det = _M()
    det.nl, det.na, det.nc, det.anchors = len(p), p[0].shape[1], p[0].shape[4] - 5, anchors
    m = _M()
    m.model = [det]
    m.hyp = dict(box=gains[0], obj=gains[1], cls=gains[2], cls_pw=gains[3], obj_pw=gains[4], anchor_t=gains[5], fl_gamma=0.0, slide_ratio=0,
                 nwdloss=0, shapeloss=0, label_smoothing=0.0)
    crit = ComputeLoss(m)
    crit.cp, crit.cn, crit.gr = gains[6], gains[7], gains[8]
    crit.balance = list(balance)
    out, grads = crit._launch([t.detach().contiguous() for t in p], targets, need_grad)
    return [out] + (grads if need_grad else [])


@yolo_loss.register_fake
def _(p, targets, anchors, balance, gains, need_grad):
    return [p[0].new_empty(4)] + ([torch.empty_like(t) for t in p] if need_grad else [])
