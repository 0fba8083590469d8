"""Host-side mirror of the reference's DCNv3 extension and module, over libsomi_hip.so.

`dcnv3_forward` / `dcnv3_backward` keep the positional signatures of the reference's pybind module `DCNv3`
(models/ops_dcnv3/src/vision.cpp:15-16, src/dcnv3.h:20-59); `DCNv3Function` mirrors
models/ops_dcnv3/functions/dcnv3_func.py:19-61 and `DCNv3` mirrors modules/dcnv3.py:222-379 with the same
parameter names, so a reference state_dict loads unchanged.  Errors raise RuntimeError like AT_ASSERTM.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops
from .pack import pad4


def _checked(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f'{name} tensor has to be contiguous')          # dcnv3_cuda.cu:29-31
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')               # dcnv3_cuda.cu:32-34 (HIP device here)
    if t.dtype != torch.float32:
        raise RuntimeError(f'{name}: only float32 is implemented on the MI355X path')
    return t


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                  group, group_channels, offset_scale, im2col_step):
    """-> output (N,Ho,Wo,group*group_channels); src/dcnv3.h:20-26."""
    for n, t in (('input', input), ('offset', offset), ('mask', mask)):
        _checked(t, n)
    if input.shape[3] != group * group_channels:
        raise RuntimeError(f'Input channels and group times group channels wont match: '
                           f'({input.shape[3]} vs {group * group_channels}).')
    return ops.dcnv3_forward_raw(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                                 dilation_w, group, group_channels, offset_scale, im2col_step)


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                   group, group_channels, offset_scale, grad_output, im2col_step):
    """-> [grad_input, grad_offset, grad_mask]; src/dcnv3.h:40-47."""
    for n, t in (('input', input), ('offset', offset), ('mask', mask), ('grad_output', grad_output)):
        _checked(t, n)
    if input.shape[3] != group * group_channels:
        raise RuntimeError(f'Input channels and group times group channels wont match: '
                           f'({input.shape[3]} vs {group * group_channels}).')
    return list(ops.dcnv3_backward_raw(input, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h,
                                       pad_w, dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step))


class DCNv3Function(Function):
    """functions/dcnv3_func.py:19-61 (forward saves input/offset/mask; backward returns 3 grads + 12 None)."""

    @staticmethod
    def forward(ctx, input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                group, group_channels, offset_scale, im2col_step):
        ctx.cfg = (kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, group_channels,
                   offset_scale)
        ctx.im2col_step = im2col_step
        out = dcnv3_forward(input, offset, mask, *ctx.cfg, im2col_step)
        ctx.save_for_backward(input, offset, mask)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask = ctx.saved_tensors
        gi, go, gm = dcnv3_backward(input, offset, mask, *ctx.cfg, grad_output.contiguous(), ctx.im2col_step)
        return (gi, go, gm) + (None,) * 12


class _ToChannelsLast(nn.Module):
    def forward(self, x):
        return x.permute(0, 2, 3, 1)


class DCNv3(nn.Module):
    """DCNv3 layer, NHWC in / NHWC out (modules/dcnv3.py:222-379).

    Inference (`torch.no_grad`, eval) runs entirely on HIP kernels: the four Linear layers are 1x1 NHWC convs on the
    MFMA kernel, the depthwise conv + LayerNorm + GELU and the mask softmax are torch device ops on the same stream
    (widening target: fuse them), the deformable gather is `somi_dcnv3_forward_f32`.
    """

    def __init__(self, channels=64, kernel_size=3, dw_kernel_size=None, stride=1, pad=1, dilation=1, group=4,
                 offset_scale=1.0, act_layer='GELU', norm_layer='LN', center_feature_scale=False):
        super().__init__()
        if channels % group != 0:
            raise ValueError(f'channels must be divisible by group, but got {channels} and {group}')
        if act_layer != 'GELU' or norm_layer != 'LN':
            raise NotImplementedError('only the LN/GELU configuration is on the SOMI path')
        dwk = dw_kernel_size if dw_kernel_size is not None else kernel_size
        self.channels, self.kernel_size, self.dw_kernel_size = channels, kernel_size, dwk
        self.stride, self.dilation, self.pad = stride, dilation, pad
        self.group, self.group_channels = group, channels // group
        self.offset_scale, self.center_feature_scale = offset_scale, center_feature_scale
        self.dw_conv = nn.Sequential(nn.Conv2d(channels, channels, dwk, 1, (dwk - 1) // 2, groups=channels),
                                     nn.Sequential(_ToChannelsLast(), nn.LayerNorm(channels, eps=1e-6)), nn.GELU())
        K = kernel_size * kernel_size
        self.offset = nn.Linear(channels, group * K * 2)
        self.mask = nn.Linear(channels, group * K)
        self.input_proj = nn.Linear(channels, channels)
        self.output_proj = nn.Linear(channels, channels)
        for lin in (self.offset, self.mask):
            nn.init.constant_(lin.weight, 0.)
            nn.init.constant_(lin.bias, 0.)
        for lin in (self.input_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0.)
        if center_feature_scale:
            self.center_feature_scale_proj_weight = nn.Parameter(torch.zeros(group, channels))
            self.center_feature_scale_proj_bias = nn.Parameter(torch.zeros(group))

    def _linear(self, x, lin):
        """nn.Linear on NHWC == 1x1 conv; weight (out,in) is already [Cout][K]."""
        cin = lin.in_features
        if cin % 4:
            raise RuntimeError('DCNv3 channels must be a multiple of 4 on the MI355X path')
        return ops.conv2d_nhwc(x, lin.weight.detach().contiguous(), lin.bias.detach(), kh=1, kw=1)

    def forward(self, input):
        N, H, W, _ = input.shape
        x = self._linear(input, self.input_proj)
        x_proj = x
        x1 = self.dw_conv(input.permute(0, 3, 1, 2)).contiguous()
        offset = self._linear(x1, self.offset)
        mask = self._linear(x1, self.mask).reshape(N, H, W, self.group, -1)
        mask = F.softmax(mask, -1).reshape(N, H, W, -1).contiguous()
        x = DCNv3Function.apply(x, offset, mask, self.kernel_size, self.kernel_size, self.stride, self.stride, self.pad,
                                self.pad, self.dilation, self.dilation, self.group, self.group_channels,
                                self.offset_scale, 256)
        if self.center_feature_scale:
            cfs = torch.sigmoid(F.linear(x1, self.center_feature_scale_proj_weight, self.center_feature_scale_proj_bias))
            cfs = cfs[..., None].repeat(1, 1, 1, 1, self.channels // self.group).flatten(-2)
            x = x * (1 - cfs) + x_proj * cfs
        return self._linear(x, self.output_proj)
