"""Oracle (test infrastructure): CPU restatement of the DCNv3 operator and module.

The operator follows the reference's CUDA kernel semantics
(models/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-80, 216-275): un-padded NHWC input,
padding folded into the base point, taps outside the image contribute zero, a sampling point
is used only if -1 < loc < size.  It is written with differentiable torch ops so autograd
yields the gradients the reference's backward kernels compute (:82-147, 278-370).
The reference's own debug path ``dcnv3_core_pytorch`` (functions/dcnv3_func.py:147-188, explicit
padding + F.grid_sample) agrees with this to rounding, which is exactly what the reference's
test asserts (models/ops_dcnv3/test.py:55,85) and what tests/test_oracle_golden.py re-checks
against vectors generated from ``dcnv3_core_pytorch``.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def dcnv3_out_size(size, k, s, p, d):
    """models/ops_dcnv3/src/cuda/dcnv3_cuda.cu:40-45."""
    return (size + 2 * p - (d * (k - 1) + 1)) // s + 1


def dcnv3_core(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
               dilation_h, dilation_w, group, group_channels, offset_scale):
    """input (N,H,W,G*Gc), offset (N,Ho,Wo,G*K*2) [x,y interleaved], mask (N,Ho,Wo,G*K) -> (N,Ho,Wo,G*Gc).

    Point order inside K: kernel_w outer, kernel_h inner (dcnv3_im2col_cuda.cuh:253-254).
    """
    N, H, W, C = input.shape
    G, Gc, K = group, group_channels, kernel_h * kernel_w
    assert C == G * Gc
    Ho = dcnv3_out_size(H, kernel_h, stride_h, pad_h, dilation_h)
    Wo = dcnv3_out_size(W, kernel_w, stride_w, pad_w, dilation_w)
    assert offset.shape == (N, Ho, Wo, G * K * 2) and mask.shape == (N, Ho, Wo, G * K)
    dt, dev = input.dtype, input.device
    half_w, half_h = (dilation_w * (kernel_w - 1)) >> 1, (dilation_h * (kernel_h - 1)) >> 1
    # base points: p0 - half*offset_scale  (:249-252)
    base_w = (half_w - pad_w + torch.arange(Wo, device=dev) * stride_w).to(dt) - half_w * offset_scale
    base_h = (half_h - pad_h + torch.arange(Ho, device=dev) * stride_h).to(dt) - half_h * offset_scale
    ki = torch.arange(kernel_w, device=dev).repeat_interleave(kernel_h).to(dt) * dilation_w   # i outer
    kj = torch.arange(kernel_h, device=dev).repeat(kernel_w).to(dt) * dilation_h             # j inner
    off = offset.view(N, Ho, Wo, G, K, 2)
    loc_w = base_w.view(1, 1, Wo, 1, 1) + (ki.view(1, 1, 1, 1, K) + off[..., 0]) * offset_scale
    loc_h = base_h.view(1, Ho, 1, 1, 1) + (kj.view(1, 1, 1, 1, K) + off[..., 1]) * offset_scale
    use = (loc_h > -1) & (loc_w > -1) & (loc_h < H) & (loc_w < W)                          # :262-263
    h0, w0 = torch.floor(loc_h), torch.floor(loc_w)
    lh, lw = loc_h - h0, loc_w - w0
    h0, w0 = h0.long(), w0.long()
    src = input.view(N, H * W, G, Gc).permute(0, 2, 1, 3)                                   # N,G,HW,Gc
    m = mask.view(N, Ho, Wo, G, K)
    acc = torch.zeros(N, Ho, Wo, G, Gc, dtype=dt, device=dev)
    for dh_, dw_, wt in ((0, 0, (1 - lh) * (1 - lw)), (0, 1, (1 - lh) * lw), (1, 0, lh * (1 - lw)), (1, 1, lh * lw)):
        hh, ww = h0 + dh_, w0 + dw_
        inside = use & (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)                # :57-75
        lin = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1))                                 # N,Ho,Wo,G,K
        idx = lin.permute(0, 3, 1, 2, 4).reshape(N, G, Ho * Wo * K, 1).expand(-1, -1, -1, Gc)
        val = torch.gather(src, 2, idx).view(N, G, Ho, Wo, K, Gc).permute(0, 2, 3, 1, 4, 5)  # N,Ho,Wo,G,K,Gc
        coef = (wt * m * inside.to(dt)).unsqueeze(-1)
        acc = acc + (coef * val).sum(dim=4)
    return acc.reshape(N, Ho, Wo, C)


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                  dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step):
    """Same positional signature as the reference extension's ``dcnv3_forward`` (src/dcnv3.h:20-26)."""
    _check(input, offset, mask, group, group_channels, im2col_step)
    with torch.no_grad():
        return dcnv3_core(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                          dilation_h, dilation_w, group, group_channels, offset_scale)


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                   dilation_h, dilation_w, group, group_channels, offset_scale, grad_output, im2col_step):
    """``dcnv3_backward`` (src/dcnv3.h:40-47): returns [grad_input, grad_offset, grad_mask]."""
    _check(input, offset, mask, group, group_channels, im2col_step)
    with torch.enable_grad():
        i, o, m = (t.detach().clone().requires_grad_(True) for t in (input, offset, mask))
        out = dcnv3_core(i, o, m, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                         dilation_h, dilation_w, group, group_channels, offset_scale)
        gi, go, gm = torch.autograd.grad(out, (i, o, m), grad_output)
    return [gi, go, gm]


def _check(input, offset, mask, group, group_channels, im2col_step):
    """Argument checks of the reference host launcher (src/cuda/dcnv3_cuda.cu:29-53)."""
    for name, t in (('input', input), ('offset', offset), ('mask', mask)):
        if not t.is_contiguous():
            raise RuntimeError(f'{name} tensor has to be contiguous')
    batch = input.shape[0]
    step = min(batch, im2col_step)
    if batch % step != 0:
        raise RuntimeError(f'batch({batch}) must divide im2col_step({step})')
    if input.shape[3] != group * group_channels:
        raise RuntimeError(f'Input channels and group times group channels wont match: '
                           f'({input.shape[3]} vs {group * group_channels}).')


class _CL(nn.Module):
    """NCHW -> NHWC view (modules/dcnv3.py:32-38), parameter-free."""

    def forward(self, x):
        return x.permute(0, 2, 3, 1)


class DCNv3(nn.Module):
    """DCNv3 layer on NHWC input (models/ops_dcnv3/modules/dcnv3.py:222-379).

    input_proj Linear || depthwise kxk conv -> LayerNorm -> GELU -> {offset Linear, mask Linear + softmax over K}
    -> dcnv3 op -> (optional centre-feature-scale blend) -> output_proj Linear.
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

        self.dw_conv = nn.Sequential(
            nn.Conv2d(channels, channels, dwk, 1, (dwk - 1) // 2, groups=channels),
            nn.Sequential(_CL(), nn.LayerNorm(channels, eps=1e-6)),
            nn.GELU())
        K = kernel_size * kernel_size
        self.offset = nn.Linear(channels, group * K * 2)
        self.mask = nn.Linear(channels, group * K)
        self.input_proj = nn.Linear(channels, channels)
        self.output_proj = nn.Linear(channels, channels)
        for lin in (self.offset, self.mask):      # :303-311
            nn.init.constant_(lin.weight, 0.)
            nn.init.constant_(lin.bias, 0.)
        for lin in (self.input_proj, self.output_proj):
            nn.init.xavier_uniform_(lin.weight)
            nn.init.constant_(lin.bias, 0.)
        if center_feature_scale:
            self.center_feature_scale_proj_weight = nn.Parameter(torch.zeros(group, channels))
            self.center_feature_scale_proj_bias = nn.Parameter(torch.zeros(group))

    def forward(self, input):
        N, H, W, _ = input.shape
        x = self.input_proj(input)
        x_proj = x
        x1 = self.dw_conv(input.permute(0, 3, 1, 2))
        offset = self.offset(x1)
        mask = F.softmax(self.mask(x1).reshape(N, H, W, self.group, -1), -1).reshape(N, H, W, -1).type(x.dtype)
        x = dcnv3_core(x, offset, mask, self.kernel_size, self.kernel_size, self.stride, self.stride,
                       self.pad, self.pad, self.dilation, self.dilation, self.group, self.group_channels,
                       self.offset_scale)
        if self.center_feature_scale:            # :370-376
            cfs = torch.sigmoid(F.linear(x1, self.center_feature_scale_proj_weight,
                                         self.center_feature_scale_proj_bias))
            cfs = cfs[..., None].repeat(1, 1, 1, 1, self.channels // self.group).flatten(-2)
            x = x * (1 - cfs) + x_proj * cfs
        return self.output_proj(x)
