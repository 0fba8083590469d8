"""Host-side mirror of the reference's DCNv3 extension and module, over libsomi_hip.so.

`dcnv3_forward` / `dcnv3_backward` keep the positional signatures of the reference's pybind module `DCNv3`
(models/ops_dcnv3/src/vision.cpp:15-16, src/dcnv3.h:20-59); `DCNv3Function` mirrors
models/ops_dcnv3/functions/dcnv3_func.py:19-61 and `DCNv3` mirrors modules/dcnv3.py:222-379 with the same
parameter names, so a reference state_dict loads unchanged.  Errors raise RuntimeError like AT_ASSERTM.
"""
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import ops


def _checked(t, name, like=None):
    if not t.is_contiguous():
        raise RuntimeError(f'{name} tensor has to be contiguous')          # dcnv3_cuda.cu:29-31
    if not t.is_cuda:
        raise RuntimeError(f'{name} must be a CUDA tensor')               # dcnv3_cuda.cu:32-34 (HIP device here)
    if t.dtype not in (torch.float32, torch.float16, torch.float64):      # AT_DISPATCH_FLOATING_TYPES_AND_HALF, dcnv3_cuda.cu:69
        raise RuntimeError(f'"dcnv3" not implemented for \'{t.dtype}\'')
    if like is not None and t.dtype != like.dtype:
        raise RuntimeError(f'{name}: expected scalar type {like.dtype} but found {t.dtype}')
    return t


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                  group, group_channels, offset_scale, im2col_step):
    """-> output (N,Ho,Wo,group*group_channels); src/dcnv3.h:20-26."""
    for n, t in (('input', input), ('offset', offset), ('mask', mask)):
        _checked(t, n, input)
    if input.shape[3] != group * group_channels:
        raise RuntimeError(f'Input channels and group times group channels wont match: '
                           f'({input.shape[3]} vs {group * group_channels}).')
    return ops.dcnv3_forward_raw(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h,
                                 dilation_w, group, group_channels, offset_scale, im2col_step)


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                   group, group_channels, offset_scale, grad_output, im2col_step):
    """-> [grad_input, grad_offset, grad_mask]; src/dcnv3.h:40-47."""
    for n, t in (('input', input), ('offset', offset), ('mask', mask), ('grad_output', grad_output)):
        _checked(t, n, input)
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

    Forward and backward run entirely on HIP kernels: with gradients enabled the whole module is one autograd node
    (`_DCNv3ModuleFunction`) whose backward chains the operator's backward kernel with the 1x1-conv data / weight gradients
    of the four Linear layers, the depthwise-conv, LayerNorm+GELU, softmax and centre-feature-scale backward kernels.
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

    def _linear(self, x, weight, bias, bn_stats=None):
        """nn.Linear on NHWC == 1x1 conv on the MFMA kernel; weight (out,in) is already [Cout][K].  Cout is padded to a
        multiple of 4 with zero rows (the pad columns are sliced off by the caller when it matters)."""
        cout, cin = weight.shape
        if cin % 4:
            raise RuntimeError('DCNv3 channels must be a multiple of 4 on the MI355X path')
        w, b = weight.detach().float(), bias.detach().float()
        if cout % 4:
            pad = 4 - cout % 4
            w = torch.cat([w, w.new_zeros(pad, cin)])
            b = torch.cat([b, b.new_zeros(pad)])
        return ops.conv2d_nhwc(x, w.contiguous(), b.contiguous(), kh=1, kw=1, bn_stats=bn_stats if cout % 4 == 0 else None)

    def _stacked_om(self):
        """The offset and mask Linear layers as ONE weight [2GK offset rows | GK mask rows | zero rows up to a multiple of 32] and bias: 216 -> 224
        columns at G 8, K 9, so that the data gradient through the stack (reduction over its columns) takes the conv kernel's uniform-tap fast
        path instead of the generic one (MFMA busy 0.53 there, profiles/r04_conv_pmc.txt); the pad columns of `om` hold zeros."""
        w = torch.cat([self.offset.weight.detach(), self.mask.weight.detach()]).float()
        b = torch.cat([self.offset.bias.detach(), self.mask.bias.detach()]).float()
        pad = -w.shape[0] % 32
        if pad:
            w = torch.cat([w, w.new_zeros(pad, w.shape[1])])
            b = torch.cat([b, b.new_zeros(pad)])
        return w.contiguous(), b.contiguous()

    def _params(self):
        dw, ln = self.dw_conv[0], self.dw_conv[1][1]
        ps = [dw.weight, dw.bias, ln.weight, ln.bias, self.offset.weight, self.offset.bias, self.mask.weight, self.mask.bias,
              self.input_proj.weight, self.input_proj.bias, self.output_proj.weight, self.output_proj.bias]
        if self.center_feature_scale:
            ps += [self.center_feature_scale_proj_weight, self.center_feature_scale_proj_bias]
        return ps

    def _forward_impl(self, input, keep=False, out_proj=None, bn_stats=None):
        """Every arithmetic step runs in libsomi_hip.so: 4 Linear layers = 1x1 MFMA convs, depthwise conv, LayerNorm+GELU,
        mask softmax, the deformable gather and the centre-feature-scale blend.  keep: also return what backward needs.
        out_proj = (weight, bias, act): replaces the output projection's parameters (a caller folding a following BatchNorm into it)
        and puts `act` into that conv's epilogue.  bn_stats = {'pivot': running_mean}: the output projection's epilogue also leaves the
        per-channel partial sums of the output for a following batch-statistics BatchNorm (ops.conv2d_nhwc(..., bn_stats=))."""
        N, H, W, C = input.shape
        if self.dw_kernel_size != 3:
            raise NotImplementedError('depthwise kernel size 3 only on the MI355X path')
        if not input.is_cuda or input.dtype != torch.float32:
            raise RuntimeError('DCNv3 runs on float32 GPU tensors only (no CPU fallback)')
        input = input.contiguous()
        x_proj = self._linear(input, self.input_proj.weight, self.input_proj.bias)
        dw = self.dw_conv[0]
        ln = self.dw_conv[1][1]
        wdw = dw.weight.detach().float()[:, 0].permute(1, 2, 0).reshape(9, C).contiguous()
        if C == 256 and ops.FUSE_DWLN:                           # one pass: a wave owns a pixel's 256 channels, the row statistics are wave reductions
            u, x1 = ops.dwconv3x3_ln(input, wdw, dw.bias.detach().float().contiguous(), ln.weight.detach().float().contiguous(),
                                     ln.bias.detach().float().contiguous(), ln.eps, 'gelu')
        else:
            u = ops.dwconv3x3(input, wdw, dw.bias.detach().float().contiguous())
            x1 = ops.layernorm_act(u, ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps, 'gelu')
        K = self.kernel_size * self.kernel_size
        GK = self.group * K
        cfg = (self.kernel_size, self.kernel_size, self.stride, self.stride, self.pad, self.pad, self.dilation, self.dilation, self.group,
               self.group_channels, self.offset_scale, 256)
        if GK % 4 == 0:
            # offset and mask logits share their input: ONE 1x1 GEMM over the stacked weights -> rows [2GK offsets | GK logits]; the softmax
            # runs in place on the logit columns and the operator reads both column ranges where they lie (modules/dcnv3.py:330-334)
            w_om, b_om = self._stacked_om()
            om = ops.conv2d_nhwc(x1, w_om, b_om, kh=1, kw=1)
            ops.group_softmax_cols_(om, self.group, K, 2 * GK)
            offset = mask = None
            y = ops.dcnv3_forward_merged(x_proj, om, *cfg)
        else:
            om = None
            offset = self._linear(x1, self.offset.weight, self.offset.bias)
            mlog = self._linear(x1, self.mask.weight, self.mask.bias)
            if offset.shape[-1] != GK * 2:
                offset = offset[..., :GK * 2].contiguous()
            if mlog.shape[-1] != GK:
                mlog = mlog[..., :GK].contiguous()
            mask = ops.group_softmax(mlog, K)
            y = dcnv3_forward(x_proj, offset, mask, *cfg)
        logit, yb = None, y
        if self.center_feature_scale:
            logit = self._linear(x1, self.center_feature_scale_proj_weight, self.center_feature_scale_proj_bias)
            yb = ops.cfs_blend(y, x_proj, logit, self.group, self.group_channels)
        if out_proj is not None:
            out = ops.conv2d_nhwc(yb, out_proj[0], out_proj[1], kh=1, kw=1, act=out_proj[2])
        else:
            out = self._linear(yb, self.output_proj.weight, self.output_proj.bias, bn_stats=bn_stats)
        if keep:
            return out, (input, x_proj, wdw, u, x1, om, y, logit, yb)
        return out

    def forward(self, input):
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            return _DCNv3ModuleFunction.apply(self, input, *self._params())
        return self._forward_impl(input)

    # ------------------------------------------------------------------------------------------ backward of the whole module
    def _backward_impl(self, saved, dout):
        """-> (dinput, [parameter gradients in the order of _params()])."""
        input, x_proj, wdw, u, x1, om, y, logit, yb = saved
        N, H, W, C = input.shape
        G, Gc, K = self.group, self.group_channels, self.kernel_size * self.kernel_size
        if (G * K) % 4 or (self.center_feature_scale and G % 4) or (H, W) != tuple(y.shape[1:3]):
            raise NotImplementedError('DCNv3 module backward: group*K (and group, with centre feature scale) must be multiples of 4, '
                                      'stride 1')
        dev = input.device
        dout = dout.contiguous().float()

        def lin_bwd(x, w, dy, need_dx=True):
            """y = x @ w.T + b as a 1x1 conv: -> (dx, dw (out,in), db)."""
            cout, cin = w.shape
            gw = ops.conv2d_wgrad_nhwc(x, dy, kh=1, kw=1, cin=cin, cout=cout)
            gb = torch.zeros(cout, device=dev)
            ops.chan_sum_(dy, cout, 0, gb)
            dx = None
            if need_dx:
                dx = ops.conv2d_dgrad_nhwc(dy, w.detach().float().t().contiguous(), B=N, H=x.shape[1], W=x.shape[2], cin=cin, kh=1, kw=1,
                                           cout=cout)
            return dx, gw, gb

        dyb, g_wout, g_bout = lin_bwd(yb, self.output_proj.weight, dout)
        g_wc = g_bc = None
        dx1 = None
        if self.center_feature_scale:
            dy, dxp, dlogit = ops.cfs_blend_backward(y, x_proj, logit, dyb, G, Gc)
            dx1, g_wc, g_bc = lin_bwd(x1, self.center_feature_scale_proj_weight, dlogit)
        else:
            dy, dxp = dyb, None
        # the operator writes grad_offset / grad_mask into one tensor laid out like `om`; the softmax backward runs in place on its mask
        # columns, and ONE weight-gradient / data-gradient pair over the stacked Linear weights finishes both layers
        dxp_op, d_om = ops.dcnv3_backward_merged(x_proj, om, dy.contiguous(), self.kernel_size, self.kernel_size, self.stride, self.stride,
                                                 self.pad, self.pad, self.dilation, self.dilation, G, Gc, self.offset_scale, 256)
        dxp = dxp_op if dxp is None else ops.add_(dxp, 0, dxp_op, 0, C)
        ops.group_softmax_backward_cols_(om, d_om, G, K, 2 * G * K)
        w_om, _ = self._stacked_om()
        d1, g_wom, g_bom = lin_bwd(x1, w_om, d_om)
        g_woff, g_wm = g_wom[:2 * G * K], g_wom[2 * G * K:3 * G * K]          # (rows beyond 3GK are the zero pad of the stack)
        g_boff, g_bm = g_bom[:2 * G * K], g_bom[2 * G * K:3 * G * K]
        dx1 = d1 if dx1 is None else ops.add_(dx1, 0, d1, 0, C)
        ln = self.dw_conv[1][1]
        g_lnw, g_lnb = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        du = ops.layernorm_gelu_backward(u, ln.weight.detach().float().contiguous(), ln.bias.detach().float().contiguous(), ln.eps, dx1,
                                         g_lnw, g_lnb)
        dinp, g_win, g_bin = lin_bwd(input, self.input_proj.weight, dxp)
        g_dw9, g_dwb = torch.zeros(9, C, device=dev), torch.zeros(C, device=dev)
        dinput = ops.dwconv3x3_backward(du, input, wdw, g_dw9, g_dwb, dx_accumulate=dinp)
        g_dw = g_dw9.view(3, 3, C).permute(2, 0, 1).unsqueeze(1).contiguous()
        grads = [g_dw, g_dwb, g_lnw, g_lnb, g_woff, g_boff, g_wm, g_bm, g_win, g_bin, g_wout, g_bout]
        if self.center_feature_scale:
            grads += [g_wc, g_bc]
        return dinput, grads


class _DCNv3ModuleFunction(Function):
    """The whole DCNv3 module as one autograd node (forward and backward on HIP kernels only)."""

    @staticmethod
    def forward(ctx, module, input, *params):
        out, saved = module._forward_impl(input.detach(), keep=True)
        ctx.module, ctx.saved = module, saved
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dout):
        dinput, grads = ctx.module._backward_impl(ctx.saved, dout)
        ctx.saved = None
        return (None, dinput) + tuple(grads)
