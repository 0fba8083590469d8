"""SOMI building blocks executing on libsomi_hip.so (NHWC fp32, MI355X).

Each class keeps the reference's constructor signature and parameter names (so reference-format yaml and
state_dicts load unchanged) but its forward launches HIP kernels through the C ABI:
Conv+BN+SiLU is ONE implicit-GEMM launch with the BN folded into the packed weights (eval), `torch.cat` never
happens (producers write channel slices of the consumer's buffer), nearest upsampling is folded into the BiFPN
read and the CBAM scaling into the following conv's operand load.

Activations travel as `Act` = (NHWC tensor, channel offset, logical channels).  Channel strides are multiples of 4;
padded channels hold zeros (zero weight rows in the producer), so consumers may read them.
Reference citations are relative to /root/reference.
"""
import math

import torch
import torch.nn as nn

from . import ops
from .pack import pack_conv_weight, pack_dgrad_weight, pad4, bn_fold


class Act:
    """A channel slice [coff, coff+c) of an NHWC tensor."""
    __slots__ = ('t', 'coff', 'c', 'up', 'pooled', 'pool')

    def __init__(self, t, coff=0, c=None, up=0):
        self.t, self.coff, self.c, self.up = t, coff, (t.shape[3] - coff if c is None else c), up
        # a GRADIENT may carry a part that is constant over an image's pixels and not yet added to the tensor: (davg, dmax, amaxp) with the meaning of
        # ops.bn_act_backward's `pooled` (value = t + davg / HW ...).  Whoever consumes it folds it in (Conv.backward) or adds it (settle_pooled)
        self.pooled = None
        self.pool = None          # an ACTIVATION may carry its global (average, max) per (image, channel) when the pass that wrote it took them (Conv(pool=))

    @property
    def shape(self):
        return self.t.shape

    def slice(self, coff, c):
        return Act(self.t, self.coff + coff, c)


def settle_pooled(a):
    """Adds a gradient's pending pooled part (Act.pooled) to its tensor."""
    if a is not None and a.pooled is not None:
        ops.pool_backward_add_(a.t, a.coff, a.c, *a.pooled)
        a.pooled = None
    return a


def new_act(like, H, W, c):
    return Act(torch.empty(like.shape[0], H, W, pad4(c), device=like.device, dtype=torch.float32), 0, c)


def concat_act(like, H, W, c):
    """The buffer several producers write channel slices of (the reference's torch.cat, never materialised).  Stored at pad4(c) channels like every
    activation: a total that is >= 64 and not a multiple of 32 (5 x 16 = 80 in a C2fCBAM of hidden width 16, n = 3) gets zero pad channels the
    consumer conv may read (its weight columns there are zero).  Round 4: the buffers were allocated at exactly c and such a width failed the
    consumer's channel-range check (found by the n = 3 case of test_c2fcbam_train_forward_backward; no shipped graph has such a width)."""
    cp = pad4(c)
    alloc = torch.empty if cp == c else torch.zeros
    return Act(alloc(like.shape[0], H, W, cp, device=like.device, dtype=torch.float32), 0, c)


def autopad(k, p=None, d=1):
    """models/common.py:43-51."""
    if p is not None:
        return p
    return (d * (k - 1) + 1) // 2 if d > 1 else k // 2


class _Packed(nn.Module):
    """Mixin: device-side packed parameters are rebuilt lazily after any parameter change."""

    def _packed(self, dev):
        # in-place edits through torch (load_state_dict, copy_) bump the version counters; the fused optimizer writes through
        # raw pointers and calls Model.invalidate() itself
        key = (dev, self.training, tuple(p._version for p in self.parameters()))
        cache = self.__dict__.get('_pk')
        if cache is None or cache[0] != key:
            with torch.no_grad():
                cache = (key, self._pack(dev))
            self.__dict__['_pk'] = cache
        return cache[1]

    def invalidate(self):
        self.__dict__.pop('_pk', None)


def _acc_grad(param, g):
    """param.grad += g (reference layout, like autograd's accumulation)."""
    g = g.detach().to(param.dtype)
    if param.grad is None:
        param.grad = g.clone().contiguous()
    else:
        param.grad.add_(g)


def _grad_target(param):
    """Buffer a backward kernel may accumulate into directly: the parameter's gradient when it is a contiguous fp32 tensor
    (the optimizer's flat views are), else a zero scratch that the caller adds with _acc_grad.  -> (buffer, is_scratch)"""
    g = param.grad
    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g, False
    return torch.zeros_like(param, dtype=torch.float32), True


def _grad_targets(*params):
    """_grad_target for several parameters -> (buffers, finish): the kernels accumulate into `buffers`, finish() adds the scratch ones (if any) to .grad."""
    tg = [_grad_target(p_) for p_ in params]

    def finish():
        for p_, (buf, scratch) in zip(params, tg):
            if scratch:
                _acc_grad(p_, buf)
    return [t[0] for t in tg], finish


_NBT_PENDING = None      # Model.forward collects the BatchNorm step counters here and bumps them with one multi-tensor add


def _bump_batches_tracked(bn):
    if _NBT_PENDING is not None:
        _NBT_PENDING.append(bn.num_batches_tracked)
    else:
        bn.num_batches_tracked += 1


class collect_batches_tracked:
    """Context manager: BatchNorm step counters touched inside are incremented together on exit."""

    def __enter__(self):
        global _NBT_PENDING
        self.outer, _NBT_PENDING = _NBT_PENDING, []
        return self

    def __exit__(self, *exc):
        global _NBT_PENDING
        pending, _NBT_PENDING = _NBT_PENDING, self.outer
        if pending and exc[0] is None:
            torch._foreach_add_(pending, 1)
        return False


def _act_name(m):
    if isinstance(m, nn.SiLU):
        return 'silu'
    if isinstance(m, nn.Identity):
        return 'none'
    if isinstance(m, nn.ReLU):
        return 'relu'
    if isinstance(m, nn.GELU):
        return 'gelu'
    raise NotImplementedError(f'activation {type(m).__name__} is not on the SOMI path')


def _fold_conv_bn(conv, bn):
    """W' = diag(gamma/sqrt(var+eps)) W, b' = beta - gamma*mean/sqrt(var+eps) (+ scaled conv bias)
    (utils/torch_utils.py:202-222)."""
    w = conv.weight.detach().float()
    b = conv.bias.detach().float() if conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
    if bn is not None:
        s, t = bn_fold(bn)
        w = w * s.view(-1, 1, 1, 1)
        b = b * s + t
    return w, b


def _pack_wb(w, b, dev):
    """-> packed weight [pad4(Cout)][k*k*pad4(Cin)], bias [pad4(Cout)] on dev (pads zero)."""
    cout = w.shape[0]
    wp = pack_conv_weight(w, cout_pad=pad4(cout)).to(dev)
    bp = torch.zeros(pad4(cout), device=dev)
    bp[:cout] = b.to(dev)
    return wp, bp


class Conv(_Packed):
    """conv2d(bias=False) -> BatchNorm2d -> SiLU as one fused launch (models/common.py:53-70)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        if g != 1 or d != 1:
            raise NotImplementedError('grouped / dilated Conv is not on the SOMI path')
        if isinstance(k, (tuple, list)):                         # C3 hands its bottlenecks (1, 1) / (3, 3) (models/common.py:1558)
            if len(k) != 2 or k[0] != k[1]:
                raise NotImplementedError('square kernels only')
            k = int(k[0])
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    def _pack(self, dev):
        if self.training:                                        # raw weights, batch-norm applied from batch statistics
            w = self.conv.weight
            c2, c1, k = w.shape[0], w.shape[1], w.shape[2]
            cp, c1p = pad4(c2), pad4(c1)
            master = self.__dict__.get('_master')               # (weights, gradients) [cp][k*k*c1p] inside the optimizer's flat buffers
            if master is not None and master[0].device == dev:
                wp, gp = master
                wt = ops.pack_dgrad_weights(wp, cp, k * k, c1p)
            else:
                wf = w.detach().float()
                wp, gp, wt = pack_conv_weight(wf, cout_pad=cp).to(dev), None, pack_dgrad_weight(wf).to(dev)
            bn = self.bn
            if cp == c2 and bn.weight.device == dev:             # no padding: the module's own tensors, running statistics in place
                vec = dict(gamma=bn.weight.detach(), beta=bn.bias.detach(), rm=bn.running_mean, rv=bn.running_var, inplace=True)
            else:
                padv = lambda t, fill: torch.cat([t.detach().float().to(dev), torch.full((cp - c2,), fill, device=dev)])   # noqa: E731
                vec = dict(gamma=padv(bn.weight, 0.), beta=padv(bn.bias, 0.), rm=padv(bn.running_mean, 0.), rv=padv(bn.running_var, 1.),
                           inplace=False)
            return dict(wp=wp, gp=gp, wt=wt, **vec)
        return _pack_wb(*_fold_conv_bn(self.conv, getattr(self, 'bn', None)), dev)

    # ------------------------------------------------------------------------------------------ training mode
    def _forward_train(self, x, out, residual, pool=None):
        """y = conv(x) (raw) -> batch statistics -> z = act(y*scale+shift) [+ residual]; keeps what backward needs.
        pool: a dict - when the shape allows, the pass that writes z also takes its global average / max pools per (image, channel) and leaves
        them as pool['avg'], pool['max'] (a channel attention behind this block then needs no pass of its own over z)."""
        pk = self._packed(x.t.device)                            # rebuilt after every optimizer step (Model.invalidate)
        k, s, p = self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0]
        c2, cp = self.conv.out_channels, pad4(self.conv.out_channels)
        B, H, W, _ = x.shape
        Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
        y = torch.empty(B, Ho, Wo, cp, device=x.t.device, dtype=torch.float32)
        if ops.FUSED_BN_STATS and y.numel() * 4 <= 0xE0000000:      # the conv epilogue leaves the per-channel partial sums of y
            st = {'pivot': pk['rm']}
            ops.conv2d_nhwc(x.t, pk['wp'], None, kh=k, kw=k, stride=s, pad=p, act='none', cin=pad4(x.c), x_coff=x.coff, out=y, cout=cp,
                            alg_cin=x.c, alg_cout=c2, bn_stats=st)
            mean, rstd, scale, shift = ops.bn_stats_from_partials(st['part'], st['rows'], B * Ho * Wo, cp, pk['gamma'], pk['beta'],
                                                                  self.bn.eps, self.bn.momentum, pk['rm'], pk['rv'])
        else:
            ops.conv2d_nhwc(x.t, pk['wp'], None, kh=k, kw=k, stride=s, pad=p, act='none', cin=pad4(x.c), x_coff=x.coff, out=y, cout=cp,
                            alg_cin=x.c, alg_cout=c2)
            mean, rstd, scale, shift = ops.bn_stats(y, cp, 0, pk['gamma'], pk['beta'], self.bn.eps, self.bn.momentum, pk['rm'], pk['rv'])
        with torch.no_grad():                                    # running statistics back into the module buffers
            if not pk['inplace']:
                self.bn.running_mean.copy_(pk['rm'][:c2])
                self.bn.running_var.copy_(pk['rv'][:c2])
            _bump_batches_tracked(self.bn)
        if out is None:
            out = new_act(x.t, Ho, Wo, c2)
        elif c2 % 4:
            raise NotImplementedError('writing into a channel slice needs c2 % 4 == 0')
        cw = cp if out.coff == 0 and out.t.shape[3] == cp else c2
        pooled = None
        if pool is not None and residual is None and cw == c2 and isinstance(self.act, nn.SiLU):
            pooled = ops.affine_silu_pool(y, c2, 0, scale, shift, out.t, out.coff)
        if pooled is not None:
            pool['avg'], pool['max'] = pooled
        elif residual is not None and cw != c2:
            ops.chan_affine_act(y, cw, 0, scale, shift, _act_name(self.act), 0, out.t, out.coff)
            ops.add_(out.t, out.coff, residual.t, residual.coff, c2)
        else:
            ops.chan_affine_act(y, cw, 0, scale, shift, _act_name(self.act), 0, out.t, out.coff,
                                residual=None if residual is None else residual.t, res_coff=0 if residual is None else residual.coff)
        self.__dict__['_ctx'] = (x, y, mean, rstd, scale, shift, pk)
        return Act(out.t, out.coff, c2)

    def backward(self, dz, dx_out=None, accumulate=False, need_dx=True, also_add=None, pooled=None, cbam=None):
        """dz: gradient w.r.t. this block's output (Act).  Returns the gradient w.r.t. the input as an Act (written into
        dx_out if given, added to it if accumulate; `also_add` is a further Act added in the same pass - a shortcut's
        gradient).  Parameter gradients are accumulated into .grad (reference layout).
        pooled: (davg, dmax, amaxp) of a channel attention that pooled this block's output (ops.bn_act_backward): its gradient joins dz inside
        the BatchNorm backward kernels.
        cbam: state of ops.cbam_backward(bn=...) - dz is then the gradient w.r.t. t*ca*sa of the CBAM bottleneck this conv opens, and the attention's
        step C runs inside the BatchNorm backward kernels too (pooled = the MLP backward's (davg, dmax, amaxp) on that call's dca)."""
        x, y, mean, rstd, scale, shift, pk = self.__dict__.pop('_ctx')
        k, s, p = self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0]
        c1, c2, cp = self.conv.in_channels, self.conv.out_channels, pad4(self.conv.out_channels)
        dev = y.device
        dy = torch.zeros_like(y) if cp != c2 else torch.empty_like(y)
        cw = cp if (dz.coff == 0 and dz.t.shape[3] == cp) else c2     # whole padded tensor, or an aligned slice
        if cw % 4:
            raise NotImplementedError('training backward on a channel slice needs out_channels % 4 == 0')
        gw, gb = self.bn.weight.grad, self.bn.bias.grad
        direct = cp == c2 and gw is not None and gb is not None and gw.is_contiguous() and gb.is_contiguous() and gw.device == dev
        if cbam is not None:                                      # CBAM's step C + pooled gradients + BatchNorm backward in two passes over (dz, y)
            if cw != cp or cp != c2:
                raise NotImplementedError('fused CBAM backward works on whole, unpadded tensors')
            dgam, dbet = (gw, gb) if direct else (torch.zeros(cp, device=dev), torch.zeros(cp, device=dev))
            ops.cbam_bn_backward_apply(cbam, rstd, pooled[0], pooled[1], dy, dgam, dbet)
            if not direct:
                _acc_grad(self.bn.weight, dgam)
                _acc_grad(self.bn.bias, dbet)
        elif direct:                                              # the kernel accumulates straight into the gradient buffers
            ops.bn_act_backward(dz.t, dz.coff, y, 0, cw, mean, rstd, scale, shift, _act_name(self.act), 0, True, dy, 0, gw, gb, pooled=pooled)
        else:
            dgam, dbet = torch.zeros(cp, device=dev), torch.zeros(cp, device=dev)
            ops.bn_act_backward(dz.t, dz.coff, y, 0, cw, mean, rstd, scale, shift, _act_name(self.act), 0, True, dy, 0, dgam, dbet, pooled=pooled)
            _acc_grad(self.bn.weight, dgam[:c2])
            _acc_grad(self.bn.bias, dbet[:c2])
        B, H, W, _ = x.shape
        if pk['gp'] is not None:                                  # accumulate into the packed gradient master
            ops.conv2d_wgrad_nhwc(x.t, dy, kh=k, kw=k, stride=s, pad=p, cin=pad4(c1), x_coff=x.coff, cout=cp, out=pk['gp'],
                                  accumulate=pk['gp'])
        else:
            dw = ops.conv2d_wgrad_nhwc(x.t, dy, kh=k, kw=k, stride=s, pad=p, cin=pad4(c1), x_coff=x.coff, cout=cp)
            _acc_grad(self.conv.weight, dw.view(cp, k, k, pad4(c1))[:c2, :, :, :c1].permute(0, 3, 1, 2))
        if not need_dx:
            return None
        if dx_out is None:
            dx_out = Act(torch.empty(B, H, W, pad4(c1), device=dev, dtype=torch.float32), 0, c1)
        ops.conv2d_dgrad_nhwc(dy, pk['wt'], B=B, H=H, W=W, cin=pad4(c1), kh=k, kw=k, stride=s, pad=p, cout=cp, out=dx_out.t,
                              dx_coff=dx_out.coff, accumulate=dx_out.t if accumulate else None, acc_coff=dx_out.coff,
                              accumulate2=None if also_add is None else also_add.t,
                              acc2_coff=0 if also_add is None else also_add.coff)
        return dx_out

    def forward(self, x, out=None, residual=None, a_chan=None, a_pix=None, pool=None):
        if self.training:
            return self._forward_train(x, out, residual, pool)
        wp, bp = self._packed(x.t.device)
        k, s, p = self.conv.kernel_size[0], self.conv.stride[0], self.conv.padding[0]
        c2 = self.conv.out_channels
        B, H, W, _ = x.shape
        Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
        if out is None:
            out = new_act(x.t, Ho, Wo, c2)
            cout = pad4(c2)
        else:
            if c2 % 4:
                raise NotImplementedError('writing into a channel slice needs c2 % 4 == 0')
            cout = c2
        ops.conv2d_nhwc(x.t, wp[:cout], bp, kh=k, kw=k, stride=s, pad=p, act=_act_name(self.act), cin=pad4(x.c),
                        x_coff=x.coff, out=out.t, cout=cout, y_coff=out.coff,
                        residual=None if residual is None else residual.t,
                        res_coff=0 if residual is None else residual.coff, a_chan_scale=a_chan, a_pix_scale=a_pix,
                        alg_cin=x.c, alg_cout=c2)
        return Act(out.t, out.coff, c2)


class PlainConv(_Packed):
    """nn.Conv2d with bias, no norm / activation (Decouple.b3 / c3, models/yolo.py:1057,1063)."""

    def __init__(self, conv):
        super().__init__()
        self.conv = conv

    def _pack(self, dev):
        return _pack_wb(*_fold_conv_bn(self.conv, None), dev)

    def forward(self, x):
        if self.training:
            self.invalidate()
        wp, bp = self._packed(x.t.device)
        k = self.conv.kernel_size[0]
        B, H, W, _ = x.shape
        out = new_act(x.t, H, W, self.conv.out_channels)
        ops.conv2d_nhwc(x.t, wp, bp, kh=k, kw=k, stride=1, pad=k // 2, act='none', cin=pad4(x.c), x_coff=x.coff,
                        out=out.t, cout=pad4(self.conv.out_channels), alg_cin=x.c, alg_cout=self.conv.out_channels)
        if self.training:
            self.__dict__['_ctx'] = x
        return out

    def backward(self, dy):
        """dy: whole padded gradient tensor (pad channels zero).  Returns dx (Act)."""
        x = self.__dict__.pop('_ctx')
        k = self.conv.kernel_size[0]
        c1, c2 = self.conv.in_channels, self.conv.out_channels
        cp = pad4(c2)
        B, H, W, _ = x.shape
        dw = ops.conv2d_wgrad_nhwc(x.t, dy, kh=k, kw=k, stride=1, pad=k // 2, cin=pad4(c1), x_coff=x.coff, cout=cp)
        _acc_grad(self.conv.weight, dw.view(cp, k, k, pad4(c1))[:c2, :, :, :c1].permute(0, 3, 1, 2))
        db = torch.zeros(cp, device=dy.device)
        ops.chan_sum_(dy, cp, 0, db)
        _acc_grad(self.conv.bias, db[:c2])
        wt = pack_dgrad_weight(self.conv.weight.detach().float()).to(dy.device)
        dx = Act(torch.empty(B, H, W, pad4(c1), device=dy.device, dtype=torch.float32), 0, c1)
        ops.conv2d_dgrad_nhwc(dy, wt, B=B, H=H, W=W, cin=pad4(c1), kh=k, kw=k, stride=1, pad=k // 2, cout=cp, out=dx.t)
        return dx


class ChannelAttentionModule(_Packed):
    """sigmoid(MLP(GAP) + MLP(GMP)) (models/common.py:339-358) -> (B,C) scale vector."""

    def __init__(self, c1, reduction=16):
        super().__init__()
        mid = c1 // reduction
        self.shared_MLP = nn.Sequential(nn.Linear(c1, mid), nn.ReLU(), nn.Linear(mid, c1))

    def _pack(self, dev):
        l1, l2 = self.shared_MLP[0], self.shared_MLP[2]
        return tuple(t.detach().float().contiguous().to(dev) for t in (l1.weight, l1.bias, l2.weight, l2.bias))

    def forward(self, x, pooled=None):
        """pooled: (avg, max) of x per (image, channel) when the producer of x already took them (Conv(..., pool=))."""
        if self.training:
            self.invalidate()
        W1, b1, W2, b2 = self._packed(x.t.device)
        avg, mx = pooled if pooled is not None else ops.global_pool(x.t, c=x.c, x_coff=x.coff)
        ca = ops.attn_mlp(0, avg, mx, W1, b1, W2, b2)
        if self.training:
            self.__dict__['_ctx'] = (x, avg, mx, ca, (W1, b1, W2, b2))
        return ca

    def backward(self, dca, dt, amaxp=None, defer=False):
        """dca (B,C): gradient w.r.t. the attention vector; adds the pooled-input gradient into dt (Act) in place - or, defer=True, returns
        (davg, dmax, amaxp) for the producer's BatchNorm backward to fold in (Conv.backward(pooled=...)) and leaves dt alone.
        amaxp (B,C) int32: first pixel of every channel's spatial maximum when the caller already has it (the CBAM backward takes it in its own
        pass over x); else it is found here."""
        x, avg, mx, ca, (W1, b1, W2, b2) = self.__dict__.pop('_ctx')
        l1, l2 = self.shared_MLP[0], self.shared_MLP[2]
        prms = (l1.weight, l1.bias, l2.weight, l2.bias)
        g = [_grad_target(prm) for prm in prms]                  # the kernel accumulates; same layout as the parameters
        davg, dmax = ops.attn_mlp_backward(0, dca, ca, avg, mx, W1, b1, W2, g[0][0], g[1][0], g[2][0], g[3][0])
        for prm, (gr, scratch) in zip(prms, g):
            if scratch:
                _acc_grad(prm, gr)
        if amaxp is None:
            amaxp = ops.pool_argmax(x.t, x.c, x.coff)
        if defer:
            return davg, dmax, amaxp
        ops.pool_backward_add_(dt.t, dt.coff, x.c, davg, dmax, amaxp)


class SpatialAttentionModule(_Packed):
    """sigmoid(conv_kxk([mean_c, max_c])) (models/common.py:392-405) -> (B,H,W) scale map of ca*x."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in (3, 5, 7)
        self.cv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2)

    def _pack(self, dev):
        w = self.cv1.weight.detach().float()[0].permute(1, 2, 0).contiguous().to(dev)     # [k][k][2]
        return w, self.cv1.bias.detach().float().contiguous().to(dev)     # stays on the device: no host sync per repack

    def forward(self, x, ca):
        """Applies both attentions: x <- x * ca * sa.  Eval: in place (one stats pass + one apply pass).  Train: x is kept
        (backward needs it) and the product goes to a new tensor."""
        if self.training:
            self.invalidate()
        w, b = self._packed(x.t.device)
        stats = ops.chan_stats(x.t, ca, c=x.c, x_coff=x.coff)
        k = self.cv1.kernel_size[0]
        if not self.training:
            ops.cbam_apply_(x.t, ca, stats, w, b, k, c=x.c, x_coff=x.coff)
            return x
        if x.coff != 0 or x.t.shape[3] != x.c:
            raise NotImplementedError('CBAM training path works on whole tensors')
        sa = ops.spatial_attn(stats, w, b, k)
        out = ops.scale_channels(x.t, ca, sa)
        self.__dict__['_ctx'] = (x, ca, stats, sa, w)
        return Act(out, 0, x.c)

    def backward(self, dt2, t_max=None, bn=None):
        """dt2: gradient tensor w.r.t. x*ca*sa (whole tensor, modified in place into the x-gradient through both products and
        the spatial branch).  Returns dca (B,C) and - with t_max (B,C), the spatial maximum of x the channel attention pooled - amaxp (B,C): the
        first pixel holding each channel's maximum (for the max-pool's gradient), else None.
        bn = (y, scale, shift, mean) of the Conv that produced x: dt2 is left as it is and a third value - the state for that Conv's
        backward(cbam=...) - is returned (ops.cbam_backward)."""
        x, ca, stats, sa, w = self.__dict__.pop('_ctx')
        k = self.cv1.kernel_size[0]
        (dw, sw), (db, sb) = _grad_target(self.cv1.weight), _grad_target(self.cv1.bias)    # the kernel accumulates in nn.Conv2d's (1,2,k,k) layout
        res = ops.cbam_backward(dt2, x.t, x.coff, x.c, ca, sa, stats, w, k, dw, db, t_max=t_max, dw_chw=True, bn=bn)
        if sw:
            _acc_grad(self.cv1.weight, dw)
        if sb:
            _acc_grad(self.cv1.bias, db)
        return res


class CBAMBottleneck(nn.Module):
    """cv1 3x3 -> channel attn -> spatial attn -> cv2 3x3 (+x) (models/common.py:671-691).
    The two attention scalings are applied to cv1's output in place by one streaming kernel (the in-operand form of
    the conv kernel measured 26 % slower than the plain one; the extra pass costs ~2 %)."""

    def __init__(self, c1, c2, shortcut=True, g=1, e=1.0, k=(3, 3), ratio=8, kernel_size=3):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=1)
        self.add = shortcut and c1 == c2
        self.channel_attention = ChannelAttentionModule(c_, ratio)
        self.spatial_attention = SpatialAttentionModule(kernel_size)

    def forward(self, x, out=None):
        pool = {} if (self.training and ops.FUSE_POOL) else None
        t = self.cv1(x, pool=pool)                                # training: the BatchNorm + SiLU pass also takes the attention's global pools
        ca = self.channel_attention(t, pooled=(pool['avg'], pool['max']) if pool else None)
        t2 = self.spatial_attention(t, ca)
        if self.training:
            self.__dict__['_ctx'] = (x, t)
        return self.cv2(t2, out=out, residual=x if self.add else None)

    def backward(self, dout, dx_out):
        """dout: gradient w.r.t. the block output (Act); the input gradient is ADDED into dx_out (Act, e.g. a slice of the
        C2f gradient buffer that already holds the gradient of the input's other consumers)."""
        x, t = self.__dict__.pop('_ctx')
        d = self.cv2.backward(dout)                               # d(t*ca*sa)
        # d.t then holds the direct part of dt; the channel attention's pooled maximum lets the same pass find its arg-max pixels
        t_max = self.channel_attention.__dict__['_ctx'][2] if ops.AMAX_BY_VALUE else None
        c_ = self.cv1.conv.out_channels
        bn = None
        if (ops.CBAM_FUSED_BN and ops.BN_POOLED and ops.SYNC_BN is None and t_max is not None and isinstance(self.cv1.act, nn.SiLU) and pad4(c_) == c_
                and d.coff == 0 and d.t.shape[3] == c_ and t.coff == 0):
            _, y1, mean1, _, scale1, shift1, _ = self.cv1.__dict__['_ctx']      # step C + the pooled terms + BatchNorm backward: two passes over (d, y1)
            bn = (y1, scale1, shift1, mean1)
        dca, amaxp, *state = self.spatial_attention.backward(d.t, t_max=t_max, bn=bn)
        pooled = self.channel_attention.backward(dca, d, amaxp, defer=True)   # the pooled paths join d inside cv1's BatchNorm backward
        c1 = self.cv1.conv.in_channels
        fuse = self.add and pad4(c1) == c1 and dout.coff % 4 == 0  # the shortcut's gradient rides the dgrad epilogue
        self.cv1.backward(d, dx_out=dx_out, accumulate=True, also_add=dout if fuse else None, pooled=pooled, cbam=state[0] if state else None)
        if self.add and not fuse:
            ops.add_(dx_out.t, dx_out.coff, dout.t, dout.coff, x.c)
        return dx_out


class C2fCBAM(nn.Module):
    """models/common.py:2671-2695; all (2+n) pieces live in one buffer, nothing is concatenated."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5, kernel_size=7):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(
            CBAMBottleneck(self.c, self.c, shortcut, g, k=(3, 3), e=1.0, ratio=16, kernel_size=kernel_size)
            for _ in range(n))

    def forward(self, x, pool=None):
        """pool: a dict the closing conv's BatchNorm + SiLU pass fills with the output's global average / max (Conv._forward_train)."""
        c, n = self.c, len(self.m)
        if c % 4:
            raise NotImplementedError('C2fCBAM hidden width must be a multiple of 4 on the MI355X path')
        B, H, W, _ = x.shape
        cat = concat_act(x.t, H, W, (2 + n) * c)
        self.cv1(x, out=cat.slice(0, 2 * c))
        for i, blk in enumerate(self.m):
            blk(cat.slice((1 + i) * c, c), out=cat.slice((2 + i) * c, c))
        return self.cv2(cat, pool=pool)

    def backward(self, dout, dx_out=None, accumulate=False, pooled=None):
        """pooled: a pending part of dout that is constant over each image's pixels (Act.pooled): the closing conv's BatchNorm backward folds it in."""
        c, n = self.c, len(self.m)
        dcat = self.cv2.backward(dout, pooled=pooled)             # gradient of every piece through the 1x1 mix
        for i in reversed(range(n)):
            self.m[i].backward(dcat.slice((2 + i) * c, c), dcat.slice((1 + i) * c, c))
        return self.cv1.backward(dcat.slice(0, 2 * c), dx_out=dx_out, accumulate=accumulate)


class SPPF(nn.Module):
    """models/common.py:1846-1861: the three chained max-pools are one kernel writing the concat slices."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        if k != 5:
            raise NotImplementedError('SPPF kernel size 5 only')
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)

    def forward(self, x):
        c_ = self.cv1.conv.out_channels
        B, H, W, _ = x.shape
        cat = concat_act(x.t, H, W, 4 * c_)
        self.cv1(x, out=cat.slice(0, c_))
        if self.training:                                         # level by level, leaving each window's arg-max for the backward pass
            self.__dict__['_ctx'] = (cat, ops.sppf_pool_(cat.t, c_, 0, codes=True)[1])
        else:
            ops.sppf_pool_(cat.t, c_, 0)
        return self.cv2(cat)

    def backward(self, dout, dx_out=None, accumulate=False):
        cat, codes = self.__dict__.pop('_ctx')
        c_ = self.cv1.conv.out_channels
        dcat = self.cv2.backward(dout)
        ops.sppf_pool_backward_(cat.t, dcat.t, c_, 0, codes=codes)
        return self.cv1.backward(dcat.slice(0, c_), dx_out=dx_out, accumulate=accumulate)


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class BiFPN(_Packed):
    """w_i / (sum_j swish(w_j) + 1e-4) weighted sum (models/common.py:3688-3704); inputs may be virtual 2x-upsampled.
    The weights are normalised inside the kernels from the raw parameter on the device."""

    def __init__(self, length):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(length, dtype=torch.float32), requires_grad=True)
        self.swish = Swish()
        self.epsilon = 0.0001

    def forward(self, xs):
        for a in xs:
            if a.coff != 0 or a.t.shape[3] != xs[0].t.shape[3]:
                raise NotImplementedError('BiFPN inputs must be whole tensors of equal width')
        w = self.weight.detach()
        if not w.is_cuda:
            raise RuntimeError('somi_amd BiFPN runs on the MI355X only (no CPU fallback)')
        out = ops.bifpn([a.t for a in xs], [a.up for a in xs], w, self.epsilon)      # normalised inside the kernel
        if self.training:
            self.__dict__['_ctx'] = xs
        return Act(out, 0, xs[0].c)

    def backward(self, dout):
        """Returns the list of input gradients (low-resolution for the virtually upsampled inputs)."""
        xs = self.__dict__.pop('_ctx')
        dw, scratch = _grad_target(self.weight)
        ds = ops.bifpn_backward([a.t for a in xs], [a.up for a in xs], self.weight.detach(), dout.t, dw, self.epsilon)
        if scratch:
            _acc_grad(self.weight, dw)
        return [Act(d, 0, a.c) for d, a in zip(ds, xs)]


class Upsample(nn.Module):
    """nn.Upsample(None, 2, 'nearest') (YOLO-SOMI.yaml:37,42,47): a view flag; the consumer (BiFPN) reads at (h>>1, w>>1)."""

    def __init__(self, size=None, scale_factor=None, mode='nearest'):
        super().__init__()
        if size is not None or scale_factor != 2 or mode != 'nearest':
            raise NotImplementedError('only 2x nearest upsampling is on the SOMI path')

    def forward(self, x):
        return Act(x.t, x.coff, x.c, up=x.up + 1)

    def backward(self, dout):
        return dout                                              # the consumer (BiFPN) already produced the low-resolution gradient


class ODConv2d_3rd(_Packed):
    """Parameter layout of models/common.py:4495-4536; executed by ODConv_3rd below."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=True,
                 K=4, r=1 / 16):
        super().__init__()
        if groups != 1 or dilation != 1:
            raise NotImplementedError('grouped / dilated ODConv is not on the SOMI path')
        self.in_channels, self.out_channels, self.K = in_channels, out_channels, K
        self.kernel_size, self.stride, self.padding = (kernel_size, kernel_size), stride, padding
        self.weight = nn.Parameter(torch.empty(K, out_channels, in_channels, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.zeros(K, out_channels)) if bias else None
        hidden = max(int(in_channels * r), 16)
        self.reduction = nn.Linear(in_channels, hidden)          # unused in the reference too (:4521)
        self.fc = nn.Conv2d(in_channels, hidden, 1, bias=False)
        self.bn = nn.BatchNorm2d(hidden)
        self.fc_f = nn.Linear(hidden, out_channels)
        self.fc_s = nn.Linear(hidden, kernel_size * kernel_size)
        self.fc_c = nn.Linear(hidden, in_channels)
        self.fc_w = nn.Linear(hidden, K)
        fan_out = kernel_size * kernel_size * out_channels
        with torch.no_grad():
            for i in range(K):
                self.weight[i].normal_(0, math.sqrt(2.0 / fan_out))


class ODConv_3rd(_Packed):
    """ODConv2d_3rd -> BN -> SiLU (models/common.py:4638-4653): GAP kernel -> attention + per-sample weight synthesis
    (with the outer BN folded in) -> per-sample implicit-GEMM conv with SiLU epilogue."""

    def __init__(self, c1, c2, k=1, s=1, kerNums=4, g=1, p=None, act=True):
        super().__init__()
        self.conv = ODConv2d_3rd(c1, c2, k, s, autopad(k, p), groups=g, K=kerNums)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.SiLU() if act is True else (act if isinstance(act, nn.Module) else nn.Identity())

    # ------------------------------------------------------------------------------------------ training mode
    def _forward_train(self, x):
        cv = self.conv
        dev = x.t.device
        f = lambda t: t.detach().float().contiguous()            # noqa: E731
        B, H, W, _ = x.shape
        k, s, p = cv.kernel_size[0], cv.stride, cv.padding
        cin, cout, kk, K = cv.in_channels, cv.out_channels, k * k, cv.K
        if cin % 4 or cout % 4 or x.coff != 0 or x.t.shape[3] != cin:
            raise NotImplementedError('ODConv training path needs whole tensors with channels % 4 == 0')
        hid = cv.fc.weight.shape[0]
        fcw = f(cv.fc.weight).flatten(1)
        gap = x.pool[0] if x.pool is not None else ops.global_pool(x.t, want_max=False)[0]    # the producer's BatchNorm + SiLU pass may have taken it
        zpre = ops.linear(gap, fcw, None, 'none')
        one, zero = torch.ones(hid, device=dev), torch.zeros(hid, device=dev)
        if B > 1:                                                 # BatchNorm over the B samples (models/common.py:4562-4563)
            rm, rv = cv.bn.running_mean.detach().clone(), cv.bn.running_var.detach().clone()
            st = ops.bn_stats(zpre.view(B, 1, 1, hid), hid, 0, f(cv.bn.weight), f(cv.bn.bias), cv.bn.eps, cv.bn.momentum, rm, rv)
            with torch.no_grad():
                cv.bn.running_mean.copy_(rm)
                cv.bn.running_var.copy_(rv)
                _bump_batches_tracked(cv.bn)
        else:
            st = (zero, one, one, zero)
        z = ops.chan_affine_act(zpre.view(B, 1, 1, hid), hid, 0, st[2], st[3], 'relu', 0, torch.empty(B, 1, 1, hid, device=dev)).view(B, hid)
        na = cout + kk + cin + K
        attn = torch.empty(B, na, device=dev)
        heads = ((cv.fc_f, 0, 'sigmoid'), (cv.fc_s, cout, 'sigmoid'), (cv.fc_c, cout + kk, 'sigmoid'), (cv.fc_w, cout + kk + cin, 'softmax'))
        for lin, off, act in heads:
            ops.linear(z, f(lin.weight), f(lin.bias), act, attn, off)
        Wk = pack_conv_weight(cv.weight.detach().float()).to(dev)            # [K][Cout][kk*Cin]
        biask = f(cv.bias) if cv.bias is not None else None
        wout, bout = ops.odconv_synth(attn, Wk, biask, cin, cin, cout, kk, K)
        Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
        y = torch.empty(B, Ho, Wo, cout, device=dev)
        ops.conv2d_nhwc(x.t, wout, bout, kh=k, kw=k, stride=s, pad=p, act='none', out=y, cout=cout, per_sample_w=True)
        rm, rv = self.bn.running_mean.detach().clone(), self.bn.running_var.detach().clone()
        so = ops.bn_stats(y, cout, 0, f(self.bn.weight), f(self.bn.bias), self.bn.eps, self.bn.momentum, rm, rv)
        with torch.no_grad():
            self.bn.running_mean.copy_(rm)
            self.bn.running_var.copy_(rv)
            _bump_batches_tracked(self.bn)
        out = torch.empty_like(y)
        ops.chan_affine_act(y, cout, 0, so[2], so[3], _act_name(self.act), 0, out)
        self.__dict__['_ctx'] = (x, gap, fcw, zpre, st, z, attn, heads, Wk, biask, wout, y, so)
        return Act(out, 0, cout)

    def backward(self, dz, need_dx=True, dx_out=None, accumulate=False, defer_pool=False):
        """dx_out / accumulate: the data gradient is written (added) into that Act (a whole tensor) by the dgrad epilogue.  defer_pool: the squeeze's
        gradient - constant over an image's pixels - is not added here by a pass of its own but handed on as the result's `pooled` part, for the
        producing Conv's BatchNorm backward to fold in."""
        x, gap, fcw, zpre, st, z, attn, heads, Wk, biask, wout, y, so = self.__dict__.pop('_ctx')
        cv = self.conv
        dev = y.device
        B, H, W, _ = x.shape
        k, s, p = cv.kernel_size[0], cv.stride, cv.padding
        cin, cout, kk, K = cv.in_channels, cv.out_channels, k * k, cv.K
        hid = fcw.shape[0]
        f = lambda t: t.detach().float().contiguous()            # noqa: E731
        (dg, db), fin = _grad_targets(self.bn.weight, self.bn.bias)      # the kernel accumulates straight into .grad when it can
        dy = ops.bn_act_backward(dz.t, dz.coff, y, 0, cout, *so, _act_name(self.act), 0, True, torch.empty_like(y), 0, dg, db)
        fin()
        # per-sample conv: bias, weight and data gradients
        dbias_b, _ = ops.global_pool(dy, want_max=False)
        dbias_b = dbias_b * float(dy.shape[1] * dy.shape[2])                  # sum over pixels = mean * HoWo
        dWb = ops.conv2d_wgrad_nhwc(x.t, dy, kh=k, kw=k, stride=s, pad=p, per_sample_w=True)
        dx = None
        if need_dx:
            wt = wout.view(B, cout, kk, cin).permute(0, 3, 2, 1).contiguous().view(B, cin, kk * cout)
            if dx_out is not None and (dx_out.coff != 0 or dx_out.t.shape[3] != cin):
                raise NotImplementedError('ODConv backward writes whole input-gradient tensors')
            dx = ops.conv2d_dgrad_nhwc(dy, wt, B=B, H=H, W=W, cin=cin, kh=k, kw=k, stride=s, pad=p, per_sample_w=True,
                                       out=None if dx_out is None else dx_out.t, accumulate=dx_out.t if (dx_out is not None and accumulate) else None)
        # synthesis backward
        dWk = torch.zeros_like(Wk)
        dbk = torch.zeros_like(biask) if biask is not None else None
        dattn = ops.odconv_synth_backward(dWb, attn, Wk, biask, dbias_b if biask is not None else None, dWk, dbk, cin, cin, cout, kk, K)
        _acc_grad(cv.weight, dWk.view(K, cout, k, k, cin).permute(0, 1, 4, 2, 3))
        if biask is not None:
            _acc_grad(cv.bias, dbk)
        # attention heads -> dz
        dzv = torch.empty(B, hid, device=dev)
        for i, (lin, off, act) in enumerate(heads):
            gW, gb = torch.zeros_like(lin.weight.data), torch.zeros_like(lin.bias.data)
            ops.linear_backward(z, f(lin.weight), dattn, attn, off, act, gW, gb, dzv, accumulate=i > 0)
            _acc_grad(lin.weight, gW)
            _acc_grad(lin.bias, gb)
        # relu + BatchNorm over the batch (or plain relu for one sample)
        if B > 1:
            (g2, b2), fin = _grad_targets(cv.bn.weight, cv.bn.bias)
        else:                                                     # one sample: no BatchNorm in the forward (models/common.py:4562), nothing for its parameters
            g2, b2, fin = torch.zeros(hid, device=dev), torch.zeros(hid, device=dev), lambda: None
        dzpre = ops.bn_act_backward(dzv.view(B, 1, 1, hid), 0, zpre.view(B, 1, 1, hid), 0, hid, *st, 'relu', 0, B > 1,
                                    torch.empty(B, 1, 1, hid, device=dev), 0, g2, b2).view(B, hid)
        fin()
        gfc = torch.zeros_like(fcw)
        dgap = torch.empty_like(gap) if need_dx else None
        ops.linear_backward(gap, fcw, dzpre, dzpre, 0, 'none', gfc, None, dgap)
        _acc_grad(cv.fc.weight, gfc.view_as(cv.fc.weight))
        if not need_dx:
            return None
        res = Act(dx, 0, cin)
        res.pooled = (dgap, None, None)
        return res if defer_pool else settle_pooled(res)

    def _pack(self, dev):
        cv = self.conv
        f = lambda t: t.detach().float().contiguous().to(dev)   # noqa: E731
        s_in, t_in = bn_fold(cv.bn)
        fcw = cv.fc.weight.detach().float().flatten(1)
        pk = dict(fc_w=f(fcw), fc_w_bn=f(fcw * s_in[:, None]), fc_b_bn=f(t_in),
                  Wf=f(cv.fc_f.weight), bf=f(cv.fc_f.bias), Ws=f(cv.fc_s.weight), bs=f(cv.fc_s.bias),
                  Wc=f(cv.fc_c.weight), bc=f(cv.fc_c.bias), Ww=f(cv.fc_w.weight), bw=f(cv.fc_w.bias),
                  Wk=pack_conv_weight(cv.weight.detach().float()).to(dev),           # [K][Cout][kk*Cin_pad]
                  biask=None if cv.bias is None else f(cv.bias))
        s_out, t_out = bn_fold(self.bn)
        pk['bn_s'], pk['bn_t'] = f(s_out), f(t_out)
        return pk

    def forward(self, x):
        if self.training:
            return self._forward_train(x)
        pk = self._packed(x.t.device)
        cv = self.conv
        B, H, W, _ = x.shape
        k, s, p = cv.kernel_size[0], cv.stride, cv.padding
        cin, cin_pad, cout = cv.in_channels, pad4(cv.in_channels), cv.out_channels
        if x.c != cin:
            raise ValueError(f'Expected input{[B, x.c, H, W]} to have {cin} channels, but got {x.c} channels instead')
        if cout % 4 or cin % 4:
            raise NotImplementedError('ODConv channels must be multiples of 4 on the MI355X path')
        gap, _ = ops.global_pool(x.t, c=cin, x_coff=x.coff, want_max=False)
        bn_attn = B > 1                                        # the reference skips the squeeze BN for one sample (:4562)
        wout = torch.empty(B, cout, k * k * cin_pad, device=x.t.device, dtype=torch.float32)
        bout = torch.empty(B, cout, device=x.t.device, dtype=torch.float32)
        ops.odconv_weights(gap, pk['fc_w_bn'] if bn_attn else pk['fc_w'], pk['fc_b_bn'] if bn_attn else None, pk, wout,
                           bout, cin, cin_pad, cout, k * k, cv.K)
        Ho, Wo = ops.conv_out_size(H, k, s, p), ops.conv_out_size(W, k, s, p)
        out = new_act(x.t, Ho, Wo, cout)
        ops.conv2d_nhwc(x.t, wout, bout, kh=k, kw=k, stride=s, pad=p, act=_act_name(self.act), cin=cin_pad,
                        x_coff=x.coff, out=out.t, cout=cout, per_sample_w=True)
        return out


class Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn


class SEAM(_Packed):
    """models/common.py:8448-8505: dw3x3+GELU+BN -> Residual(dw3x3+GELU+BN) -> 1x1+GELU+BN -> GAP -> MLP -> x*exp(.)."""

    def __init__(self, c1, c2, n, reduction=16):
        super().__init__()
        if c1 != c2:
            c2 = c1
        if n != 1:
            raise NotImplementedError('SEAM with n != 1 is not on the SOMI path')
        stage = nn.Sequential(
            Residual(nn.Sequential(nn.Conv2d(c2, c2, 3, 1, 1, groups=c2), nn.GELU(), nn.BatchNorm2d(c2))),
            nn.Conv2d(c2, c2, 1, 1, 0, groups=1), nn.GELU(), nn.BatchNorm2d(c2))
        self.DCovN = nn.Sequential(nn.Conv2d(c1, c2, 3, 1, 1, groups=c1), nn.GELU(), nn.BatchNorm2d(c2), stage)
        self.fc = nn.Sequential(nn.Linear(c2, c2 // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(c2 // reduction, c2, bias=False), nn.Sigmoid())
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=1)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def _pack(self, dev):
        f = lambda t: t.detach().float().contiguous().to(dev)   # noqa: E731
        dw = lambda conv: f(conv.weight[:, 0].permute(1, 2, 0).reshape(9, -1))   # noqa: E731  [9][C]
        d = self.DCovN
        if self.training:
            st = d[3]
            return dict(dw0=dw(d[0]), b0=f(d[0].bias), dw1=dw(st[0].fn[0]), b1=f(st[0].fn[0].bias), pw=f(st[1].weight.flatten(1)),
                        pb=f(st[1].bias), pwt=pack_dgrad_weight(st[1].weight.detach().float()).to(dev), W1=f(self.fc[0].weight),
                        W2=f(self.fc[2].weight))
        st = d[3]
        pw = st[1]
        return dict(dw0=dw(d[0]), b0=f(d[0].bias), bn0=tuple(map(f, bn_fold(d[2]))),
                    dw1=dw(st[0].fn[0]), b1=f(st[0].fn[0].bias), bn1=tuple(map(f, bn_fold(st[0].fn[2]))),
                    pw=f(pw.weight.flatten(1)), pb=f(pw.bias), bn2=tuple(map(f, bn_fold(st[3]))),
                    W1=f(self.fc[0].weight), W2=f(self.fc[2].weight))

    # ------------------------------------------------------------------------------------------ training mode
    def _bn_train(self, u, bn, residual=None):
        """z = BN_batch(GELU(u)) [+ residual] (act before the norm, models/common.py:8455-8457; the Residual wrapper's add :7183 rides the same
        pass); returns z and the saved statistics."""
        dev, c = u.device, u.shape[3]
        rm, rv = bn.running_mean.detach().clone(), bn.running_var.detach().clone()
        if ops.SYNC_BN is None:
            # two passes over u: the statistics of gelu(u) taken on the fly, then z = gelu(u) * scale + shift (order 1) - gelu(u) is never stored
            # (the backward reads u too); three passes and a tensor less than act -> statistics -> affine
            st = ops.bn_stats(u, c, 0, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, rm, rv, act='gelu')
            z = ops.chan_affine_act(u, c, 0, st[2], st[3], 'gelu', 1, torch.empty_like(u), residual=residual)
        else:
            g = ops.chan_affine_act(u, c, 0, torch.ones(c, device=dev), torch.zeros(c, device=dev), 'gelu', 0, torch.empty_like(u))
            st = ops.bn_stats(g, c, 0, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, rm, rv)
            z = ops.chan_affine_act(g, c, 0, st[2], st[3], 'none', 0, g, residual=residual)
        with torch.no_grad():
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            _bump_batches_tracked(bn)
        return z, tuple(st)

    def _bn_backward(self, dz, u, st, bn, pooled=None):
        """gradient w.r.t. u of BN_batch(GELU(u)) (order 1), parameter gradients accumulated.  pooled: ops.bn_act_backward (dz may then be None)."""
        c = u.shape[3]
        (dg, db), fin = _grad_targets(bn.weight, bn.bias)
        du = ops.bn_act_backward(dz, 0, u, 0, c, *st, 'gelu', 1, True, torch.empty_like(u), 0, dg, db, pooled=pooled)
        fin()
        return du

    def _forward_train(self, x):
        self.invalidate()
        pk = self._packed(x.t.device)
        d, st = self.DCovN, self.DCovN[3]
        u0 = ops.dwconv3x3(x.t, pk['dw0'], pk['b0'])
        y0, s0 = self._bn_train(u0, d[2])
        u1 = ops.dwconv3x3(y0, pk['dw1'], pk['b1'])
        y1, s1 = self._bn_train(u1, st[0].fn[2], residual=y0)    # Residual: fn(y0) + y0
        u2 = ops.conv2d_nhwc(y1, pk['pw'], pk['pb'], kh=1, kw=1)
        if ops.SYNC_BN is None and ops.FUSE_POOL:
            # BN(GELU(u2)) is read by the global average pool only, and the mean of an affine map is the affine map of the mean: the statistics pass,
            # then ONE pass that averages gelu(u2) per image and applies scale / shift to the (B,C) result - the normalised tensor is never written
            bn2 = st[3]
            rm, rv = bn2.running_mean.detach().clone(), bn2.running_var.detach().clone()
            s2 = tuple(ops.bn_stats(u2, c := u2.shape[3], 0, bn2.weight.detach(), bn2.bias.detach(), bn2.eps, bn2.momentum, rm, rv, act='gelu'))
            with torch.no_grad():
                bn2.running_mean.copy_(rm)
                bn2.running_var.copy_(rv)
                _bump_batches_tracked(bn2)
            avg = ops.global_pool_act(u2, 'gelu', s2[2], s2[3], c=c)
        else:
            y2, s2 = self._bn_train(u2, st[3])
            avg, _ = ops.global_pool(y2, want_max=False)
        sc = ops.attn_mlp(1, avg, None, pk['W1'], None, pk['W2'], None)
        self.__dict__['_ctx'] = (x, u0, s0, y0, u1, s1, y1, u2, s2, avg, sc, pk)
        return Act(ops.scale_channels(x.t, sc), 0, x.c)

    def backward(self, dout):
        x, u0, s0, y0, u1, s1, y1, u2, s2, avg, sc, pk = self.__dict__.pop('_ctx')
        d, st = self.DCovN, self.DCovN[3]
        c = x.c
        dev = x.t.device
        dx, dsc = ops.scale_channels_backward(dout.t, x.t, sc)
        (gW1, gW2), fin = _grad_targets(self.fc[0].weight, self.fc[2].weight)
        davg, _ = ops.attn_mlp_backward(1, dsc, sc, avg, None, pk['W1'], None, pk['W2'], gW1, None, gW2, None)
        fin()
        # y2 is read by the global average pool only: its gradient is davg / HW at every pixel - handed to the BatchNorm backward as such
        # (no zero tensor filled, added to and read twice)
        du2 = self._bn_backward(None, u2, s2, st[3], pooled=(davg, None, None))
        dwp = ops.conv2d_wgrad_nhwc(y1, du2, kh=1, kw=1)
        _acc_grad(st[1].weight, dwp.view(c, c, 1, 1))
        (dbp,), fin = _grad_targets(st[1].bias)
        ops.chan_sum_(du2, c, 0, dbp)
        fin()
        dy1 = ops.conv2d_dgrad_nhwc(du2, pk['pwt'], B=x.shape[0], H=x.shape[1], W=x.shape[2], cin=c, kh=1, kw=1)
        du1 = self._bn_backward(dy1, u1, s1, st[0].fn[2])
        gw, gb = torch.zeros_like(pk['dw1']), torch.zeros(c, device=dev)
        dy0 = ops.dwconv3x3_backward(du1, y0, pk['dw1'], gw, gb, dx_accumulate=dy1)       # + the residual branch
        _acc_grad(st[0].fn[0].weight, gw.view(3, 3, c).permute(2, 0, 1).unsqueeze(1))
        _acc_grad(st[0].fn[0].bias, gb)
        du0 = self._bn_backward(dy0, u0, s0, d[2])
        gw0, gb0 = torch.zeros_like(pk['dw0']), torch.zeros(c, device=dev)
        dx = ops.dwconv3x3_backward(du0, x.t, pk['dw0'], gw0, gb0, dx_accumulate=dx)
        _acc_grad(d[0].weight, gw0.view(3, 3, c).permute(2, 0, 1).unsqueeze(1))
        _acc_grad(d[0].bias, gb0)
        return Act(dx, 0, c)

    def forward(self, x):
        if x.coff != 0 or x.t.shape[3] != x.c or x.c % 4:
            raise NotImplementedError('SEAM input must be a whole tensor with channels a multiple of 4')
        if self.training:
            return self._forward_train(x)
        pk = self._packed(x.t.device)
        y0 = ops.dwconv3x3(x.t, pk['dw0'], pk['b0'], *pk['bn0'], act='gelu')
        y1 = ops.dwconv3x3(y0, pk['dw1'], pk['b1'], *pk['bn1'], residual=y0, act='gelu')
        y2 = ops.conv2d_nhwc(y1, pk['pw'], pk['pb'], kh=1, kw=1, act='gelu', post_scale=pk['bn2'][0],
                             post_shift=pk['bn2'][1])
        avg, _ = ops.global_pool(y2, want_max=False)
        s = ops.attn_mlp(1, avg, None, pk['W1'], None, pk['W2'], None)
        return Act(ops.scale_channels(x.t, s), 0, x.c)


class Decouple(nn.Module):
    """Decoupled head for one level (models/yolo.py:1042-1073); the [5|nc] interleave happens in the decode kernel."""

    def __init__(self, c1, nc=80, na=3):
        super().__init__()
        c_ = min(c1, 256)
        self.na, self.nc = na, nc
        self.a = Conv(c1, c_, 1)
        c = [int(v + na * 5) for v in (c_ - na * 5) * torch.linspace(1, 0, 4)]
        self.b1, self.b2, self.b3 = Conv(c_, c[1], 3), Conv(c[1], c[2], 3), nn.Conv2d(c[2], na * 5, 1)
        self.c1, self.c2, self.c3 = Conv(c_, c_, 1), Conv(c_, c_, 1), nn.Conv2d(c_, na * nc, 1)
        self.__dict__['_b3'] = PlainConv(self.b3)              # runners share the parameters, stay out of state_dict
        self.__dict__['_c3'] = PlainConv(self.c3)

    def invalidate(self):
        self._b3.invalidate()
        self._c3.invalidate()

    def forward(self, x):
        if self.training:
            self._b3.train(), self._c3.train()
        else:
            self._b3.eval(), self._c3.eval()
        x = self.a(x)
        b = self._b3(self.b2(self.b1(x)))
        c = self._c3(self.c2(self.c1(x)))
        return b, c

    def backward(self, dbox, dcls):
        d = self.c1.backward(self.c2.backward(self._c3.backward(dcls)))
        d = self.b1.backward(self.b2.backward(self._b3.backward(dbox)), dx_out=d, accumulate=True)
        return self.a.backward(d)


class DecoupledDetect(nn.Module):
    """models/yolo.py:925-980.  forward returns (z, [raw_i]) in eval like the reference; raw_i is (B,na,ny,nx,no)."""
    stride = None

    def __init__(self, nc=10, anchors=(), ch=(), inplace=False):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(Decouple(c, self.nc, self.na) for c in ch)
        self.inplace = False

    def invalidate(self):
        self.__dict__.pop('_anchors_host', None)

    def forward(self, xs):
        B = xs[0].shape[0]
        dev = xs[0].t.device
        total = sum(self.na * a.shape[1] * a.shape[2] for a in xs)
        z = None if self.training else torch.empty(B, total, self.no, device=dev, dtype=torch.float32)
        raws, row = [], 0
        anchors = self.__dict__.get('_anchors_host')
        if anchors is None:                                   # cached host copy: no device sync per forward
            anchors = self.__dict__['_anchors_host'] = self.anchors.detach().float().cpu()
        for i in range(self.nl):
            b, c = self.m[i](xs[i])
            _, ny, nx, _ = b.shape
            raw = torch.empty(B, self.na, ny, nx, self.no, device=dev, dtype=torch.float32)
            stride = float(self.stride[i])
            ops.detect_decode(b.t, c.t, (anchors[i] * stride).flatten().tolist(), stride, self.na, self.nc, raw=raw,
                              z=z, total=total, row_off=row)
            raws.append(raw)
            row += self.na * ny * nx
            if self.training:
                self.__dict__.setdefault('_ctx', []).append((b.t.shape[3], c.t.shape[3]))
        return raws if self.training else (z, raws)

    def backward(self, draws):
        """draws: gradients w.r.t. the nl training outputs (B,na,ny,nx,no).  Returns the per-level input gradients."""
        widths = self.__dict__.pop('_ctx')
        outs = []
        for i in range(self.nl):
            dbox, dcls = ops.detect_raw_backward(draws[i].contiguous(), widths[i][0], widths[i][1], self.na, self.nc)
            outs.append(self.m[i].backward(dbox, dcls))
        return outs


# ================================================================================================ stock YOLOv5 module set
# north_star: "CSP/Darknet conv backbone, PANet/FPN neck, anchor-based detection head"; BASELINE configs[0] (yolov5s).
class Bottleneck(nn.Module):
    """cv1 -> cv2 (+x) (models/common.py:1494-1509); the shortcut rides cv2's epilogue (forward) and cv1's dgrad epilogue (backward)."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x, out=None):
        return self.cv2(self.cv1(x), out=out, residual=x if self.add else None)

    def backward(self, dout, dx_out=None, accumulate=False, need_dx=True):
        d = self.cv2.backward(dout)
        c1 = self.cv1.conv.in_channels
        fuse = self.add and pad4(c1) == c1 and dout.coff % 4 == 0
        dx = self.cv1.backward(d, dx_out=dx_out, accumulate=accumulate, need_dx=need_dx, also_add=dout if fuse else None)
        if self.add and not fuse and dx is not None:
            ops.add_(dx.t, dx.coff, dout.t, dout.coff, c1)
        return dx


class C3(nn.Module):
    """cv3(cat(m(cv1(x)), cv2(x))) (models/common.py:1541-1565).  The concatenation never materialises: the last bottleneck and cv2
    write the two halves of cv3's input buffer; in backward the halves of cv3's data gradient are read in place."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def forward(self, x):
        c_ = self.cv1.conv.out_channels
        if c_ % 4:
            raise NotImplementedError('C3 hidden width must be a multiple of 4 on the MI355X path')
        B, H, W, _ = x.shape
        cat = concat_act(x.t, H, W, 2 * c_)
        n = len(self.m)
        t = self.cv1(x, out=cat.slice(0, c_) if n == 0 else None)
        for i, blk in enumerate(self.m):
            t = blk(t, out=cat.slice(0, c_) if i == n - 1 else None)
        self.cv2(x, out=cat.slice(c_, c_))
        return self.cv3(cat)

    def backward(self, dout, dx_out=None, accumulate=False, need_dx=True):
        c_ = self.cv1.conv.out_channels
        dcat = self.cv3.backward(dout)
        dx = self.cv2.backward(dcat.slice(c_, c_), dx_out=dx_out, accumulate=accumulate, need_dx=need_dx)
        d = dcat.slice(0, c_)
        for blk in reversed(self.m):
            d = blk.backward(d)
        self.cv1.backward(d, dx_out=dx, accumulate=True, need_dx=need_dx)
        return dx


class SPP(nn.Module):
    """models/common.py:1806-1826 with the stock kernel sizes (5, 9, 13): a stride-1 max-pool of window 9 (13) is exactly two (three)
    chained 5x5 pools, so the three parallel pools are the same single kernel SPPF uses, writing the concat slices; likewise the
    gradient (every pooled value's gradient ends at the arg-max pixel of its window whichever way it is routed)."""

    def __init__(self, c1, c2, k=(5, 9, 13)):
        super().__init__()
        if tuple(k) != (5, 9, 13):
            raise NotImplementedError('SPP kernel sizes (5, 9, 13) only')
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])    # parameter-free; kept for state_dict / repr parity

    forward = SPPF.forward
    backward = SPPF.backward


class Focus(nn.Module):
    """Space-to-depth + Conv (models/common.py:1973-1997)."""

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, act=True):
        super().__init__()
        if act is not True:
            raise NotImplementedError("the reference passes `act` into Conv's dilation slot (models/common.py:1993); only act=True runs there")
        self.conv = Conv(c1 * 4, c2, k, s, p, g)

    def forward(self, x):
        if x.shape[1] % 2 or x.shape[2] % 2:
            raise RuntimeError('Focus needs even height and width')
        deep = ops.space_to_depth(x.t, x.coff, x.c)
        return self.conv(Act(deep, 0, 4 * x.c))

    def backward(self, dout, dx_out=None, accumulate=False, need_dx=True):
        d = self.conv.backward(dout, need_dx=need_dx)
        if not need_dx:
            return None
        if dx_out is not None or accumulate:
            raise NotImplementedError('Focus writes its own input gradient')
        c = self.conv.conv.in_channels // 4
        return Act(ops.space_to_depth(d.t, d.coff, c, inverse=True), 0, c)


class Concat(nn.Module):
    """torch.cat(x, 1) (models/common.py:2085-2097) of NHWC channel slices; an input that is a 2x nearest-upsampled view
    (`Upsample` only sets a flag) is expanded by the same copy.  Backward hands out slices of the incoming gradient (no copy) and
    block-sums the slice of an upsampled input."""

    def __init__(self, dimension=1):
        super().__init__()
        if dimension != 1:
            raise NotImplementedError('channel concatenation only')
        self.d = dimension

    def forward(self, xs):
        B, H, W = xs[0].shape[0], xs[0].shape[1] << xs[0].up, xs[0].shape[2] << xs[0].up
        for a in xs:
            if a.c % 4 or a.coff % 4:
                raise NotImplementedError('Concat inputs must be channel slices with offsets and widths that are multiples of 4')
            if (a.shape[0], a.shape[1] << a.up, a.shape[2] << a.up) != (B, H, W):
                raise RuntimeError(f'Sizes of tensors must match except in dimension 1. Got {[tuple(v.shape) for v in xs]}')
        out = concat_act(xs[0].t, H, W, sum(a.c for a in xs))
        off = 0
        for a in xs:
            ops.resample_slice(a.t, a.coff, out.t, off, a.c, up=a.up)
            off += a.c
        if self.training:
            self.__dict__['_ctx'] = [(a.c, a.up, a.shape) for a in xs]
        return out

    def backward(self, dout):
        ctx = self.__dict__.pop('_ctx')
        outs, off = [], 0
        for c, up, shp in ctx:
            if up == 0:
                outs.append(dout.slice(off, c))
            else:
                lo = torch.empty(shp[0], shp[1], shp[2], c, device=dout.t.device, dtype=torch.float32)
                ops.resample_slice(dout.t, dout.coff + off, lo, 0, c, up=up, reduce=True)
                outs.append(Act(lo, 0, c))
            off += c
        return outs


class Detect(nn.Module):
    """The stock anchor head (models/yolo.py:46-109): one 1x1 conv (+bias) per level, the view/permute and the eval decode in one
    kernel.  forward returns (z, [raw_i]) in eval and [raw_i] in training like the reference; raw_i is (B,na,ny,nx,no)."""
    stride = None

    def __init__(self, nc=80, anchors=(), ch=(), inplace=True):
        super().__init__()
        self.nc, self.no = nc, nc + 5
        self.nl, self.na = len(anchors), len(anchors[0]) // 2
        self.register_buffer('anchors', torch.tensor(anchors).float().view(self.nl, -1, 2))
        self.m = nn.ModuleList(nn.Conv2d(x, self.no * self.na, 1) for x in ch)
        self.inplace = inplace
        self.__dict__['_runners'] = [PlainConv(m) for m in self.m]      # share the parameters, stay out of state_dict

    def invalidate(self):
        self.__dict__.pop('_anchors_host', None)
        for r in self._runners:
            r.invalidate()

    def forward(self, xs):
        B, dev = xs[0].shape[0], xs[0].t.device
        total = sum(self.na * a.shape[1] * a.shape[2] for a in xs)
        z = None if self.training else torch.empty(B, total, self.no, device=dev, dtype=torch.float32)
        anchors = self.__dict__.get('_anchors_host')
        if anchors is None:                                   # cached host copy: no device sync per forward
            anchors = self.__dict__['_anchors_host'] = self.anchors.detach().float().cpu()
        raws, row, widths = [], 0, []
        for i, run in enumerate(self._runners):
            run.conv = self.m[i]                                 # _initialize_biases replaces the bias Parameter, the module stays
            run.train(self.training)
            t = run(xs[i])
            _, ny, nx, cs = t.shape
            raw = torch.empty(B, self.na, ny, nx, self.no, device=dev, dtype=torch.float32)
            stride = float(self.stride[i])
            ops.detect_plain_decode(t.t, (anchors[i] * stride).flatten().tolist(), stride, self.na, self.nc, raw=raw, z=z, total=total,
                                    row_off=row)
            raws.append(raw)
            widths.append(cs)
            row += self.na * ny * nx
        if self.training:
            self.__dict__['_ctx'] = widths
        return raws if self.training else (z, raws)

    def backward(self, draws):
        widths = self.__dict__.pop('_ctx')
        return [run.backward(ops.detect_plain_raw_backward(draws[i].contiguous(), widths[i], self.na, self.nc))
                for i, run in enumerate(self._runners)]


# ================================================================================================ DCNv3 wired into a graph
class DCNv3_YOLO(_Packed):
    """NHWC DCNv3 layer (models/ops_dcnv3/modules/dcnv3.py:222-379) -> BatchNorm2d -> SiLU, channel-preserving: the build's wiring
    of the reference's deformable-conv layer into the YOLO graph (the reference wires it into no model).  The
    activations are already NHWC on this path, so the NCHW<->NHWC permutes of an NCHW graph do not exist.  Eval: BN is folded into
    the layer's output projection (+SiLU in that conv's epilogue): no pass of its own.  Train: the DCNv3 module's hand-written
    backward (somi_amd.dcnv3.DCNv3._backward_impl) is chained with the BN / SiLU backward kernel."""

    def __init__(self, c, k=3, s=1, g=4, offset_scale=1.0, center_feature_scale=False):
        super().__init__()
        from .dcnv3 import DCNv3
        if s != 1:
            raise NotImplementedError('DCNv3_YOLO is wired with stride 1')
        self.dcnv3 = DCNv3(c, kernel_size=k, stride=s, pad=k // 2, group=g, offset_scale=offset_scale,
                           center_feature_scale=center_feature_scale)
        self.bn = nn.BatchNorm2d(c)
        self.act = nn.SiLU()

    def _pack(self, dev):
        """eval: output projection with the BatchNorm folded in (W' = diag(s) W, b' = s b + t)."""
        s_, t_ = bn_fold(self.bn)
        w = (self.dcnv3.output_proj.weight.detach().float() * s_[:, None]).contiguous().to(dev)
        b = (self.dcnv3.output_proj.bias.detach().float() * s_ + t_).contiguous().to(dev)
        return w, b

    def forward(self, x):
        c = self.bn.num_features
        if x.coff != 0 or x.t.shape[3] != c or x.c != c or c % 4:
            raise NotImplementedError('DCNv3_YOLO input must be a whole tensor with channels a multiple of 4')
        if not self.training:
            w, b = self._packed(x.t.device)
            return Act(self.dcnv3._forward_impl(x.t, out_proj=(w, b, _act_name(self.act))), 0, c)
        self.invalidate()
        bn, dev = self.bn, x.t.device
        rm, rv = bn.running_mean.detach().clone(), bn.running_var.detach().clone()
        sd = {'pivot': rm} if ops.FUSED_BN_STATS and x.t.numel() * 4 <= 0xE0000000 else None
        u, saved = self.dcnv3._forward_impl(x.t, keep=True, bn_stats=sd)
        if sd is not None and 'part' in sd:                       # the output projection's epilogue left the partial sums: no extra read of u
            st = ops.bn_stats_from_partials(sd['part'], sd['rows'], u.numel() // c, c, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum,
                                            rm, rv)
        else:
            st = ops.bn_stats(u, c, 0, bn.weight.detach(), bn.bias.detach(), bn.eps, bn.momentum, rm, rv)
        with torch.no_grad():
            bn.running_mean.copy_(rm)
            bn.running_var.copy_(rv)
            _bump_batches_tracked(bn)
        out = ops.chan_affine_act(u, c, 0, st[2], st[3], _act_name(self.act), 0, torch.empty_like(u))
        self.__dict__['_ctx'] = (u, st, saved)
        return Act(out, 0, c)

    def backward(self, dz, need_dx=True):
        u, st, saved = self.__dict__.pop('_ctx')
        c, dev = self.bn.num_features, u.device
        (dg, db), fin = _grad_targets(self.bn.weight, self.bn.bias)
        du = ops.bn_act_backward(dz.t, dz.coff, u, 0, c, *st, _act_name(self.act), 0, True, torch.empty_like(u), 0, dg, db)
        fin()
        dinput, grads = self.dcnv3._backward_impl(saved, du)
        for p_, g_ in zip(self.dcnv3._params(), grads):
            _acc_grad(p_, g_.view_as(p_) if g_.numel() == p_.numel() else g_[:p_.shape[0]])
        return Act(dinput, 0, c) if need_dx else None
