"""Fused Adam + model-EMA over flat parameter segments, mirroring train.py:125-149,271-277.

`build_optimizer(model, hyp)` reproduces the reference's three parameter groups (BatchNorm weights / other weights with
weight decay / biases, train.py:125-140) but stores each group contiguously: parameters, gradients, Adam moments and the EMA
shadow are flat fp32 buffers, the modules' `.data` / `.grad` are views into them.  One `somi_adam_ema_step_f32` launch per group
does the Adam update and the EMA update in a single pass (the reference sweeps all 77.5 M parameters twice more).
The flat gradient buffer is also what the data-parallel all-reduce works on (`ddp.GradBuckets`).
"""
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import check
from .ops import _ptr, _stream


def reference_param_groups(model):
    """train.py:125-133: g0 = BatchNorm2d weights, g1 = every other `.weight` Parameter, g2 = every `.bias` Parameter."""
    g0, g1, g2 = [], [], []
    for v in model.modules():
        if hasattr(v, 'bias') and isinstance(v.bias, nn.Parameter):
            g2.append(v.bias)
        if isinstance(v, nn.BatchNorm2d):
            g0.append(v.weight)
        elif hasattr(v, 'weight') and isinstance(v.weight, nn.Parameter):
            g1.append(v.weight)
    return g0, g1, g2


class _Flat:
    """Flat fp32 storage for a list of parameters.  `packed` maps id(param) -> (cout_pad, kh, kw, cin_pad) for the conv weights
    that are stored in the kernels' forward packing [cout_pad][kh][kw][cin_pad] (pads zero, never touched by Adam: their
    gradient is 0); the module parameter is then a strided (Cout,Cin,kh,kw) view of that storage."""

    def __init__(self, tensors, dev, packed=None):
        self.tensors = tensors
        self.packed = [(packed or {}).get(id(t)) for t in tensors]
        self.sizes = [t.numel() if pk is None else pk[0] * pk[1] * pk[2] * pk[3] for t, pk in zip(tensors, self.packed)]
        self.offsets, o = [], 0
        for n in self.sizes:                                      # every tensor starts 16 B aligned (kernels take float4)
            self.offsets.append(o)
            o += (n + 3) // 4 * 4
        self.n = self.n_pad = o
        self.data = torch.zeros(max(o, 4), device=dev, dtype=torch.float32)

    def view_of(self, i, buf=None):
        """Parameter-shaped (reference layout) view of entry i inside `buf` (default: this buffer)."""
        buf = self.data if buf is None else buf
        t, n, o, pk = self.tensors[i], self.sizes[i], self.offsets[i], self.packed[i]
        if pk is None:
            return buf[o:o + n].view_as(t)
        cout, cin = t.shape[0], t.shape[1]
        return buf[o:o + n].view(*pk)[:cout, :, :, :cin].permute(0, 3, 1, 2)

    def storage_of(self, i, buf=None):
        """The kernels' view of a packed entry: [cout_pad][kh*kw*cin_pad]."""
        buf = self.data if buf is None else buf
        pk = self.packed[i]
        return buf[self.offsets[i]:self.offsets[i] + self.sizes[i]].view(pk[0], pk[1] * pk[2] * pk[3])

    def views(self):
        for i, t in enumerate(self.tensors):
            yield t, self.view_of(i)


class FusedAdamEMA:
    """Adam with the reference's group layout and an optional fused EMA shadow.  `param_groups[i]['lr']` etc. can be edited
    between steps exactly like torch.optim (the warm-up of train.py:250-256 does that)."""

    def __init__(self, model, lr=3e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, ema_decay=0.9999, ema=True, sgd=False):
        """sgd=True: the reference's other branch (train.py:138) - SGD with Nesterov momentum betas[0]; the groups then carry a
        'momentum' key (the warm-up of train.py:255-256 ramps it) and `lr` should be hyp['lr0']."""
        self.sgd = bool(sgd)
        dev = next(model.parameters()).device
        if dev.type != 'cuda':
            raise RuntimeError('FusedAdamEMA runs on the MI355X only (no CPU fallback)')
        self.model = model
        g0, g1, g2 = reference_param_groups(model)
        seen = set()
        groups = []
        for plist, wd in ((g0, 0.0), (g1, weight_decay), (g2, 0.0)):
            plist = [p for p in plist if id(p) not in seen and not seen.add(id(p))]
            groups.append((plist, wd))
        # Conv+BN blocks keep their weight master in the kernels' forward packing: nothing to repack per step, and the
        # weight gradient is accumulated into the gradient buffer by the wgrad kernel itself
        from .blocks import Conv
        from .pack import pad4
        owners = {id(m.conv.weight): m for m in model.modules() if isinstance(m, Conv)}
        packed = {k: (pad4(m.conv.out_channels), m.conv.kernel_size[0], m.conv.kernel_size[1], pad4(m.conv.in_channels))
                  for k, m in owners.items()}
        self.param_groups, self._flat = [], []
        for plist, wd in groups:
            fp, fg = _Flat(plist, dev, packed), _Flat(plist, dev, packed)
            with torch.no_grad():
                for i, ((p, v), (_, gv)) in enumerate(zip(fp.views(), fg.views())):
                    v.copy_(p.data)
                    p.data = v                                   # the module parameter now lives inside the flat buffer
                    p.grad = gv                                  # and so does its gradient
                    if fp.packed[i] is not None:
                        owners[id(p)].__dict__['_master'] = (fp.storage_of(i), fg.storage_of(i))
            st = dict(p=fp, g=fg, m=torch.zeros_like(fp.data), v=torch.zeros_like(fp.data),
                      ema=fp.data.clone() if ema else None)
            self._flat.append(st)
            grp = dict(params=plist, lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=wd)
            if self.sgd:
                grp.update(momentum=betas[0], nesterov=True)
            self.param_groups.append(grp)
        # float buffers (BN running statistics): re-homed into one flat buffer as well, so their EMA is one launch
        self._buf_owner = []
        for mod in model.modules():
            for name, b in mod._buffers.items():
                if b is not None and b.dtype.is_floating_point and b.numel() > 0 and b.device == dev:
                    self._buf_owner.append((mod, name))
        bufs = [getattr(mod, name) for mod, name in self._buf_owner]
        self._bflat = _Flat(bufs, dev)
        with torch.no_grad():
            for i, (mod, name) in enumerate(self._buf_owner):
                v = self._bflat.view_of(i)
                v.copy_(bufs[i])
                setattr(mod, name, v)                             # still a registered buffer, now a view of the flat storage
        self._bflat.tensors = [getattr(mod, name) for mod, name in self._buf_owner]
        self._buf_ema = self._bflat.data.clone() if ema else None
        self.ema_decay, self.updates, self.steps = ema_decay, 0, 0
        if hasattr(model, 'invalidate'):
            model.invalidate()

    @property
    def flat_params(self):
        """The flat parameter buffers (one per group)."""
        return [st['p'].data for st in self._flat]

    @property
    def flat_buffers(self):
        """The flat buffer of the model's float buffers (BatchNorm running statistics)."""
        return self._bflat.data[:self._bflat.n_pad]

    def reset_ema(self):
        """Restart the EMA shadow from the current weights (after the initial weights were broadcast)."""
        for st in self._flat:
            if st['ema'] is not None:
                st['ema'].copy_(st['p'].data)
        if self._buf_ema is not None:
            self._buf_ema.copy_(self._bflat.data)

    @property
    def flat_grads(self):
        """The flat gradient buffers (one per group) - what the data-parallel all-reduce operates on."""
        return [st['g'].data for st in self._flat]

    def zero_grad(self, set_to_none=False):
        for st in self._flat:
            st['g'].data.zero_()

    def step(self):
        self.steps += 1
        d = None
        if self._flat[0]['ema'] is not None:
            self.updates += 1
            d = self.ema_decay * (1 - math.exp(-self.updates / 2000))          # utils/torch_utils.py:331
        L = _lib.lib()
        for grp, st in zip(self.param_groups, self._flat):
            if st['p'].n == 0:
                continue
            if self.sgd:
                check(L.somi_sgd_ema_step_f32(_ptr(st['p'].data), _ptr(st['g'].data), _ptr(st['m']), _ptr(st['ema']), st['p'].n_pad, float(grp['lr']),
                                              float(grp['momentum']), float(grp['weight_decay']), self.steps, float(d if d is not None else 0.0),
                                              _stream()), 'sgd_ema_step')
                continue
            check(L.somi_adam_ema_step_f32(_ptr(st['p'].data), _ptr(st['g'].data), _ptr(st['m']), _ptr(st['v']), _ptr(st['ema']), st['p'].n_pad,
                                           float(grp['lr']), float(grp['betas'][0]), float(grp['betas'][1]), float(grp['eps']),
                                           float(grp['weight_decay']), self.steps, float(d if d is not None else 0.0), _stream()), 'adam_ema_step')
        if d is not None and self._buf_ema is not None and self._bflat.n > 0:
            check(L.somi_axpby_f32(_ptr(self._buf_ema), _ptr(self._bflat.data), self._bflat.n_pad, float(d), float(1 - d), _stream()),
                  'ema buffers')
        if hasattr(self.model, 'invalidate'):
            self.model.invalidate()

    def ema_state_dict(self):
        """The EMA weights in state_dict form (what the reference's `ema.ema` module would hold)."""
        sd = {k: v.detach().clone() for k, v in self.model.state_dict().items()}
        name_of = {id(p): n for n, p in self.model.named_parameters()}
        for st in self._flat:
            if st['ema'] is None:
                continue
            for i, p in enumerate(st['p'].tensors):
                sd[name_of[id(p)]] = st['p'].view_of(i, st['ema']).clone()
        if self._buf_ema is not None:
            mname = {id(m): n for n, m in self.model.named_modules()}
            for i, (mod, name) in enumerate(self._buf_owner):
                prefix = mname[id(mod)]
                sd[(prefix + '.' if prefix else '') + name] = self._bflat.view_of(i, self._buf_ema).clone()
        return sd


def build_optimizer(model, hyp, batch_size, nbs=64, ema=True, adam=True):
    """train.py:121-140: weight decay scaled by batch_size*accumulate/nbs; Adam(lr=3e-4, betas=(momentum, 0.999)) - what the reference
    always runs (`opt.adam = True`, :134) - or, adam=False, SGD(lr=hyp['lr0'], momentum, nesterov=True) (:138)."""
    accumulate = max(round(nbs / batch_size), 1)
    wd = hyp['weight_decay'] * batch_size * accumulate / nbs
    if not adam:
        return FusedAdamEMA(model, lr=hyp['lr0'], betas=(hyp['momentum'], 0.999), weight_decay=wd, ema=ema, sgd=True)
    return FusedAdamEMA(model, lr=3e-4, betas=(hyp['momentum'], 0.999), weight_decay=wd, ema=ema)
