"""`Model`: yaml -> SOMI layer graph -> forward on the MI355X, mirroring models/yolo.py:1164-1664.

Same constructor (`Model(cfg, ch=3, nc=None, anchors=None)`), same attributes callers use (`.stride`, `.names`, `.nc`,
`.hyp`, `.yaml`, `.model[-1].{nl,na,nc,anchors,stride}`, `.fuse()`), same outputs: eval -> `(z (B,N,no), [raw_l])`,
raw_l = (B,na,ny,nx,no).  Input is the reference's `(B,3,H,W)` NCHW batch, either float32 already divided by 255
(train.py:249) or uint8 (the /255 then happens in the ingest kernel).  Accepted module names: the SOMI set of SURVEY.md
section 8a and the stock YOLOv5 set north_star names (Bottleneck, C3, SPP, Focus, Concat, Detect - BASELINE configs[0]).
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import yaml

from . import blocks as B
from . import ops


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


_CH = {'Conv': B.Conv, 'SPPF': B.SPPF, 'C2fCBAM': B.C2fCBAM, 'SEAM': B.SEAM, 'Bottleneck': B.Bottleneck, 'C3': B.C3, 'SPP': B.SPP,
       'Focus': B.Focus}                                        # models/yolo.py:1472-1479
_REPEAT_INSIDE = ('C2fCBAM', 'C3')                              # models/yolo.py:1487-1492
_ALIASES = {'C2fEACBAM': 'C2fCBAM'}     # undefined in the reference (SURVEY "five facts" #2); documented substitution


def parse_model(d, ch):
    """models/yolo.py:1453-1664 for the SOMI module set."""
    anchors, nc, gd, gw = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    consts = {'None': None, 'nc': nc, 'anchors': anchors, 'True': True, 'False': False}
    for i, (f, n, name, args) in enumerate(d['backbone'] + d['head']):
        name = _ALIASES.get(name, name)
        args = [consts.get(a, a) if isinstance(a, str) else a for a in args]
        n = max(round(n * gd), 1) if n > 1 else n
        if name in _CH:
            m = _CH[name]
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
            if name in _REPEAT_INSIDE:
                args.insert(2, n)
                n = 1
        elif name == 'ODConv_3rd':
            m = B.ODConv_3rd
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
        elif name == 'BiFPN':
            m = B.BiFPN
            args = [len(f)]
        elif name == 'nn.Upsample':
            m = B.Upsample
            c2 = ch[f]
        elif name == 'DCNv3_YOLO':                                # the reference's generic branch (models/yolo.py:1647-1648): channels pass through
            m = B.DCNv3_YOLO
            c2 = ch[f]
            args = [c2, *args[1:]]
        elif name == 'Concat':                                    # models/yolo.py:1589-1591
            m = B.Concat
            c2 = sum(ch[x] for x in f)
        elif name in ('DecoupledDetect', 'Detect'):               # models/yolo.py:1606-1610, 1616-1619
            m = B.DecoupledDetect if name == 'DecoupledDetect' else B.Detect
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        else:
            raise NotImplementedError(f'module {name!r} is outside the SOMI hot path (SURVEY.md section 8a)')
        if n > 1:
            raise NotImplementedError(f'{n} repeats of {name!r} as an nn.Sequential are not on the path (no shipped or stock graph has them)')
        m_ = nn.Sequential(*(m(*args) for _ in range(n))) if n > 1 else m(*args)
        m_.i, m_.f, m_.type = i, f, name
        m_.np = sum(p.numel() for p in m_.parameters())
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    return nn.Sequential(*layers), sorted(save)


class Model(nn.Module):
    def __init__(self, cfg='yolov5s.yaml', ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            with open(cfg, errors='ignore') as fh:
                self.yaml = yaml.safe_load(fh)
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if ch != 3:
            raise NotImplementedError('the ingest kernel packs 3-channel images')
        if nc and nc != self.yaml['nc']:
            self.yaml['nc'] = nc
        if anchors:
            self.yaml['anchors'] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml['nc'])]
        self.nc = self.yaml['nc']
        self.inplace = self.yaml.get('inplace', False)
        det = self.model[-1]
        if not isinstance(det, (B.DecoupledDetect, B.Detect)):
            raise NotImplementedError('only the DecoupledDetect and plain Detect heads are on the path')
        # the reference probes strides with a 256x256 zero image (models/yolo.py:1197-1216); the graph's strides
        # follow from its stride-2 layers, computed here without a device
        det.stride = torch.tensor(self._probe_strides())
        det.inplace = self.inplace
        if isinstance(det, B.DecoupledDetect):                   # models/yolo.py:1209-1216: order check, then scaling
            self._check_anchor_order(det)
            det.anchors /= det.stride.view(-1, 1, 1)
            self.stride = det.stride
            self._initialize_dh_biases()
        else:                                                    # models/yolo.py:1196-1207: scaling, then order check
            det.anchors /= det.stride.view(-1, 1, 1)
            self._check_anchor_order(det)
            self.stride = det.stride
            self._initialize_biases()
        for m in self.modules():                                 # utils/torch_utils.py:165-174
            if type(m) is nn.BatchNorm2d:
                m.eps, m.momentum = 1e-3, 0.03

    # ---------------------------------------------------------------------------------------------- construction
    def _probe_strides(self):
        red = []                                                 # spatial reduction factor of every layer's output
        for m in self.model:
            src = m.f if isinstance(m.f, int) else m.f[0]
            r = 1.0 if m.i == 0 else (red[m.i - 1] if src == -1 else red[src])
            if isinstance(m, B.Conv):
                r *= m.conv.stride[0]
            elif isinstance(m, B.Focus):
                r *= 2 * m.conv.conv.stride[0]
            elif isinstance(m, B.ODConv_3rd):
                r *= m.conv.stride
            elif isinstance(m, B.Upsample):
                r /= 2
            red.append(r)
        return [float(red[j]) for j in self.model[-1].f]

    @staticmethod
    def _check_anchor_order(m):
        """utils/autoanchor.py:16-22."""
        a = m.anchors.prod(-1).view(-1)
        if (a[-1] - a[0]).sign() != (m.stride[-1] - m.stride[0]).sign():
            m.anchors[:] = m.anchors.flip(0)

    def _initialize_dh_biases(self, cf=None):
        """models/yolo.py:1334-1345."""
        det = self.model[-1]
        for mi, s in zip(det.m, det.stride):
            b = mi.b3.bias.view(det.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            mi.b3.bias = nn.Parameter(b.view(-1), requires_grad=True)
            mi._b3.conv = mi.b3
            b = mi.c3.bias.data
            b += math.log(0.6 / (det.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
            mi.c3.bias = nn.Parameter(b, requires_grad=True)

    def _initialize_biases(self, cf=None):
        """models/yolo.py:1355-1366 (plain Detect)."""
        det = self.model[-1]
        for mi, s in zip(det.m, det.stride):
            b = mi.bias.view(det.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (det.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = nn.Parameter(b.view(-1), requires_grad=True)

    # ---------------------------------------------------------------------------------------------- state handling
    def invalidate(self):
        """Drop the device-side packed weights (call after changing parameters in place)."""
        for m in self.modules():
            if hasattr(m, 'invalidate') and m is not self:
                m.invalidate()

    def load_state_dict(self, *a, **k):
        r = super().load_state_dict(*a, **k)
        self.invalidate()
        return r

    def train(self, mode=True):
        self.invalidate()
        return super().train(mode)

    def fuse(self):
        """models/yolo.py:1413-1428.  Conv+BN are always folded at pack time on this path, so fuse() only has to keep
        the reference's contract (returns self, eval semantics unchanged)."""
        return self

    # ---------------------------------------------------------------------------------------------- forward
    def forward(self, x, augment=False, profile=False, visualize=False):
        if augment:
            return self._forward_augment(x)
        return self._forward_once(x)

    def _forward_augment(self, x):
        """Test-time augmentation, models/yolo.py:1253-1267: scales 1 / 0.83 / 0.67, each also flipped left-right.  The image is ingested
        once; every variant is one resample launch (flip + torch-bilinear resize + 0.447 padding to the stride multiple), the
        predictions are de-scaled / un-flipped in place (:1292-1308) and clipped (:1310-1318)."""
        if self.training:
            raise RuntimeError('augmented inference runs in eval mode')
        H, W = x.shape[-2:]
        base = ops.image_to_nhwc4(x.contiguous(), scale=1.0 / 255.0 if x.dtype == torch.uint8 else 1.0)
        gs = int(self.stride.max())
        y = []
        for si, fl in zip([1, 1, 0.83, 0.83, 0.67, 0.67], [False, True, False, True, False, True]):
            xi = base if (si == 1 and not fl) else ops.tta_resample(base, float(si), gs, fl)
            z = self._forward_once(None, ingested=xi)[0]
            if si != 1 or fl:
                ops.tta_descale_(z, si, fl, W)
            y.append(z)
        nl = self.model[-1].nl
        g = sum(4 ** v for v in range(nl))
        y[0] = y[0][:, :-(y[0].shape[1] // g)]                                   # _clip_augmented: drop the first pass's coarsest level
        y[-1] = y[-1][:, (y[-1].shape[1] // g) * 4 ** (nl - 1):]               # ... and the last pass's finest
        return torch.cat(y, 1), None

    # ---------------------------------------------------------------------------------------------- training
    def _sources(self, m):
        f = m.f if isinstance(m.f, (list, tuple)) else [m.f]
        return [m.i - 1 if j == -1 else j for j in f]

    def _backward_walk(self, draws):
        """Reverse walk of the layer graph (the autograd of models/yolo.py:1269-1290 done by hand): `draws` are the gradients
        w.r.t. the training outputs; parameter gradients are accumulated into .grad by the blocks."""
        grads = {}

        def give(src, d):
            if src < 0:
                return                                            # the input image needs no gradient
            if src not in grads:
                grads[src] = d
            else:
                g = B.settle_pooled(grads[src])                   # whole padded tensors add their pads too; slices add exactly c
                B.settle_pooled(d)
                whole = all(a.coff == 0 and a.t.shape[3] == B.pad4(a.c) for a in (g, d))
                ops.add_(g.t, g.coff, d.t, d.coff, g.t.shape[3] if whole else g.c)
        det = self.model[-1]
        hook = self.__dict__.get('_grad_hook')                   # called with a layer index once that layer's gradients are final
        for src, d in zip(self._sources(det), det.backward(draws)):
            give(src, d)
        if hook:
            hook(det.i)
        for m in reversed(list(self.model)[:-1]):
            g = grads.pop(m.i, None)
            if g is None:
                raise RuntimeError(f'layer {m.i} ({m.type}) received no gradient')
            srcs = self._sources(m)
            pend = None
            if g.pooled is not None:                              # a part constant over each image's pixels, not added yet (ODConv's squeeze gradient)
                if type(m) in (B.Conv, B.C2fCBAM):
                    pend, g.pooled = g.pooled, None               # this layer's (closing conv's) BatchNorm backward folds it in
                else:
                    B.settle_pooled(g)
            if isinstance(m, (B.BiFPN, B.Concat)):
                for src, d in zip(srcs, m.backward(g)):
                    give(src, d)
            elif srcs[0] < 0:
                m.backward(g, need_dx=False, **({'pooled': pend} if pend is not None else {}))
            elif isinstance(m, B.DCNv3_YOLO):
                give(srcs[0], m.backward(g))
            else:
                have = grads.get(srcs[0])
                kw = {'pooled': pend} if pend is not None else {}
                if (have is not None and isinstance(m, (B.Conv, B.C2fCBAM, B.C3, B.SPPF, B.SPP, B.ODConv_3rd)) and have.coff == 0 and
                        have.t.shape[3] == B.pad4(have.c) and have.t.is_contiguous() and have.pooled is None and
                        (not isinstance(m, B.ODConv_3rd) or (ops.ODCONV_INPLACE and have.t.shape[3] == have.c))):
                    # the input already holds another consumer's gradient: the data-gradient epilogue adds to it in place
                    if (isinstance(m, B.ODConv_3rd) and type(self.model[srcs[0]]) in (B.Conv, B.C2fCBAM) and ops.BN_POOLED and ops.SYNC_BN is None):
                        have.pooled = m.backward(g, dx_out=have, accumulate=True, defer_pool=True).pooled    # the last consumer to arrive: see below
                    else:
                        m.backward(g, dx_out=have, accumulate=True, **kw)
                elif (have is None and isinstance(m, B.ODConv_3rd) and type(self.model[srcs[0]]) in (B.Conv, B.C2fCBAM) and ops.BN_POOLED and
                      ops.SYNC_BN is None and ops.ODCONV_INPLACE):
                    # the only consumer of a plain Conv's output (later layers have all been walked): its squeeze gradient rides that Conv's BatchNorm backward
                    give(srcs[0], m.backward(g, defer_pool=True))
                else:
                    give(srcs[0], m.backward(g, **kw))
            if hook:
                hook(m.i)

    def _forward_once(self, x, ingested=None):
        """models/yolo.py:1269-1290: walk the layers with the skip list.  ingested: an already converted (B,H,W,4) image (TTA variants)."""
        if ingested is None:
            if not x.is_cuda:
                raise RuntimeError('somi_amd.Model runs on the MI355X only (no CPU fallback); move the batch to cuda')
            if x.dim() != 4 or x.shape[1] != 3:
                raise RuntimeError(f'expected a (B,3,H,W) batch, got {tuple(x.shape)}')
            ingested = ops.image_to_nhwc4(x.contiguous(), scale=1.0 / 255.0 if x.dtype == torch.uint8 else 1.0)
        a = B.Act(ingested, 0, 3)
        y = []
        with B.collect_batches_tracked():                         # one multi-tensor add for all BatchNorm step counters
            for m in self.model:
                if m.f != -1:
                    a = y[m.f] if isinstance(m.f, int) else [a if j == -1 else y[j] for j in m.f]
                nxt = self.model[m.i + 1] if m.i + 1 < len(self.model) else None
                if (self.training and ops.FUSE_POOL and ops.ODCONV_INPLACE and type(m) in (B.Conv, B.C2fCBAM) and isinstance(nxt, B.ODConv_3rd) and nxt.f == -1):
                    pool = {}                                     # the next layer squeezes this output: its last BatchNorm + SiLU pass takes the average
                    a = m(a, pool=pool)
                    if 'avg' in pool:
                        a.pool = (pool['avg'], pool['max'])
                else:
                    a = m(a)
                y.append(a if m.i in self.save else None)
        if self.training:
            # the blocks keep what backward needs on themselves (one set): a later train-mode forward overwrites it, so the
            # graph node remembers which forward it belongs to and refuses a stale backward instead of using wrong activations
            gen = self.__dict__['_fwd_gen'] = self.__dict__.get('_fwd_gen', 0) + 1
            return list(_ModelGraph.apply(self._anchor(ingested.device), self, gen, *a))
        return a

    def _anchor(self, dev):
        t = self.__dict__.get('_anchor_t')
        if t is None or t.device != dev:
            t = self.__dict__['_anchor_t'] = torch.zeros(1, device=dev, requires_grad=True)
        return t


class _ModelGraph(torch.autograd.Function):
    """Makes `loss.backward()` (train.py:270) drive the hand-written reverse walk: the training outputs are this node's
    outputs, its backward hands their gradients to Model._backward_walk, which fills the parameters' .grad."""

    @staticmethod
    def forward(ctx, anchor, model, gen, *raws):
        ctx.model, ctx.gen = model, gen
        return tuple(r.view_as(r) for r in raws)

    @staticmethod
    def backward(ctx, *draws):
        if ctx.model.__dict__.get('_fwd_gen') != ctx.gen:
            raise RuntimeError('backward of a stale forward: the model ran another train-mode forward since (saved activations are '
                               'kept once per block); call backward before the next forward')
        ctx.model._backward_walk([d.contiguous() for d in draws])
        return (None, None, None) + (None,) * len(draws)
