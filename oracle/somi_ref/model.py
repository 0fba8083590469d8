"""Oracle (test infrastructure): yaml -> layer list -> forward, restating models/yolo.py.

Accepted module names: the SOMI set of SURVEY.md section 8a plus the stock YOLOv5 set north_star names (Bottleneck, C3, SPP,
Focus, Concat, Detect - BASELINE configs[0]); anything else raises.
"""
import math
from copy import deepcopy

import torch
import torch.nn as nn
import yaml

from . import blocks as B

# module name in yaml -> class; names that take (c1, c2, ...) with width scaling
_CH_MODULES = {'Conv': B.Conv, 'SPPF': B.SPPF, 'C2fCBAM': B.C2fCBAM, 'SEAM': B.SEAM, 'Bottleneck': B.Bottleneck, 'C3': B.C3, 'SPP': B.SPP,
               'Focus': B.Focus}                                      # models/yolo.py:1472-1479
_REPEAT_INSIDE = {'C2fCBAM', 'C3'}                                    # :1487-1492
# SURVEY "five facts" #2: C2fEACBAM is undefined in the reference; the documented substitution
_ALIASES = {'C2fEACBAM': 'C2fCBAM'}


def parse_model(d, ch):
    """Build the layer list from a model dict (models/yolo.py:1453-1664, SOMI branches only).

    Returns (nn.Sequential, sorted save-list).  Each layer gets .i (index), .f (from), .type, .np.
    """
    anchors, nc, gd, gw = d['anchors'], d['nc'], d['depth_multiple'], d['width_multiple']
    na = (len(anchors[0]) // 2) if isinstance(anchors, list) else anchors
    no = na * (nc + 5)
    layers, save, c2 = [], [], ch[-1]
    for i, (f, n, name, args) in enumerate(d['backbone'] + d['head']):
        name = _ALIASES.get(name, name)
        args = [({'None': None, 'nc': nc, 'anchors': anchors, 'True': True, 'False': False}.get(a, a)
                 if isinstance(a, str) else a) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n                      # :1471
        if name in _CH_MODULES:
            m = _CH_MODULES[name]
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = B.make_divisible(c2 * gw, 8)                     # :1482-1484
            args = [c1, c2, *args[1:]]
            if name in _REPEAT_INSIDE:
                args.insert(2, n)                                      # :1491-1492
                n = 1
        elif name == 'ODConv_3rd':                                    # :1516-1521
            m = B.ODConv_3rd
            c1, c2 = ch[f], args[0]
            if c2 != no:
                c2 = B.make_divisible(c2 * gw, 8)
            args = [c1, c2, *args[1:]]
        elif name == 'BiFPN':                                         # :1547-1549 (c2 keeps its last value)
            m = B.BiFPN
            args = [len(f)]
        elif name == 'nn.Upsample':
            m = nn.Upsample
            c2 = ch[f]
        elif name == 'DCNv3_YOLO':                                    # the reference's generic branch (:1647-1648): channels pass through
            m = B.DCNv3_YOLO
            c2 = ch[f]
            args = [c2, *args[1:]]
        elif name == 'Concat':                                        # :1589-1591
            m = B.Concat
            c2 = sum(ch[x] for x in f)
        elif name in ('DecoupledDetect', 'Detect'):                   # :1606-1610, :1616-1619
            m = B.DecoupledDetect if name == 'DecoupledDetect' else B.Detect
            args.append([ch[x] for x in f])
            if isinstance(args[1], int):
                args[1] = [list(range(args[1] * 2))] * len(f)
        else:
            raise NotImplementedError(f'module {name!r} is outside the SOMI hot path (SURVEY.md section 8a)')
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
    """models/yolo.py:1164-1450 restricted to the DecoupledDetect and plain Detect heads."""

    def __init__(self, cfg, ch=3, nc=None, anchors=None):
        super().__init__()
        if isinstance(cfg, dict):
            self.yaml = deepcopy(cfg)
        else:
            with open(cfg, errors='ignore') as fh:
                self.yaml = yaml.safe_load(fh)
        ch = self.yaml['ch'] = self.yaml.get('ch', ch)
        if nc and nc != self.yaml['nc']:
            self.yaml['nc'] = nc
        if anchors:
            self.yaml['anchors'] = round(anchors)
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=[ch])
        self.names = [str(i) for i in range(self.yaml['nc'])]
        self.inplace = self.yaml.get('inplace', False)
        det = self.model[-1]
        if isinstance(det, B.DecoupledDetect):                        # :1209-1216
            s = 256
            det.inplace = self.inplace
            det.stride = torch.tensor([s / x.shape[-2] for x in self.forward(torch.zeros(1, ch, s, s))])
            B.check_anchor_order(det)
            det.anchors /= det.stride.view(-1, 1, 1)
            self.stride = det.stride
            self._initialize_dh_biases()
        elif isinstance(det, B.Detect):                               # :1196-1207 (anchors are scaled BEFORE the order check here)
            s = 256
            det.inplace = self.inplace
            det.stride = torch.tensor([s / x.shape[-2] for x in self.forward(torch.zeros(1, ch, s, s))])
            det.anchors /= det.stride.view(-1, 1, 1)
            B.check_anchor_order(det)
            self.stride = det.stride
            self._initialize_biases()
        B.initialize_weights(self)                                    # :1240

    def forward(self, x, augment=False, profile=False, visualize=False):
        if augment:
            return self._forward_augment(x)
        return self._forward_once(x)

    def _forward_augment(self, x):
        """Test-time augmentation (models/yolo.py:1253-1267): scales 1 / 0.83 / 0.67, each also flipped left-right; `scale_img`
        (utils/torch_utils.py:270-282) = bilinear resize to int(size*ratio), padded with 0.447 up to the next stride multiple;
        predictions de-scaled / un-flipped (:1292-1308), the first pass loses its coarsest level's rows and the last its finest
        (:1310-1318)."""
        import torch.nn.functional as F
        img_size = x.shape[-2:]
        gs = int(self.stride.max())
        y = []
        for si, fi in zip([1, 1, 0.83, 0.83, 0.67, 0.67], [None, 3, None, 3, None, 3]):
            xi = x.flip(fi) if fi else x
            if si != 1.0:
                h, w = xi.shape[2:]
                s = (int(h * si), int(w * si))
                xi = F.interpolate(xi, size=s, mode='bilinear', align_corners=False)
                hp, wp = (math.ceil(v * si / gs) * gs for v in (h, w))
                xi = F.pad(xi, [0, wp - s[1], 0, hp - s[0]], value=0.447)
            p = self._forward_once(xi)[0]
            px, py, pwh = p[..., 0:1] / si, p[..., 1:2] / si, p[..., 2:4] / si
            if fi == 3:
                px = img_size[1] - px
            y.append(torch.cat((px, py, pwh, p[..., 4:]), -1))
        nl = self.model[-1].nl
        g = sum(4 ** v for v in range(nl))
        y[0] = y[0][:, :-(y[0].shape[1] // g)]
        y[-1] = y[-1][:, (y[-1].shape[1] // g) * 4 ** (nl - 1):]
        return torch.cat(y, 1), None

    def _forward_once(self, x):
        """Walk the layers with the skip list (models/yolo.py:1269-1290)."""
        y = []
        for m in self.model:
            if m.f != -1:
                x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
            if isinstance(m, nn.Upsample):
                m.recompute_scale_factor = False
            if isinstance(m, B.ODConv_3rd):
                x = x.contiguous()
            x = m(x)
            y.append(x if m.i in self.save else None)
        return x

    def _initialize_dh_biases(self, cf=None):
        """obj / cls bias priors of the decoupled head (models/yolo.py:1334-1345)."""
        det = self.model[-1]
        for mi, s in zip(det.m, det.stride):
            b = mi.b3.bias.view(det.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            mi.b3.bias = nn.Parameter(b.view(-1), requires_grad=True)
            b = mi.c3.bias.data
            b += math.log(0.6 / (det.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
            mi.c3.bias = nn.Parameter(b, requires_grad=True)

    def _initialize_biases(self, cf=None):
        """obj / cls bias priors of the plain Detect head (models/yolo.py:1355-1366)."""
        det = self.model[-1]
        for mi, s in zip(det.m, det.stride):
            b = mi.bias.view(det.na, -1)
            b.data[:, 4] += math.log(8 / (640 / s) ** 2)
            b.data[:, 5:] += math.log(0.6 / (det.nc - 0.999999)) if cf is None else torch.log(cf / cf.sum())
            mi.bias = nn.Parameter(b.view(-1), requires_grad=True)

    def fuse(self):
        """Fold Conv+BN pairs for inference (models/yolo.py:1413-1428): only `Conv` instances."""
        for m in self.model.modules():
            if isinstance(m, B.Conv) and hasattr(m, 'bn'):
                m.conv = B.fuse_conv_and_bn(m.conv, m.bn)
                delattr(m, 'bn')
        return self
