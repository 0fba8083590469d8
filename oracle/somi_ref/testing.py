"""Oracle (test infrastructure): deterministic, construction-order-independent parameter fill and
the synthetic VisDrone-shaped batch of SURVEY.md section 8d.

``fill_state`` assigns every parameter and buffer of a module from a generator seeded by the
CRC32 of its state_dict name, so the reference model (built in oracle/gen_golden.py) and the
restated model (built anywhere) get bit-identical weights without shipping a state_dict.
"""
import math
import zlib

import numpy as np
import torch


def fill_state(module, seed=0):
    """In-place deterministic fill of module.state_dict(); returns the module."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if not t.dtype.is_floating_point:
                continue                                        # num_batches_tracked etc.
            g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
            leaf = name.rsplit('.', 1)[-1]
            if leaf == 'running_var':
                v = torch.rand(t.shape, generator=g) + 0.5       # U(0.5, 1.5)
            elif leaf == 'running_mean':
                v = torch.randn(t.shape, generator=g) * 0.1
            elif leaf == 'anchors':
                continue                                        # geometry, not a weight
            elif t.dim() == 1 and leaf == 'weight':              # BN / LN scale (BiFPN weight too)
                v = torch.rand(t.shape, generator=g) + 0.5
            elif t.dim() <= 1 or leaf == 'bias':
                v = torch.randn(t.shape, generator=g) * 0.1
            else:                                               # conv / linear / ODConv kernels
                fan_in = t[0].numel() if t.dim() < 5 else t[0, 0].numel()
                v = torch.randn(t.shape, generator=g) * (1.0 / math.sqrt(max(fan_in, 1)))
            t.copy_(v.to(t.dtype))
    return module


def synthetic_batch(batch, size, nc=10, seed=0):
    """uint8 images (B,3,S,S) and targets (nt,6) = [img, cls, x, y, w, h] as SURVEY.md section 8d specifies."""
    rng = np.random.RandomState(seed)
    imgs = torch.from_numpy(rng.randint(0, 256, (batch, 3, size, size), dtype=np.uint8))
    rows = []
    for b in range(batch):
        n = int(np.clip(rng.poisson(54), 1, 300))
        cls = rng.randint(0, nc, n).astype(np.float32)
        xy = rng.uniform(0.02, 0.98, (n, 2)).astype(np.float32)
        wh = np.clip(np.exp(rng.normal(math.log(0.03), 0.7, (n, 2))), 0.004, 0.5).astype(np.float32)
        rows.append(np.concatenate([np.full((n, 1), b, np.float32), cls[:, None], xy, wh], 1))
    return imgs, torch.from_numpy(np.concatenate(rows, 0))


SOMI_ANCHORS = [[4, 6, 12, 8, 7, 14, 20, 12], [13, 22, 31, 18, 21, 33, 46, 23],
                [37, 37, 31, 56, 65, 34, 55, 57], [95, 52, 60, 90, 147, 76, 103, 134]]
"""The 16 anchor pairs listed in models/modules/YOLO-SOMI.yaml:9 (comment), 4 per level P2..P5."""

HYP_VISDRONE = dict(lr0=0.0032, lrf=0.12, momentum=0.843, weight_decay=0.00036, warmup_epochs=2.0,
                    warmup_momentum=0.5, warmup_bias_lr=0.05, box=0.07, cls=0.18, cls_pw=0.631, obj=0.15,
                    obj_pw=0.911, iou_t=0.2, anchor_t=3, fl_gamma=0.0, alpha=0.01, beta=0.1, Rp_nms=0.1,
                    deta=0.5, slide_ratio=0, nwdloss=0, shapeloss=0, label_smoothing=0.0)
"""Loss-relevant keys of data/hyps/hyp.VisDrone.yaml (values are configuration data)."""

HYP_AUGMENT = dict(hsv_h=0.4, hsv_s=0.3, hsv_v=0.5, degrees=0.2, translate=0.0, scale=0.4, shear=0.0, perspective=0.0,
                   flipud=0.0, fliplr=0.5, mosaic=1.0, mixup=0.2, copy_paste=0.0)
"""Augmentation keys of data/hyps/hyp.VisDrone.yaml:17-29."""


def synthetic_image_set(img_size, n=6, seed=0):
    """`n` BGR uint8 images whose longer side is `img_size` (what the reference's image cache holds) with (k,5) float32
    [cls, x, y, w, h] labels: smooth colour gradients plus noise, so HSV jitter and bilinear taps see varied values."""
    rng = np.random.RandomState(seed)
    fr = [(1.0, 0.75), (0.75, 1.0), (1.0, 1.0), (0.625, 1.0), (1.0, 0.875), (1.0, 1.0), (0.5, 1.0), (1.0, 0.5)]
    imgs, labels = [], []
    for i in range(n):
        h, w = (max(8, int(round(img_size * f))) for f in fr[i % len(fr)])
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
        base = np.stack([127 + 120 * np.sin(xx / (3 + 2 * c + i) + c) * np.cos(yy / (5 + i - c) + i) for c in range(3)], -1)
        img = np.clip(base + rng.randint(-40, 41, size=(h, w, 3)), 0, 255).astype(np.uint8)
        if i % 4 == 3:
            img[: h // 3, : w // 3] = rng.randint(0, 256, size=3)          # a flat patch: saturation 0 / grey paths
        k = int(rng.randint(3, 9)) if i != 1 else 0                         # image 1 has no labels
        wh = rng.uniform(0.05, 0.35, size=(k, 2))
        xy = rng.uniform(0, 1, size=(k, 2)) * (1 - wh) + wh / 2
        labels.append(np.concatenate((rng.randint(0, 10, size=(k, 1)), xy, wh), 1).astype(np.float32).reshape(k, 5))
        imgs.append(img)
    return imgs, labels


def somi_cfg(width=1.0, depth=1.0, nc=10, anchors=4, dcn=False, dcn_group=8):
    """The layer table of models/modules/YOLO-SOMI.yaml as a dict (C2fEACBAM -> C2fCBAM).

    dcn=True: "yolov5l-SOMI (DCNv3 blocks)" of BASELINE configs[1].  The reference vendors DCNv3 but wires it into no yaml (SURVEY
    fact 3), so the sites are the build's choice: one DCNv3_YOLO block (DCNv3 -> BN -> SiLU, 3x3, `dcn_group` groups) behind each of
    the two high-resolution lateral convs of the neck - P2 (160x160 at 640) and P3 (80x80), 256 channels: the representative
    shapes of SURVEY section 8a row F10.  Later layers shift by one / two; their `from` indices are rewritten accordingly."""
    bb = [[-1, 1, 'Conv', [64, 3, 2]], [-1, 1, 'ODConv_3rd', [128, 3, 2, 4]], [-1, 3, 'C2fCBAM', [128, True]],
          [-1, 1, 'Conv', [256, 3, 2]], [-1, 6, 'C2fCBAM', [256, True]], [-1, 1, 'Conv', [512, 3, 2]],
          [-1, 6, 'C2fCBAM', [512, True]], [-1, 1, 'Conv', [1024, 3, 2]], [-1, 3, 'C2fCBAM', [1024, True]],
          [-1, 1, 'SPPF', [1024, 5]]]
    up = [-1, 1, 'nn.Upsample', [None, 2, 'nearest']]
    hd = [[2, 1, 'Conv', [256]], [4, 1, 'Conv', [256]], [6, 1, 'Conv', [256]], [9, 1, 'Conv', [256]],
          up, [[-1, 12], 1, 'BiFPN', []], [-1, 1, 'SEAM', [256, 1, 16]], [-1, 3, 'C2fCBAM', [256]],
          up, [[-1, 11], 1, 'BiFPN', []], [-1, 1, 'SEAM', [256, 1, 16]], [-1, 3, 'C2fCBAM', [256]],
          up, [[-1, 10], 1, 'BiFPN', []], [-1, 1, 'SEAM', [256, 1, 16]], [-1, 3, 'C2fCBAM', [256]],
          [-1, 1, 'ODConv_3rd', [256, 3, 2, 4]], [[-1, 11, 21], 1, 'BiFPN', []], [-1, 3, 'C2fCBAM', [256]],
          [-1, 1, 'ODConv_3rd', [256, 3, 2, 4]], [[-1, 12, 17], 1, 'BiFPN', []], [-1, 3, 'C2fCBAM', [512]],
          [-1, 1, 'ODConv_3rd', [256, 3, 2, 4]], [[-1, 13], 1, 'BiFPN', []], [-1, 3, 'C2fCBAM', [1024]],
          [[25, 28, 31, 34], 1, 'DecoupledDetect', ['nc', 'anchors']]]
    import copy
    bb, hd = copy.deepcopy(bb), copy.deepcopy(hd)
    if dcn:
        layers = bb + hd
        for after in (11, 10):                                   # insert behind layer 11 first, so that index 10 stays valid
            for l in layers:                                     # absolute references to later layers move up by one
                l[0] = [j + 1 if j > after else j for j in l[0]] if isinstance(l[0], list) else (l[0] + 1 if l[0] > after else l[0])
            layers.insert(after + 1, [after, 1, 'DCNv3_YOLO', [256, 3, 1, dcn_group]])
        # the lateral convs' other consumers (the BiFPN inputs) now read the DCNv3 output: references to 10 / 11 move to 11 / 13
        for l in layers:
            if l[2] != 'DCNv3_YOLO':
                l[0] = [{10: 11, 12: 13}.get(j, j) for j in l[0]] if isinstance(l[0], list) else {10: 11, 12: 13}.get(l[0], l[0])
        bb, hd = layers[:len(bb)], layers[len(bb):]
    return dict(nc=nc, depth_multiple=depth, width_multiple=width, anchors=copy.deepcopy(anchors), backbone=bb, head=hd)


COCO_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
"""The stock YOLOv5 P3-P5 anchors (upstream yolov5s.yaml; the reference ships no yolov5s.yaml, SURVEY section 2 #21)."""


def yolov5_cfg(width=0.50, depth=0.33, nc=80, anchors=None, version='6.0'):
    """Stock YOLOv5 layer tables authored here (BASELINE configs[0]; the reference ships none): version '6.0' = Conv 6x6 stem + SPPF
    (upstream yolov5s.yaml of the release this fork is based on); '5.0' = Focus stem + SPP(5,9,13), which exercises the remaining
    stock modules.  Defaults are yolov5s (depth 0.33, width 0.50, 80 classes -> 7,235,389 parameters for '6.0')."""
    import copy
    if version == '6.0':
        bb = [[-1, 1, 'Conv', [64, 6, 2, 2]], [-1, 1, 'Conv', [128, 3, 2]], [-1, 3, 'C3', [128]], [-1, 1, 'Conv', [256, 3, 2]],
              [-1, 6, 'C3', [256]], [-1, 1, 'Conv', [512, 3, 2]], [-1, 9, 'C3', [512]], [-1, 1, 'Conv', [1024, 3, 2]],
              [-1, 3, 'C3', [1024]], [-1, 1, 'SPPF', [1024, 5]]]
    else:
        bb = [[-1, 1, 'Focus', [64, 3]], [-1, 1, 'Conv', [128, 3, 2]], [-1, 3, 'C3', [128]], [-1, 1, 'Conv', [256, 3, 2]],
              [-1, 9, 'C3', [256]], [-1, 1, 'Conv', [512, 3, 2]], [-1, 9, 'C3', [512]], [-1, 1, 'Conv', [1024, 3, 2]],
              [-1, 1, 'SPP', [1024, [5, 9, 13]]], [-1, 3, 'C3', [1024, False]]]
    up = [-1, 1, 'nn.Upsample', [None, 2, 'nearest']]
    hd = [[-1, 1, 'Conv', [512, 1, 1]], up, [[-1, 6], 1, 'Concat', [1]], [-1, 3, 'C3', [512, False]],
          [-1, 1, 'Conv', [256, 1, 1]], up, [[-1, 4], 1, 'Concat', [1]], [-1, 3, 'C3', [256, False]],
          [-1, 1, 'Conv', [256, 3, 2]], [[-1, 14], 1, 'Concat', [1]], [-1, 3, 'C3', [512, False]],
          [-1, 1, 'Conv', [512, 3, 2]], [[-1, 10], 1, 'Concat', [1]], [-1, 3, 'C3', [1024, False]],
          [[17, 20, 23], 1, 'Detect', ['nc', 'anchors']]]
    return dict(nc=nc, depth_multiple=depth, width_multiple=width, anchors=copy.deepcopy(anchors or COCO_ANCHORS),
                backbone=copy.deepcopy(bb), head=copy.deepcopy(hd))


def tiny_somi_cfg(nc=10):
    """A cut-down graph that uses every module class of the SOMI yaml once (Conv, ODConv_3rd, C2fCBAM, SPPF, nn.Upsample, BiFPN, SEAM,
    DecoupledDetect) at 32 / 64 channels, two detection levels: ~0.2 M parameters - for fixtures that carry whole pickled models."""
    import copy
    bb = [[-1, 1, 'Conv', [32, 3, 2]], [-1, 1, 'ODConv_3rd', [32, 3, 2, 4]], [-1, 1, 'C2fCBAM', [32, True]], [-1, 1, 'Conv', [64, 3, 2]],
          [-1, 1, 'SPPF', [64, 5]]]
    hd = [[2, 1, 'Conv', [32]], [4, 1, 'Conv', [32]], [-1, 1, 'nn.Upsample', [None, 2, 'nearest']], [[-1, 5], 1, 'BiFPN', []],
          [-1, 1, 'SEAM', [32, 1, 16]], [-1, 1, 'C2fCBAM', [32]], [[10, 6], 1, 'DecoupledDetect', ['nc', 'anchors']]]
    return dict(nc=nc, depth_multiple=1.0, width_multiple=1.0, anchors=[[4, 6, 12, 8, 7, 14, 20, 12], [13, 22, 31, 18, 21, 33, 46, 23]],
                backbone=copy.deepcopy(bb), head=copy.deepcopy(hd))


class AbsTermSums:
    """Condition numbers of parameter gradients, measured - test infrastructure for the training-parity tests.

    A parameter gradient is a sum of products, g = sum_t a_t * b_t (pixels x samples for a conv weight, pixels for a bias or a
    BatchNorm scale, ...).  An fp32 realisation of that sum - whatever its order - is off from the exact value by a multiple of
    2^-24 * S with S = sum_t |a_t * b_t|; when the terms cancel (S >> |g|) a RELATIVE bar on g is meaningless, and a bar on
    |error| / (2^-24 * S) is the principled one.  This context computes S for every parameter of a model during ONE ordinary
    fp64 autograd pass (forward hooks keep the layer inputs, tensor hooks on the layer outputs see the output gradients):

        with AbsTermSums(model64) as cond:
            loss(model64(x)).backward()
        S = cond.sums            # {parameter name: tensor like the parameter}

    Covered: nn.Conv2d / nn.Linear / nn.BatchNorm2d / nn.LayerNorm (everything the graphs are built from), the candidate kernels and
    biases of ODConv2d_3rd, the BiFPN fusion weights.  A parameter of any other kind is simply absent from `sums`.
    """

    def __init__(self, model, sums=True, squares=True):
        """sums / squares: which of S = sum |terms| and sum terms^2 to measure (each costs one extra weight-gradient convolution per conv
        layer in fp64: a caller that needs only `sums` (conditioned_errors, fp64_anchored_errors) or only `rss` (noise_scaled_errors) says so)."""
        self.model, self.sums, self.sq, self._handles = model, {}, {}, []
        self.want_s, self.want_q = bool(sums), bool(squares)
        self._names = {id(p): n for n, p in model.named_parameters()}

    def _add(self, p, s, q2=None):
        """s: sum |terms|; q2: sum terms^2 (optional) - both shaped like the parameter (or reshapeable to it)."""
        n = self._names.get(id(p))
        if n is not None:
            if s is not None:
                s = s.detach().reshape(p.shape)
                self.sums[n] = self.sums[n] + s if n in self.sums else s
            if q2 is not None:
                q2 = q2.detach().reshape(p.shape)
                self.sq[n] = self.sq[n] + q2 if n in self.sq else q2

    @property
    def rss(self):
        """{parameter name: sqrt(sum terms^2)} - the scale of what INDEPENDENT relative perturbations of the terms (the fp32 rounding of
        everything upstream) do to the sum: between |g| (no cancellation) and S (full cancellation, all terms alike)."""
        return {n: v.clamp_min(0).sqrt() for n, v in self.sq.items()}

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        from .blocks import BiFPN, ODConv2d_3rd

        def on_output(mod, fn):
            def fwd(m, inp, out):
                if torch.is_grad_enabled() and isinstance(out, torch.Tensor) and out.requires_grad:
                    saved = [t.detach() if isinstance(t, torch.Tensor) else t for t in inp]
                    out.register_hook(lambda g: fn(m, saved, g.detach()))
            self._handles.append(mod.register_forward_hook(fwd))

        def conv(m, inp, dy):
            x = inp[0]
            self._add(m.weight,
                      torch.nn.grad.conv2d_weight(x.abs(), m.weight.shape, dy.abs(), m.stride, m.padding, m.dilation, m.groups) if self.want_s else None,
                      torch.nn.grad.conv2d_weight(x * x, m.weight.shape, dy * dy, m.stride, m.padding, m.dilation, m.groups) if self.want_q else None)
            if m.bias is not None:
                self._add(m.bias, dy.abs().sum((0, 2, 3)), (dy * dy).sum((0, 2, 3)))

        def linear(m, inp, dy):
            x = inp[0].reshape(-1, inp[0].shape[-1])
            d = dy.reshape(-1, dy.shape[-1])
            self._add(m.weight, d.abs().t() @ x.abs(), (d * d).t() @ (x * x))
            if m.bias is not None:
                self._add(m.bias, d.abs().sum(0), (d * d).sum(0))

        def bnorm(m, inp, dy):
            x = inp[0]
            if m.training or m.running_mean is None:
                mean, var = x.mean((0, 2, 3)), x.var((0, 2, 3), unbiased=False)
            else:
                mean, var = m.running_mean, m.running_var
            xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + m.eps)
            if m.weight is not None:
                t = dy * xh
                self._add(m.weight, t.abs().sum((0, 2, 3)), (t * t).sum((0, 2, 3)))
                self._add(m.bias, dy.abs().sum((0, 2, 3)), (dy * dy).sum((0, 2, 3)))

        def lnorm(m, inp, dy):
            x = inp[0]
            dims = tuple(range(-len(m.normalized_shape), 0))
            xh = (x - x.mean(dims, keepdim=True)) / torch.sqrt(x.var(dims, unbiased=False, keepdim=True) + m.eps)
            lead = tuple(range(x.dim() - len(m.normalized_shape)))
            if m.weight is not None:
                t = dy * xh
                self._add(m.weight, t.abs().sum(lead), (t * t).sum(lead))
                self._add(m.bias, dy.abs().sum(lead), (dy * dy).sum(lead))

        def odconv(m, inp, dy):
            # dW[k] = sum_b attn[b,k] * dW_b with dW_b the weight gradient of sample b's own convolution (blocks.ODConv2d_3rd.get_weight_bias)
            x = inp[0]
            ctx = inp[1] if len(inp) > 1 and isinstance(inp[1], torch.Tensor) else x
            B = x.shape[0]
            with torch.no_grad():
                a_f, a_s, a_c, a_w = m.attentions(ctx)
                attn = a_f.view(B, 1, -1, 1, 1, 1)
                if a_s is not None:
                    attn = attn * a_s.view(B, 1, 1, 1, *m.kernel_size)
                if a_c is not None:
                    attn = attn * a_c.view(B, 1, 1, -1, 1, 1)
                if a_w is not None:
                    attn = attn * a_w.view(B, -1, 1, 1, 1, 1)
                per = m.weight.shape[1:]
                s, s2 = torch.zeros_like(m.weight), torch.zeros_like(m.weight)
                for b in range(B):
                    xb, db = x[b:b + 1], dy[b:b + 1]
                    if self.want_s:
                        s += attn[b].abs() * torch.nn.grad.conv2d_weight(xb.abs(), per, db.abs(), m.stride, m.padding, m.dilation, m.groups)[None]
                    if self.want_q:
                        s2 += attn[b] ** 2 * torch.nn.grad.conv2d_weight(xb * xb, per, db * db, m.stride, m.padding, m.dilation, m.groups)[None]
                self._add(m.weight, s, s2)
                if m.bias is not None:
                    w = a_w.abs() if a_w is not None else torch.ones(B, m.K, dtype=dy.dtype)
                    self._add(m.bias, w.t() @ dy.abs().sum((2, 3)), (w * w).t() @ (dy * dy).sum((2, 3)))

        def bifpn(m, inp, dy):
            xs = inp[0]
            s = torch.stack([(dy * x).abs().sum() for x in xs])                   # |terms| of d(loss)/d(w_i)
            wgt = m.weight.detach().clone().requires_grad_(True)
            with torch.enable_grad():
                jac = torch.autograd.functional.jacobian(lambda v: v / (m.swish(v).sum(dim=0) + m.epsilon), wgt)   # d w_i / d weight_j
            s2 = torch.stack([((dy * x) ** 2).sum() for x in xs])
            self._add(m.weight, (jac.abs() * s[:len(xs), None]).sum(0), (jac ** 2 * s2[:len(xs), None]).sum(0))

        for mod in self.model.modules():
            if isinstance(mod, ODConv2d_3rd):
                on_output(mod, odconv)
            elif type(mod) is nn.Conv2d:
                on_output(mod, conv)
            elif isinstance(mod, nn.Linear):
                on_output(mod, linear)
            elif isinstance(mod, nn.BatchNorm2d):
                on_output(mod, bnorm)
            elif isinstance(mod, nn.LayerNorm):
                on_output(mod, lnorm)
            elif isinstance(mod, BiFPN):
                on_output(mod, bifpn)
        return self

    def __exit__(self, *exc):
        for h in self._handles:
            h.remove()
        self._handles = []
        return False


def conditioned_errors(named_grads, named_ref64, sums, eps=2.0 ** -24, rel=1e-3):
    """Per parameter that has both a reference gradient and a measured S (AbsTermSums):
        (name, c_req, rel_err, cond)
    rel_err = max |g - g64| / max |g64|; cond = max S / max |g64| (how ill-conditioned the sums are); c_req = the smallest c with
        |g - g64|_e <= rel * max |g64| + c * eps * S_e      for every element e,
    i.e. 0 when the plain relative bar holds, else the error beyond it in units of one fp32 rounding of the gradient's own terms."""
    out = []
    for n, g in named_grads:
        g64 = named_ref64.get(n)
        s = sums.get(n)
        if g64 is None or s is None:
            continue
        d = (g.detach().cpu().double() - g64.double()).abs()
        scale = g64.abs().max().item() + 1e-300
        s = s.double().clamp_min(1e-300)
        c_req = ((d - rel * scale).clamp_min(0) / (eps * s)).max().item()
        out.append((n, c_req, d.max().item() / scale, s.max().item() / scale))
    return out


def noise_scaled_errors(named_grads, named_ref64, rss, against=None):
    """Per parameter with a measured root-sum-square Q of its gradient's terms (AbsTermSums.rss): (name, r, rel) with
        r   = max |g - g64| / max (|g64| + Q)     the error in units of the parameter's own noise scale,
        rel = max |g - g64| / max |g64|           the plain relative error.
    against: {name: gradient} of another run - the DIFFERENCE is then taken against that run instead of g64 (same normalisation): how far
    two runs of one path are apart, e.g. with and without a rounding-sized perturbation of the weights (the parameter's measured condition).
    Independent relative perturbations of size u of the terms (what the fp32 rounding of every layer upstream amounts to) move the sum by
    about u * Q; correlated ones by up to u * |g|: r is directly comparable to u across well- and ill-conditioned parameters."""
    out = []
    for n, g in named_grads:
        g64, q = named_ref64.get(n), rss.get(n)
        if g64 is None or q is None:
            continue
        other = g64 if against is None else against[n].detach().cpu()
        d = (g.detach().cpu().double() - other.double()).abs().max().item()
        out.append((n, d / ((g64.double().abs() + q.double()).max().item() + 1e-300), d / (g64.abs().max().item() + 1e-300)))
    return out


def fp64_anchored_errors(named_grads, named_cpu32, named_ref64, sums, rel=1e-3, k_cpu=2.0, c0=16.0, eps=2.0 ** -24):
    """The gradient bar every training-parity test shares, anchored on the fp64 oracle (test infrastructure).

    For every parameter with an fp64 reference gradient g64, an fp32 CPU oracle gradient and a measured S = sum |terms| (AbsTermSums):
        (name, ratio, e, e_cpu, scale)       scale = max |g64|,  e = max |g - g64|,  e_cpu = max |g_cpu32 - g64|,
        ratio = max over elements of  |g - g64|_el / ( max(rel * scale, k_cpu * e_cpu) + c0 * eps * S_el )
    ratio <= 1 passes: BASELINE's flat `rel` (1e-3) of the gradient's scale; where the fp32 CPU restatement of the SAME computation is itself
    farther than that from fp64 (deep chaotic graphs, the floor() discontinuity of DCNv3's offset gradient) `k_cpu` times the CPU path's own
    distance - a quantity that does not depend on the implementation under test (k_cpu = 0: the flat bar alone); plus a FIXED `c0` fp32
    roundings of the gradient's own terms, which only matters for sums of cancelling terms (S >> |g|: biases in front of a normalisation,
    the 7x7 attention conv) where no relative bar means anything.  named_cpu32 may be None with k_cpu = 0."""
    cpu = dict(named_cpu32) if named_cpu32 is not None else {}
    out = []
    for n, g in named_grads:
        g64, s = named_ref64.get(n), sums.get(n)
        if g64 is None or s is None:
            continue
        g64 = g64.double()
        d = (g.detach().cpu().double() - g64).abs()
        scale = g64.abs().max().item()
        e_cpu = (cpu[n].detach().double() - g64).abs().max().item() if n in cpu else 0.0
        allow = max(rel * scale, k_cpu * e_cpu) + c0 * eps * s.double() + 1e-300
        out.append((n, (d / allow).max().item(), d.max().item(), e_cpu, scale))
    return out


class ForcedDecisions:
    """The oracle evaluated AT GIVEN arg-max decisions (test infrastructure).

    A max is continuous but its gradient is not: where two candidates of a max-pool / arg-max are closer than the fp32 rounding of what
    feeds them (relative gap < ~1e-6; measured on the full-width graph at 320x320, batch 4: one of 25 600 channel-max decisions of
    model.2.m.1 below 1e-6, ten of 600 000 SPPF window decisions below 1e-5), ANY fp32 evaluation may route the gradient to the other
    candidate, and every parameter upstream moves by ~1e-3 - a property of the function at that input, not an error of an implementation.
    To hold an implementation to a flat 1e-3 regardless, the oracle is told which candidate the implementation took and differentiates
    THAT function (its value differs from the true one by the gap, ~1e-6; its gradient is what those decisions imply):

        with ForcedDecisions(model64, table):  loss(model64(x)).backward()

    table: {module name as in named_modules(): decisions}
      ChannelAttentionModule  (B,C) int64: pixel index (h*W + w) of each channel's spatial maximum        (models/common.py:339-358  amax over H,W)
      SpatialAttentionModule  (B,H,W) int64: channel index of each pixel's maximum of ca*x                (models/common.py:392-405  amax over C)
      SPPF                    [(B,C,H,W) int64] x 3: pixel index of the 5x5 / 9x9 / 13x13 window maximum    (models/common.py:1846-1861: three chained
                              5x5 pools == these windows of the first pool's input)
    Modules without an entry keep their own decisions."""

    def __init__(self, model, table):
        self.model, self.table, self._saved = model, table, []

    def __enter__(self):
        from . import blocks as OB

        def ca_forward(mod, x):
            a = mod.shared_MLP(x.mean(dim=(2, 3)))
            m = mod.shared_MLP(x.flatten(2).gather(2, mod._forced[..., None]).squeeze(-1))
            return torch.sigmoid(a + m)[:, :, None, None]

        def sa_forward(mod, x):
            stats = torch.cat([x.mean(dim=1, keepdim=True), x.gather(1, mod._forced[:, None])], dim=1)
            return torch.sigmoid(mod.cv1(stats))

        def sppf_forward(mod, x):
            x = mod.cv1(x)
            f = x.flatten(2)
            return mod.cv2(torch.cat([x] + [f.gather(2, i.flatten(2)).view_as(x) for i in mod._forced], 1))
        kinds = {OB.ChannelAttentionModule: ca_forward, OB.SpatialAttentionModule: sa_forward, OB.SPPF: sppf_forward}
        for name, mod in self.model.named_modules():
            fn = kinds.get(type(mod))
            if fn is not None and name in self.table:
                d = self.table[name]
                mod._forced = [t.long() for t in d] if isinstance(d, (list, tuple)) else d.long()
                mod.forward = fn.__get__(mod)                      # instance attribute shadows the class method
                self._saved.append(mod)
        return self

    def __exit__(self, *exc):
        for mod in self._saved:
            del mod.forward
            del mod._forced
        self._saved = []
        return False


def max_decision_gaps(model, x):
    """{module name: smallest relative gap between the largest and second-largest candidate over the module's max decisions} for one forward
    of the oracle `model` on `x` (same module kinds as ForcedDecisions) - how close this input sits to a switch of the gradient."""
    import torch.nn.functional as F
    from . import blocks as OB
    out, hs = {}, []

    def rel_gap(cands):
        t = cands.topk(2, dim=-1).values
        return float(((t[..., 0] - t[..., 1]) / t[..., 0].abs().clamp_min(1e-30)).min())

    def ca(name):
        return lambda m, inp, o: out.__setitem__(name, rel_gap(inp[0].flatten(2)))

    def sa(name):
        return lambda m, inp, o: out.__setitem__(name, rel_gap(inp[0].permute(0, 2, 3, 1)))

    def sppf(name):
        def f(m, inp, o):
            x1 = m.cv1(inp[0])
            g = []
            for k in (5, 9, 13):
                u = F.pad(x1, (k // 2,) * 4, value=float('-inf')).unfold(2, k, 1).unfold(3, k, 1).flatten(4)
                g.append(rel_gap(u))
            out[name] = min(g)
        return f
    for name, mod in model.named_modules():
        mk = {OB.ChannelAttentionModule: ca, OB.SpatialAttentionModule: sa, OB.SPPF: sppf}.get(type(mod))
        if mk is not None:
            hs.append(mod.register_forward_hook(mk(name)))
    with torch.no_grad():
        model(x)
    for h in hs:
        h.remove()
    return out


def decision_disagreements(model, x, table):
    """(number of arg-max decisions in `table` that differ from the oracle `model`'s own on input `x`, largest relative difference between the
    oracle's values of the two candidates at such a decision).  A decision taken from an fp32 forward may differ from the fp64 oracle's only
    where the oracle's candidates are within fp32 noise of each other: the second number says how close they were (test infrastructure)."""
    import torch.nn.functional as F
    from . import blocks as OB
    stat, hs = [0, 0.0], []

    def account(cands, forced):
        """cands (..., n) oracle values of all candidates; forced (...) index taken by the other path."""
        own = cands.argmax(-1)
        diff = own != forced
        if diff.any():
            a = cands.gather(-1, own[..., None]).squeeze(-1)[diff]
            b = cands.gather(-1, forced[..., None]).squeeze(-1)[diff]
            stat[0] += int(diff.sum())
            stat[1] = max(stat[1], float(((a - b).abs() / a.abs().clamp_min(1e-30)).max()))

    def ca(name):
        return lambda m, inp, o: account(inp[0].flatten(2), table[name].long())

    def sa(name):
        return lambda m, inp, o: account(inp[0].permute(0, 2, 3, 1), table[name].long())

    def sppf(name):
        def f(m, inp, o):
            x1 = m.cv1(inp[0])
            H, W = x1.shape[2:]
            for k, forced in zip((5, 9, 13), table[name]):
                p = k // 2
                u = F.pad(x1, (p,) * 4, value=float('-inf')).unfold(2, k, 1).unfold(3, k, 1).flatten(4)
                forced = forced.long()                              # pixel index -> position inside the window
                fh, fw = forced // W, forced % W
                dh = fh - torch.arange(H).view(1, 1, H, 1) + p
                dw = fw - torch.arange(W).view(1, 1, 1, W) + p
                account(u, dh * k + dw)
        return f
    for name, mod in model.named_modules():
        mk = {OB.ChannelAttentionModule: ca, OB.SpatialAttentionModule: sa, OB.SPPF: sppf}.get(type(mod))
        if mk is not None and name in table:
            hs.append(mod.register_forward_hook(mk(name)))
    with torch.no_grad():
        model(x)
    for h in hs:
        h.remove()
    return stat[0], stat[1]
