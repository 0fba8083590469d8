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
