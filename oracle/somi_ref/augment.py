"""Oracle (test infrastructure): CPU restatement of the training / validation input pipeline.

Follows `LoadImagesAndLabels.__getitem__` (utils/datasets.py:590-673) for images that are cached at the training size
(`load_image` :710-729 then returns the cache entry), `load_mosaic` (:732-798), `collate_fn` (:675-680) and, in
utils/augmentations.py, `random_perspective` (:126-208), `box_candidates` (:313-318), `augment_hsv` (:47-61), `mixup`
(:305-310) and `letterbox` (:91-124); the box helpers are utils/general.py:550-567, 617-626.  `copy_reduce_paste`
(:237-275) does nothing at the reference's `copy_paste: 0.0` (hyp.VisDrone.yaml:29) and `Albumentations` has no transform
when the package is absent (:15-36), so neither draws a random number; both are left out.

Random numbers come from the same module-level generators the reference uses (`random`, `np.random`) in the same order,
so seeding both reproduces the reference's sample exactly.  Pixel arithmetic goes through `cv_port` (OpenCV restated,
parity unpinned); everything else is pinned by tests/golden/augment_*.npz.
"""
import math
import random

import numpy as np
import torch

from . import cv_port as cv

FILL = (114, 114, 114)


def boxes_from_normalised(x, w, h, padw=0, padh=0):
    """(n,4) xywh in [0,1] -> xyxy pixels, shifted (general.py:550-556)."""
    y = np.copy(x)
    y[:, 0] = w * (x[:, 0] - x[:, 2] / 2) + padw
    y[:, 1] = h * (x[:, 1] - x[:, 3] / 2) + padh
    y[:, 2] = w * (x[:, 0] + x[:, 2] / 2) + padw
    y[:, 3] = h * (x[:, 1] + x[:, 3] / 2) + padh
    return y


def boxes_to_normalised(x, w, h, eps):
    """(n,4) xyxy pixels -> xywh in [0,1]; clips `x` in place to [0, w-eps] x [0, h-eps] first (general.py:559-567, 624-626)."""
    x[:, [0, 2]] = x[:, [0, 2]].clip(0, w - eps)
    x[:, [1, 3]] = x[:, [1, 3]].clip(0, h - eps)
    y = np.copy(x)
    y[:, 0] = ((x[:, 0] + x[:, 2]) / 2) / w
    y[:, 1] = ((x[:, 1] + x[:, 3]) / 2) / h
    y[:, 2] = (x[:, 2] - x[:, 0]) / w
    y[:, 3] = (x[:, 3] - x[:, 1]) / h
    return y


def keep_box(before, after, wh_thr=2, ar_thr=20, area_thr=0.10, eps=1e-16):
    """augmentations.py:313-318 on (4,n) boxes: survives if > 2 px each way, keeps > 10 % area, aspect ratio < 20."""
    w1, h1 = before[2] - before[0], before[3] - before[1]
    w2, h2 = after[2] - after[0], after[3] - after[1]
    ar = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (ar < ar_thr)


def random_affine(im, targets, hyp, border=(0, 0)):
    """augmentations.py:126-208 (box labels, no segments).  Draw order: perspective x2, angle, scale, shear x2, translate x2."""
    height = im.shape[0] + border[0] * 2
    width = im.shape[1] + border[1] * 2
    centre = np.eye(3)
    centre[0, 2], centre[1, 2] = -im.shape[1] / 2, -im.shape[0] / 2
    persp = np.eye(3)
    persp[2, 0] = random.uniform(-hyp['perspective'], hyp['perspective'])
    persp[2, 1] = random.uniform(-hyp['perspective'], hyp['perspective'])
    assert hyp['perspective'] == 0, 'warpPerspective is outside the restated path (hyp perspective: 0.0)'
    rot = np.eye(3)
    angle = random.uniform(-hyp['degrees'], hyp['degrees'])
    s = random.uniform(1 - hyp['scale'], 1 + hyp['scale'])
    rot[:2] = cv.getRotationMatrix2D(angle=angle, center=(0, 0), scale=s)
    shear = np.eye(3)
    shear[0, 1] = math.tan(random.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
    shear[1, 0] = math.tan(random.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
    shift = np.eye(3)
    shift[0, 2] = random.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * width
    shift[1, 2] = random.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * height
    M = shift @ shear @ rot @ persp @ centre
    if border[0] != 0 or border[1] != 0 or (M != np.eye(3)).any():
        im = cv.warpAffine(im, M[:2], dsize=(width, height), borderValue=FILL)
    n = len(targets)
    if n:
        corners = np.ones((n * 4, 3))
        corners[:, :2] = targets[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
        corners = (corners @ M.T)[:, :2].reshape(n, 8)
        xs, ys = corners[:, [0, 2, 4, 6]], corners[:, [1, 3, 5, 7]]
        new = np.concatenate((xs.min(1), ys.min(1), xs.max(1), ys.max(1))).reshape(4, n).T
        new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
        new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
        keep = keep_box(targets[:, 1:5].T * s, new.T)
        targets = targets[keep]
        targets[:, 1:5] = new[keep]
    return im, targets


def hsv_jitter(im, hgain, sgain, vgain):
    """augmentations.py:47-61, in place on a BGR uint8 image: one np.random.uniform(-1, 1, 3) draw, three 256-entry tables."""
    if hgain or sgain or vgain:
        r = np.random.uniform(-1, 1, 3) * [hgain, sgain, vgain] + 1
        hue, sat, val = cv.split(cv.cvtColor(im, cv.COLOR_BGR2HSV))
        x = np.arange(0, 256, dtype=r.dtype)
        tables = (((x * r[0]) % 180).astype(im.dtype), np.clip(x * r[1], 0, 255).astype(im.dtype),
                  np.clip(x * r[2], 0, 255).astype(im.dtype))
        hsv = cv.merge((cv.LUT(hue, tables[0]), cv.LUT(sat, tables[1]), cv.LUT(val, tables[2])))
        cv.cvtColor(hsv, cv.COLOR_HSV2BGR, dst=im)


def letterbox(im, new_shape, scaleup=True):
    """augmentations.py:91-124 with auto=False, scaleFill=False: centre the image on a 114-grey canvas of `new_shape`."""
    shape = im.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    new_shape = (int(new_shape[0]), int(new_shape[1]))
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    unpad = int(round(shape[1] * r)), int(round(shape[0] * r))
    dw, dh = (new_shape[1] - unpad[0]) / 2, (new_shape[0] - unpad[1]) / 2
    if shape[::-1] != unpad:
        im = cv.resize(im, unpad, interpolation=cv.INTER_LINEAR)
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return cv.copyMakeBorder(im, top, bottom, left, right, cv.BORDER_CONSTANT, value=FILL), (r, r), (dw, dh)


def aspect_ordered_batches(wh, batch_index, img_size, stride, pad):
    """datasets.py:497-523 (rect=True): sort by aspect ratio h/w; per batch the letterbox shape [h, w] is the unit box shrunk
    along one side to the extreme ratio of the batch, scaled to img_size and rounded up (with `pad` strides of slack) to the
    stride.  wh: (n,2) float64 [width, height] -> (order, (nb,2) int shapes)."""
    ar = wh[:, 1] / wh[:, 0]
    order = ar.argsort()
    ar = ar[order]
    nb = batch_index[-1] + 1
    unit = [[1, 1]] * nb
    for b in range(nb):
        mine = ar[batch_index == b]
        if mine.max() < 1:
            unit[b] = [mine.max(), 1]
        elif mine.min() > 1:
            unit[b] = [1, 1 / mine.min()]
    return order, np.ceil(np.array(unit) * img_size / stride + pad).astype(int) * stride


class CachedDataset:
    """The slice of `LoadImagesAndLabels` that runs per sample once images are cached (datasets.py:405-420 attributes).

    imgs: list of (h, w, 3) BGR uint8 arrays whose longer side is `img_size`; labels: list of (n, 5) float32 [cls, x, y, w, h]
    normalised."""

    def __init__(self, imgs, labels, img_size, hyp, augment=True, rect=False, batch_size=16, stride=32, pad=0.0, shapes=None):
        self.imgs, self.labels, self.img_size, self.hyp, self.augment = imgs, labels, img_size, hyp, augment
        self.rect = rect
        self.mosaic = augment and not rect
        self.mosaic_border = [-img_size // 2, -img_size // 2]
        self.n = len(imgs)
        self.indices = range(self.n)
        self.batch = np.floor(np.arange(self.n) / batch_size).astype(int)            # datasets.py:477-479
        if rect:
            wh = np.array([(im.shape[1], im.shape[0]) for im in imgs] if shapes is None else shapes, dtype=np.float64)
            self.order, self.batch_shapes = aspect_ordered_batches(wh, self.batch, img_size, stride, pad)
            self.imgs = [imgs[i] for i in self.order]
            self.labels = [labels[i] for i in self.order]

    def mosaic4(self, index):
        """datasets.py:732-798: four images around a random centre on a 2s x 2s canvas, then the affine crop to s x s."""
        s = self.img_size
        yc, xc = (int(random.uniform(-b, 2 * s + b)) for b in self.mosaic_border)
        picks = [index] + random.choices(self.indices, k=3)
        random.shuffle(picks)
        canvas = np.full((2 * s, 2 * s, 3), FILL[0], dtype=np.uint8)
        boxes = []
        for q, idx in enumerate(picks):
            img = self.imgs[idx]
            h, w = img.shape[:2]
            # each image is anchored with one corner on the centre (quadrants 0..3 = top-left, top-right, bottom-left,
            # bottom-right) and whatever sticks out of the canvas is cut off; this is what the reference's four explicit
            # (x1a, y1a, x2a, y2a) / (x1b, y1b, x2b, y2b) cases compute
            ox = xc - w if q % 2 == 0 else xc                 # canvas position of the image's own (0, 0)
            oy = yc - h if q < 2 else yc
            cx1, cy1, cx2, cy2 = max(ox, 0), max(oy, 0), min(ox + w, 2 * s), min(oy + h, 2 * s)
            canvas[cy1:cy2, cx1:cx2] = img[cy1 - oy:cy2 - oy, cx1 - ox:cx2 - ox]
            lab = self.labels[idx].copy()
            if lab.size:
                lab[:, 1:] = boxes_from_normalised(lab[:, 1:], w, h, ox, oy)
            boxes.append(lab)
        boxes = np.concatenate(boxes, 0)
        np.clip(boxes[:, 1:], 0, 2 * s, out=boxes[:, 1:])
        return random_affine(canvas, boxes, self.hyp, border=self.mosaic_border)

    def __getitem__(self, index):
        hyp = self.hyp
        if self.mosaic and random.random() < hyp['mosaic']:
            img, labels = self.mosaic4(index)
            shapes = None
            if random.random() < hyp['mixup']:
                img2, labels2 = self.mosaic4(random.randint(0, self.n - 1))
                r = np.random.beta(32.0, 32.0)
                img = (img * r + img2 * (1 - r)).astype(np.uint8)
                labels = np.concatenate((labels, labels2), 0)
        else:
            img = self.imgs[index]
            h, w = img.shape[:2]
            shape = self.batch_shapes[self.batch[index]] if self.rect else self.img_size
            img, ratio, pad = letterbox(img, shape, scaleup=self.augment)
            shapes = (h, w), ((1.0, 1.0), pad)
            labels = self.labels[index].copy()
            if labels.size:
                labels[:, 1:] = boxes_from_normalised(labels[:, 1:], ratio[0] * w, ratio[1] * h, padw=pad[0], padh=pad[1])
            if self.augment:
                img, labels = random_affine(img, labels, hyp)
        nl = len(labels)
        if nl:
            labels[:, 1:5] = boxes_to_normalised(labels[:, 1:5], w=img.shape[1], h=img.shape[0], eps=1e-3)
        if self.augment:
            img = np.ascontiguousarray(img)
            hsv_jitter(img, hgain=hyp['hsv_h'], sgain=hyp['hsv_s'], vgain=hyp['hsv_v'])
            if random.random() < hyp['flipud']:
                img = np.flipud(img)
                if nl:
                    labels[:, 2] = 1 - labels[:, 2]
            if random.random() < hyp['fliplr']:
                img = np.fliplr(img)
                if nl:
                    labels[:, 1] = 1 - labels[:, 1]
        out = torch.zeros((nl, 6))
        if nl:
            out[:, 1:] = torch.from_numpy(labels)
        img = np.ascontiguousarray(img.transpose((2, 0, 1))[::-1])          # HWC BGR -> CHW RGB
        return torch.from_numpy(img), out, shapes


def collate(samples):
    """datasets.py:675-680: stack the images, write the sample number into column 0 of each label block, concatenate."""
    imgs, labels, shapes = zip(*samples)
    for i, lab in enumerate(labels):
        lab[:, 0] = i
    return torch.stack(imgs, 0), torch.cat(labels, 0), shapes
