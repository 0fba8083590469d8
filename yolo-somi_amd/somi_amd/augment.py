"""Device input pipeline: the per-sample work of the reference's `LoadImagesAndLabels` (utils/datasets.py:405-420, 590-680)
with the image cache resident in HBM.

The reference's dataloader workers build every training sample on the CPU: four cached images are pasted onto a 2s x 2s
canvas (`load_mosaic` :732-798), cropped by a random affine map (`random_perspective`, utils/augmentations.py:126-208),
optionally blended with a second mosaic (`mixup` :305-310), colour-jittered in HSV (`augment_hsv` :47-61), flipped and
transposed to CHW RGB (:648-671).  Here the cache lives in HBM (the whole VisDrone training set at 640 px is a few GB of
the 288), the host only draws the random numbers and transforms the label boxes (a few hundred scalars per sample, the
same numpy arithmetic as the reference), and ONE kernel launch per batch (`somi_augment_u8`) produces the uint8
(B, 3, s, s) RGB batch that `Model.forward` ingests - no canvas is materialised, nothing crosses PCIe per step except
the 1.2 kB sample records.

Random numbers are drawn from the module-level `random` / `np.random` generators in the reference's order (mosaic?,
centre y then x, three extra indices, shuffle, perspective x2, angle, scale, shear x2, translate x2, mixup?, partner index,
second mosaic, beta, hsv gains, flipud?, fliplr?), so seeding them reproduces the reference's samples.

`copy_paste` acts on segment labels only (box datasets have none: a no-op at any value) and Albumentations has no transform
when the package is absent; neither draws a random number.  A non-zero `perspective` raises (0.0 in all the reference's files).
"""
import ctypes as C
import math
import random
from collections import namedtuple

import numpy as np
import torch

from . import _lib

FILL = 114

HYP_VISDRONE_AUGMENT = dict(hsv_h=0.4, hsv_s=0.3, hsv_v=0.5, degrees=0.2, translate=0.0, scale=0.4, shear=0.0,
                            perspective=0.0, flipud=0.0, fliplr=0.5, mosaic=1.0, mixup=0.2, copy_paste=0.0)
"""data/hyps/hyp.VisDrone.yaml:17-29."""


def xywhn2xyxy(x, w=640, h=640, padw=0, padh=0):
    """utils/general.py:550-556 on an (n,4) numpy array."""
    y = np.copy(x)
    cx, cy, bw, bh = x[:, 0], x[:, 1], x[:, 2], x[:, 3]
    y[:, 0], y[:, 2] = w * (cx - bw / 2) + padw, w * (cx + bw / 2) + padw
    y[:, 1], y[:, 3] = h * (cy - bh / 2) + padh, h * (cy + bh / 2) + padh
    return y


def xyxy2xywhn(x, w=640, h=640, clip=False, eps=0.0):
    """utils/general.py:559-567 (the clip is in place, as there)."""
    if clip:
        x[:, [0, 2]] = x[:, [0, 2]].clip(0, w - eps)
        x[:, [1, 3]] = x[:, [1, 3]].clip(0, h - eps)
    y = np.copy(x)
    y[:, 0], y[:, 1] = ((x[:, 0] + x[:, 2]) / 2) / w, ((x[:, 1] + x[:, 3]) / 2) / h
    y[:, 2], y[:, 3] = (x[:, 2] - x[:, 0]) / w, (x[:, 3] - x[:, 1]) / h
    return y


def box_candidates(box1, box2, wh_thr=2, ar_thr=20, area_thr=0.1, eps=1e-16):
    """utils/augmentations.py:313-318: which warped boxes (box2, (4,n)) survive relative to their originals (box1)."""
    w1, h1 = box1[2] - box1[0], box1[3] - box1[1]
    w2, h2 = box2[2] - box2[0], box2[3] - box2[1]
    aspect = np.maximum(w2 / (h2 + eps), h2 / (w2 + eps))
    return (w2 > wh_thr) & (h2 > wh_thr) & (w2 * h2 / (w1 * h1 + eps) > area_thr) & (aspect < ar_thr)


def _inverse_affine(M):
    """dst->src matrix as cv2.warpAffine derives it from the src->dst 2x3 matrix (double precision, same operation order)."""
    a, b, tx, c, d, ty = (float(v) for v in np.asarray(M, dtype=np.float64).reshape(6))
    det = a * d - b * c
    det = 1.0 / det if det != 0 else 0.0
    ia, id_ = d * det, a * det
    ib, ic = b * (-det), c * (-det)
    return [ia, ib, -ia * tx - ib * ty, ic, id_, -ic * tx - id_ * ty]


Plan = namedtuple('Plan', 'canvases mix_r luts flipud fliplr out_hw')
"""What the kernel needs for one sample: canvases (1 or 2), mixup ratio or None, (3,256) uint8 jitter tables or None, flips, and
the output size (h, w)."""


class _Canvas:
    """One virtual canvas: placed sources + optional affine crop (host-side description of `somi_aug_canvas`)."""

    def __init__(self, height, width):
        self.height, self.width, self.sources, self.minv = height, width, [], None

    def place(self, img_id, x1, y1, x2, y2, dx, dy):
        self.sources.append((img_id, x1, y1, x2, y2, dx, dy))


def rect_batch_shapes(wh, batch_index, img_size, stride, pad):
    """datasets.py:497-523: order the images by aspect ratio h/w and give every batch the smallest stride-multiple letterbox
    shape that holds its images.  wh: (n,2) float64 (width, height).  -> (order, batch_shapes (nb,2) int [h, w])."""
    ar = wh[:, 1] / wh[:, 0]
    order = ar.argsort()
    ar = ar[order]
    nb = int(batch_index[-1]) + 1
    shapes = [[1, 1]] * nb
    for i in range(nb):
        ari = ar[batch_index == i]
        lo, hi = ari.min(), ari.max()
        if hi < 1:
            shapes[i] = [hi, 1]
        elif lo > 1:
            shapes[i] = [1, 1 / lo]
    return order, np.ceil(np.array(shapes) * img_size / stride + pad).astype(int) * stride


class DeviceImageCache:
    """`LoadImagesAndLabels` with `cache_images` - on the GPU.

    imgs: list of (h, w, 3) BGR uint8 arrays (or tensors) whose longer side is `img_size` (what `load_image` caches,
    datasets.py:710-729); labels: list of (n, 5) float32 [cls, x, y, w, h] normalised (datasets.py:455).
    `ds[i]` -> (img uint8 (3, s, s) RGB on the device, labels_out (nl, 6), path, shapes) like `__getitem__`;
    `ds.batch(indices)` -> what `collate_fn` returns for those samples, from a single kernel launch.
    """

    def __init__(self, imgs, labels, img_size=640, hyp=None, augment=True, rect=False, batch_size=16, stride=32, pad=0.0,
                 shapes=None, img_files=None, device='cuda:0'):
        """rect / batch_size / stride / pad as in `LoadImagesAndLabels.__init__` (datasets.py:405-414): with rect=True the samples
        are re-ordered by aspect ratio and every run of `batch_size` of them shares one letterbox shape (val.py:129-138 uses
        rect=True, pad=0.5).  shapes: the (n,2) original (width, height) of the files if they differ from the cached sizes.
        img_files: the paths `__getitem__` / `collate_fn` hand back as their third element (indices when not given)."""
        if len(imgs) != len(labels) or not imgs:
            raise ValueError('need one label array per image')
        self.hyp = dict(HYP_VISDRONE_AUGMENT if hyp is None else hyp)
        if self.hyp.get('perspective', 0.0):
            raise NotImplementedError('perspective is 0.0 in every hyper-parameter file of the reference; warpPerspective is not built')
        # copy_paste needs segment labels (utils/augmentations.py:247 `if paste_prob and n` with n = len(segments)): box-only
        # datasets have none, so any value is a no-op there - and draws no random number - exactly as in the reference
        self.img_size, self.augment, self.rect = int(img_size), bool(augment), bool(rect)
        self.mosaic = self.augment and not self.rect
        self.mosaic_border = [-self.img_size // 2, -self.img_size // 2]
        self.device = torch.device(device)
        self.n = len(imgs)
        self.indices = range(self.n)
        self.labels = [np.asarray(l, dtype=np.float32).reshape(-1, 5) for l in labels]
        self.batch_index = np.floor(np.arange(self.n) / batch_size).astype(int)                # datasets.py:477-479
        self.order = np.arange(self.n)                                                    # position -> index into `imgs` as given
        self.img_files = list(img_files) if img_files is not None else list(range(self.n))
        if self.rect:
            wh = np.array([(im.shape[1], im.shape[0]) for im in imgs] if shapes is None else shapes,
                          dtype=np.float64)
            self.order, self.batch_shapes = rect_batch_shapes(wh, self.batch_index, self.img_size, stride, pad)
            imgs = [imgs[i] for i in self.order]
            self.labels = [self.labels[i] for i in self.order]
            self.img_files = [self.img_files[i] for i in self.order]
        self.img_hw, offsets, total = [], [], 0
        arrs = []
        for im in imgs:
            a = im.cpu().numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
                raise ValueError('cached images have to be (h, w, 3) uint8')
            if max(a.shape[:2]) > self.img_size:
                raise ValueError(f'cached image {a.shape[:2]} exceeds img_size {self.img_size}')
            arrs.append(np.ascontiguousarray(a))
            self.img_hw.append(a.shape[:2])
            offsets.append(total)
            total += (a.size + 255) // 256 * 256
        flat = torch.empty(total, dtype=torch.uint8)
        for a, o in zip(arrs, offsets):
            flat[o:o + a.size] = torch.from_numpy(a.reshape(-1))
        _lib.lib()                                          # fail now if the library is missing: there is no CPU path
        self.store = flat.to(self.device)                   # the HBM-resident image cache
        self._offsets = offsets

    def __len__(self):
        return self.n

    # ---------------------------------------------------------------- host side: random draws, geometry, label boxes
    def _affine(self, canvas, targets, border):
        """random_perspective (augmentations.py:126-208) for box labels: sets canvas.minv, returns the surviving labels."""
        hyp = self.hyp
        height, width = canvas.height + border[0] * 2, canvas.width + border[1] * 2
        Cm = np.eye(3)
        Cm[0, 2], Cm[1, 2] = -canvas.width / 2, -canvas.height / 2
        Pm = np.eye(3)
        Pm[2, 0] = random.uniform(-hyp['perspective'], hyp['perspective'])
        Pm[2, 1] = random.uniform(-hyp['perspective'], hyp['perspective'])
        Rm = np.eye(3)
        angle = random.uniform(-hyp['degrees'], hyp['degrees'])
        s = random.uniform(1 - hyp['scale'], 1 + hyp['scale'])
        rad = np.deg2rad(angle)                              # cv2.getRotationMatrix2D(center=(0, 0))
        alpha, beta = np.cos(rad) * s, np.sin(rad) * s
        Rm[0, :2], Rm[1, :2] = (alpha, beta), (-beta, alpha)
        Sm = np.eye(3)
        Sm[0, 1] = math.tan(random.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
        Sm[1, 0] = math.tan(random.uniform(-hyp['shear'], hyp['shear']) * math.pi / 180)
        Tm = np.eye(3)
        Tm[0, 2] = random.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * width
        Tm[1, 2] = random.uniform(0.5 - hyp['translate'], 0.5 + hyp['translate']) * height
        M = Tm @ Sm @ Rm @ Pm @ Cm
        if border[0] != 0 or border[1] != 0 or (M != np.eye(3)).any():
            canvas.minv = _inverse_affine(M[:2])
        elif (height, width) != (canvas.height, canvas.width):
            raise AssertionError('unreachable: identity map with a border')
        n = len(targets)
        if n:
            pts = np.ones((n * 4, 3))
            pts[:, :2] = targets[:, [1, 2, 3, 4, 1, 4, 3, 2]].reshape(n * 4, 2)
            pts = (pts @ M.T)[:, :2].reshape(n, 8)
            px, py = pts[:, [0, 2, 4, 6]], pts[:, [1, 3, 5, 7]]
            new = np.concatenate((px.min(1), py.min(1), px.max(1), py.max(1))).reshape(4, n).T
            new[:, [0, 2]] = new[:, [0, 2]].clip(0, width)
            new[:, [1, 3]] = new[:, [1, 3]].clip(0, height)
            keep = box_candidates(box1=targets[:, 1:5].T * s, box2=new.T, area_thr=0.10)
            targets = targets[keep]
            targets[:, 1:5] = new[keep]
        return targets, (height, width)

    def _mosaic(self, index):
        """load_mosaic (datasets.py:732-798): a 2s x 2s canvas description + labels after the affine crop to s x s."""
        s = self.img_size
        yc, xc = (int(random.uniform(-x, 2 * s + x)) for x in self.mosaic_border)
        indices = [index] + random.choices(self.indices, k=3)
        random.shuffle(indices)
        canvas = _Canvas(2 * s, 2 * s)
        labels4 = []
        for i, idx in enumerate(indices):
            h, w = self.img_hw[idx]
            left, top = i % 2 == 0, i < 2
            # canvas rectangle: the image touches the centre with the corner facing it, clipped to the canvas
            x1a, x2a = (max(xc - w, 0), xc) if left else (xc, min(xc + w, 2 * s))
            y1a, y2a = (max(yc - h, 0), yc) if top else (yc, min(2 * s, yc + h))
            # matching source rectangle: the part next to that corner
            x1b = w - (x2a - x1a) if left else 0
            y1b = h - (y2a - y1a) if top else 0
            padw, padh = x1a - x1b, y1a - y1b
            canvas.place(idx, x1a, y1a, x2a, y2a, padw, padh)
            lab = self.labels[idx].copy()
            if lab.size:
                lab[:, 1:] = xywhn2xyxy(lab[:, 1:], w, h, padw, padh)
            labels4.append(lab)
        labels4 = np.concatenate(labels4, 0)
        np.clip(labels4[:, 1:], 0, 2 * s, out=labels4[:, 1:])
        labels4, size = self._affine(canvas, labels4, self.mosaic_border)
        return canvas, labels4, size

    def plan(self, index):
        """Everything `__getitem__` decides for one sample: (canvases, mix_r, luts, flips), labels (nl,5) xywhn, shapes."""
        hyp, s = self.hyp, self.img_size
        index = self.indices[index]
        canvases, mix_r = [], None
        if self.mosaic and random.random() < hyp['mosaic']:
            canvas, labels, size = self._mosaic(index)
            canvases.append(canvas)
            shapes = None
            if random.random() < hyp['mixup']:
                canvas2, labels2, _ = self._mosaic(random.randint(0, self.n - 1))
                mix_r = float(np.random.beta(32.0, 32.0))
                canvases.append(canvas2)
                labels = np.concatenate((labels, labels2), 0)
        else:
            h, w = self.img_hw[index]
            sh, sw = (int(v) for v in self.batch_shapes[self.batch_index[index]]) if self.rect else (s, s)
            r = min(sh / h, sw / w)
            if not self.augment:
                r = min(r, 1.0)
            unpad = int(round(w * r)), int(round(h * r))
            if unpad != (w, h):
                raise NotImplementedError('letterbox would resize: cache images at img_size (datasets.py:722-727)')
            dw, dh = (sw - unpad[0]) / 2, (sh - unpad[1]) / 2
            top, left = int(round(dh - 0.1)), int(round(dw - 0.1))
            canvas = _Canvas(sh, sw)
            canvas.place(index, left, top, left + w, top + h, left, top)
            canvases.append(canvas)
            shapes = (h, w), ((1.0, 1.0), (dw, dh))
            labels = self.labels[index].copy()
            if labels.size:
                labels[:, 1:] = xywhn2xyxy(labels[:, 1:], r * w, r * h, padw=dw, padh=dh)
            size = (sh, sw)
            if self.augment:
                labels, size = self._affine(canvas, labels, (0, 0))
        nl = len(labels)
        if nl:
            labels[:, 1:5] = xyxy2xywhn(labels[:, 1:5], w=size[1], h=size[0], clip=True, eps=1E-3)
        luts, flipud, fliplr = None, False, False
        if self.augment:
            if hyp['hsv_h'] or hyp['hsv_s'] or hyp['hsv_v']:
                g = np.random.uniform(-1, 1, 3) * [hyp['hsv_h'], hyp['hsv_s'], hyp['hsv_v']] + 1
                x = np.arange(0, 256, dtype=g.dtype)
                luts = np.stack((((x * g[0]) % 180).astype(np.uint8), np.clip(x * g[1], 0, 255).astype(np.uint8),
                                 np.clip(x * g[2], 0, 255).astype(np.uint8)))
            if random.random() < hyp['flipud']:
                flipud = True
                if nl:
                    labels[:, 2] = 1 - labels[:, 2]
            if random.random() < hyp['fliplr']:
                fliplr = True
                if nl:
                    labels[:, 1] = 1 - labels[:, 1]
        return Plan(canvases, mix_r, luts, flipud, fliplr, (int(size[0]), int(size[1]))), labels, shapes

    # ---------------------------------------------------------------- device side
    def _fill_record(self, rec, plan):
        canvases, mix_r, luts, flipud, fliplr, _ = plan
        base = self.store.data_ptr()
        for ci, canvas in enumerate(canvases):
            c = rec.canvas[ci]
            if len(canvas.sources) > 4:
                raise RuntimeError('a canvas holds at most four sources')
            c.nsrc, c.height, c.width = len(canvas.sources), canvas.height, canvas.width
            for si, (idx, x1, y1, x2, y2, dx, dy) in enumerate(canvas.sources):
                h, w = self.img_hw[idx]
                if x2 > x1 and y2 > y1 and not (0 <= x1 - dx and x2 - dx <= w and 0 <= y1 - dy and y2 - dy <= h):
                    raise RuntimeError('mosaic rectangle maps outside its source image')       # never launch an OOB read
                src = c.src[si]
                src.pixels, src.h, src.w = base + self._offsets[idx], h, w
                src.x1, src.y1, src.x2, src.y2, src.dx, src.dy = x1, y1, x2, y2, dx, dy
            c.warp = int(canvas.minv is not None)
            if canvas.minv is not None:
                for k in range(6):
                    c.minv[k] = canvas.minv[k]
        rec.mix = int(mix_r is not None)
        rec.mix_r = mix_r if mix_r is not None else 1.0
        rec.hsv = int(luts is not None)
        if luts is not None:
            C.memmove(rec.lut, luts.ctypes.data, 768)
        rec.flipud, rec.fliplr = int(flipud), int(fliplr)

    def upload(self, plans):
        """The `somi_aug_sample` records of `plans` as one device byte tensor (validated on the host first)."""
        if not self.store.is_cuda:
            raise RuntimeError('the image cache is not on a GPU: somi_augment_u8 has no CPU fallback')
        recs = (_lib.AugSample * len(plans))()
        for rec, plan in zip(recs, plans):
            self._fill_record(rec, plan)
        return torch.frombuffer(bytearray(bytes(recs)), dtype=torch.uint8).to(self.device)

    def launch(self, records, B, out=None, out_hw=None):
        """somi_augment_u8 on the current stream: uint8 (B, 3, h, w) RGB."""
        h, w = out_hw or (self.img_size, self.img_size)
        if out is None:
            out = torch.empty((B, 3, h, w), dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().somi_augment_u8(records.data_ptr(), B, h, w, FILL, out.data_ptr(),
                                                  torch.cuda.current_stream().cuda_stream), 'somi_augment_u8')
        return out

    def render(self, plans):
        """One launch: the uint8 (B, 3, h, w) RGB batch of `plans` (a list of the first element `plan` returns); the samples of
        one launch share their output size (always true for mosaic batches and for the batches of a rect loader)."""
        sizes = {p.out_hw for p in plans}
        if len(sizes) != 1:
            raise RuntimeError(f'samples of one batch have different letterbox shapes {sorted(sizes)} (rect batches must not be mixed)')
        return self.launch(self.upload(plans), len(plans), out_hw=sizes.pop())

    def __getitem__(self, index):
        plan, labels, shapes = self.plan(index)
        out = torch.zeros((len(labels), 6))
        if len(labels):
            out[:, 1:] = torch.from_numpy(labels)
        return self.render([plan])[0], out, self.img_files[self.indices[index]], shapes

    def batch(self, indices):
        """`collate_fn(batch)` (datasets.py:675-680) of `[self[i] for i in indices]`: imgs (B,3,s,s) uint8 on the device,
        targets (nt, 6) [sample, cls, x, y, w, h] (host tensor, as the loader yields it), paths, shapes."""
        plans, labs, shapes, paths = [], [], [], []
        for j, i in enumerate(indices):
            plan, labels, shp = self.plan(i)
            block = torch.zeros((len(labels), 6))
            if len(labels):
                block[:, 1:] = torch.from_numpy(labels)
            block[:, 0] = j
            plans.append(plan), labs.append(block), shapes.append(shp), paths.append(self.img_files[self.indices[i]])
        return self.render(plans), torch.cat(labs, 0), tuple(paths), tuple(shapes)
