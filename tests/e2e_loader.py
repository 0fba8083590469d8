#!/usr/bin/env python3
"""End-to-end check of the device input pipeline's GEOMETRY (test infrastructure): train the small SOMI graph on batches that
come out of `DeviceImageCache` (mosaic, affine crop, mixup, saturation / value jitter, flips), then evaluate on the plain
rectangular validation batches of the same loader class.  The OpenCV arithmetic under the pipeline is restated, not pinned, so
bit-parity with the oracle cannot show that boxes and pixels still belong together after the warp - a detector that learns
from the augmented samples and then finds the objects in un-augmented images does.

  python tests/e2e_loader.py [steps]      -> one JSON line (profiles/r01_e2e_loader.json)
"""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def rectangles_dataset(n, S, nc, seed):
    """n BGR uint8 images (longer side S) with 1-4 class-coloured rectangles on dark noise + (k,5) [cls, x, y, w, h] labels."""
    rng = np.random.RandomState(seed)
    imgs, labels = [], []
    for i in range(n):
        h, w = (S, int(rng.randint(3 * S // 4, S + 1))) if i % 2 else (int(rng.randint(3 * S // 4, S + 1)), S)
        im = rng.randint(0, 40, (h, w, 3)).astype(np.uint8)
        lab = []
        for _ in range(int(rng.randint(1, 5))):
            c = int(rng.randint(0, nc))
            bw, bh = int(rng.randint(S // 8, S // 3)), int(rng.randint(S // 8, S // 3))
            x0, y0 = int(rng.randint(0, w - bw)), int(rng.randint(0, h - bh))
            im[y0:y0 + bh, x0:x0 + bw] = (200 - 50 * (c % 4), 60 + 90 * ((c // 3) % 3), 60 + 60 * (c % 3))      # B, G, R
            lab.append([c, (x0 + bw / 2) / w, (y0 + bh / 2) / h, bw / w, bh / h])
        imgs.append(im), labels.append(np.array(lab, dtype=np.float32))
    return imgs, labels


def main(steps=500, S=128, B=16, nc=4, mosaic=0.8, settle=True):
    from somi_amd import val as V
    from somi_amd.augment import DeviceImageCache, HYP_VISDRONE_AUGMENT
    from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg
    from somi_amd.model import Model
    from somi_amd.train import TrainStep
    cfg = somi_cfg(0.25, 0.33, nc=nc, anchors=SOMI_ANCHORS)
    model = fill_state(Model(cfg), 1).cuda()
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.zero_(); m.running_var.fill_(1.0); m.weight.fill_(1.0); m.bias.zero_()
    tr = TrainStep(model, dict(HYP_VISDRONE), B)
    for g_ in tr.optimizer.param_groups:
        g_['lr'] = 2e-3
    # the classes differ by colour: keep the hue, jitter the rest; one sample in five is a single letterboxed image (what
    # validation looks like), the others are mosaics
    hyp = dict(HYP_VISDRONE_AUGMENT, hsv_h=0.01, mosaic=mosaic)
    train_imgs, train_labels = rectangles_dataset(96, S, nc, 1)
    val_imgs, val_labels = rectangles_dataset(48, S, nc, 2)
    loader = DeviceImageCache(train_imgs, train_labels, S, hyp, augment=True)
    random.seed(0), np.random.seed(0)
    t0 = time.time()
    first = last = None
    for it in range(steps):
        if settle and it == steps - steps // 4:                   # settle: the last quarter runs at a quarter of the rate
            for g_ in tr.optimizer.param_groups:
                g_['lr'] = 5e-4
        imgs, targets, _, _ = loader.batch([random.randrange(len(loader)) for _ in range(B)])
        loss, _ = tr.step(imgs, targets.cuda())
        first = float(loss) if it == 0 else first
        last = float(loss)
    torch.cuda.synchronize()
    secs = time.time() - t0
    vl = DeviceImageCache(val_imgs, val_labels, S, hyp, augment=False, rect=True, batch_size=B, stride=32, pad=0.5)
    batches = [vl.batch(range(b0, min(b0 + B, len(vl)))) for b0 in range(0, len(vl), B)]
    mp, mr, m50, m, _ = V.run(model, batches, conf_thres=0.001, iou_thres=0.6)
    res = {'train_steps': steps, 'train_seconds_incl_loader': round(secs, 1), 'imgsz': S, 'batch': B, 'loss_first': round(first, 3),
           'loss_last': round(last, 3), 'val_batch_shapes': [list(map(int, b[0].shape[2:])) for b in batches],
           'val': {'P': mp, 'R': mr, 'mAP50': m50, 'mAP50_95': m}}
    print(json.dumps(res))
    return res


if __name__ == '__main__':
    main(*(int(v) for v in sys.argv[1:2]))
