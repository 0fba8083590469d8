#!/usr/bin/env python3
"""Loader throughput, device pipeline vs the CPU restatement (test infrastructure: imports the oracle as the CPU baseline).

  python tests/loader_rate.py > profiles/r01_loader_rate.json      (on the GPU box)

CPU leg: `oracle.somi_ref.augment.CachedDataset` (numpy restatement of the reference's per-sample work with the OpenCV
arithmetic restated in numpy - slower than cv2's SIMD code, so this is a *port* baseline, not the reference's own speed),
one thread, 16 samples of a 640 px mosaic.  Device leg: `DeviceImageCache.batch` end to end - host planning (random draws,
label boxes), record upload and the kernel - for batches of 32, synchronised per batch.
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

from oracle.somi_ref.augment import CachedDataset  # noqa: E402
from somi_amd.augment import DeviceImageCache, HYP_VISDRONE_AUGMENT  # noqa: E402


def main(S=640, nimg=128, B=32):
    rng = np.random.RandomState(0)
    imgs = [rng.randint(0, 256, (360, 640, 3)).astype(np.uint8) for _ in range(nimg)]
    labs = []
    for _ in range(nimg):
        k = int(rng.randint(20, 90))
        wh = rng.uniform(0.01, 0.1, (k, 2))
        labs.append(np.concatenate((rng.randint(0, 10, (k, 1)), rng.uniform(0, 1, (k, 2)) * (1 - wh) + wh / 2, wh), 1).astype(np.float32))
    hyp = dict(HYP_VISDRONE_AUGMENT)
    torch.set_num_threads(1)
    cpu = CachedDataset(imgs, labs, S, hyp)
    random.seed(0), np.random.seed(0)
    cpu[0]
    t0 = time.perf_counter()
    for i in range(16):
        cpu[i]
    cpu_rate = 16 / (time.perf_counter() - t0)
    dev = DeviceImageCache(imgs, labs, S, hyp)
    random.seed(0), np.random.seed(0)
    dev.batch(range(B))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(10):
        dev.batch([(it * B + j) % nimg for j in range(B)])
        torch.cuda.synchronize()
    dev_rate = 10 * B / (time.perf_counter() - t0)
    print(json.dumps({'workload': f'training samples (mosaic + affine crop + mixup 0.2 + HSV + flip) at {S} px from {nimg} cached 360x640 images',
                      'device_images_per_s_end_to_end': round(dev_rate, 1), 'device_batch': B,
                      'cpu_port_images_per_s_one_thread': round(cpu_rate, 2), 'cpu_kind': 'port (numpy restatement incl. OpenCV arithmetic)',
                      'training_step_consumes_images_per_s': 97.0}))


if __name__ == '__main__':
    main()
