#!/usr/bin/env python3
"""How far the host runs ahead of the GPU in the training step: wall time of enqueueing four steps vs the time until they have
executed (the step is GPU-bound when the first is well below the second).  usage: cpu_ahead_probe.py  (runs on the MI355X)"""
import os
import sys
import time

import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
from somi_amd.model import Model
from somi_amd.train import TrainStep
dev = torch.device('cuda')
model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).to(dev)
imgs, targets = synthetic_batch(32, 640, seed=1)
imgs, targets = imgs.to(dev), targets.to(dev)
tr = TrainStep(model, dict(HYP_VISDRONE), 32)
for _ in range(3): tr.step(imgs, targets)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(4): tr.step(imgs, targets)
t1 = time.time()
torch.cuda.synchronize()
t2 = time.time()
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    a = time.time()
    tr.step(imgs, targets)
    ts.append(time.time() - a)
torch.cuda.synchronize()
print('single step enqueue (queue empty at start):', [round(t * 1e3, 1) for t in ts], 'ms')
print(f'cpu enqueue per step {(t1-t0)/4*1e3:.1f} ms ; wall per step {(t2-t0)/4*1e3:.1f} ms')
