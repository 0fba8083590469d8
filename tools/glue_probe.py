#!/usr/bin/env python3
"""Which host-side torch ops (fills, copies, adds, cats ...) the training step still launches, grouped by the somi_amd source
line that issued them.  usage: glue_probe.py [batch]  (runs on the MI355X)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch  # noqa: E402
from somi_amd.model import Model  # noqa: E402
from somi_amd.train import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda')
model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).to(dev)
imgs, targets = synthetic_batch(B, 640, seed=1)
imgs, targets = imgs.to(dev), targets.to(dev)
tr = TrainStep(model, dict(HYP_VISDRONE), B)
for _ in range(2):
    tr.step(imgs, targets)
torch.cuda.synchronize()
count = collections.Counter()


def spy(owner, name):
    orig = getattr(owner, name)

    def wrapped(*a, **k):
        f = sys._getframe(1)
        while f is not None and 'somi_amd' not in f.f_code.co_filename:
            f = f.f_back
        where = f'{os.path.basename(f.f_code.co_filename)}:{f.f_lineno} {f.f_code.co_name}' if f else '?'
        count[(name, where)] += 1
        return orig(*a, **k)
    setattr(owner, name, wrapped)


for owner, names in ((torch, ['zeros', 'zeros_like', 'full', 'cat', 'ones', 'empty_like']),
                     (torch.Tensor, ['add_', 'copy_', 'clone', 'item', 'contiguous', 'zero_', '__iadd__', '__float__', 'tolist'])):
    for n in names:
        spy(owner, n)
tr.step(imgs, targets)
torch.cuda.synchronize()
for (name, where), n in count.most_common(70):
    print(f'{n:5d}  {name:12s} {where}')
