#!/usr/bin/env python3
"""Soak run of the full-size training step on synthetic data: loss trend, finite checks, steady memory.
usage: soak_train.py [steps] [dcn]   (dcn: the bench graph with its two DCNv3 sites; reports the taps that left the backward's windows)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd.configs import HYP_VISDRONE, SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch  # noqa: E402
from somi_amd.model import Model  # noqa: E402
from somi_amd.train import TrainStep, one_cycle, warmup_lr  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dcn = len(sys.argv) > 2 and sys.argv[2] == 'dcn'
dev = torch.device('cuda')
model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS, dcn=dcn)), 1).to(dev)
hyp = dict(HYP_VISDRONE)
tr = TrainStep(model, hyp, 32)
lf = one_cycle(1, hyp['lrf'], 300)
batches = [synthetic_batch(32, 640, seed=50 + i) for i in range(4)]
batches = [(a.to(dev), b.to(dev)) for a, b in batches]
log, t0, max_far = [], time.time(), 0
for it in range(steps):
    warmup_lr(tr.optimizer, it, max(steps, 1), 0, lf, hyp, 32)            # the reference's warm-up ramp over the whole soak
    imgs, tg = batches[it % len(batches)]
    loss, items = tr.step(imgs, tg)
    if it % 10 == 0 or it == steps - 1:
        log.append({'step': it, 'loss': round(float(loss), 4), 'items': [round(float(v), 4) for v in items],
                    'mem_GB': round(torch.cuda.max_memory_allocated() / 1e9, 2)})
        assert torch.isfinite(loss).all(), log[-1]
        if dcn:
            from somi_amd import ops as _ops
            max_far = max(max_far, _ops.dcn_overflow_taps() or 0)               # the last site's call alone (the total below covers every call)
torch.cuda.synchronize()
finite = all(bool(torch.isfinite(b).all()) for b in tr.optimizer.flat_params)
out = {'steps': steps, 'seconds': round(time.time() - t0, 1), 'weights_finite': finite, 'log': log}
if dcn:
    from somi_amd import ops
    out['dcn_graph'] = True
    out['far_taps_last_backward'] = ops.dcn_overflow_taps()      # taps that went through fp32 atomics (beyond the window AND the near pass): 0 <=> bit-reproducible
    out['far_taps_max_over_run'] = max_far
    out['far_taps_total_all_sites_all_steps'] = ops.dcn_overflow_taps(total=True)   # device counter summed over every windowed backward of the run
print(json.dumps(out))
