#!/usr/bin/env python3
"""Per-shape table of the conv launches (forward / dgrad / wgrad) of one training step at batch 32, 640x640 (HIP events)."""
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402
from somi_amd.configs import somi_cfg, SOMI_ANCHORS, fill_state, synthetic_batch, HYP_VISDRONE  # noqa: E402
from somi_amd.model import Model  # noqa: E402
from somi_amd.train import TrainStep  # noqa: E402

B = 32
m = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).cuda()
tr = TrainStep(m, dict(HYP_VISDRONE), B, amp=os.environ.get('SOMI_AMP') or None)    # SOMI_AMP=bf16 / bf16x3: the opt-in reduced-precision step
imgs, tg = synthetic_batch(B, 640, seed=0)
imgs, tg = imgs.cuda(), tg.cuda()
tr.step(imgs, tg)
ops.PROFILE = prof = []
tr.step(imgs, tg)
torch.cuda.synchronize()
ops.PROFILE = None
agg = defaultdict(lambda: [0, 0.0, 0.0])
for name, fl, e0, e1, shp in prof:
    kind = {2: 'dgrad', 3: 'wgrad'}.get(shp[7], 'fwd' + ('/ps' if shp[7] == 1 else ''))
    a = agg[(kind,) + tuple(shp[:7])]
    a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(a[2] for a in agg.values())
print(f'total conv time {tot*1e3:.1f} ms, {sum(a[1] for a in agg.values())/tot/1e12:.1f} TFLOP/s')
for kind in ('fwd', 'fwd/ps', 'dgrad', 'wgrad'):
    t = sum(a[2] for k, a in agg.items() if k[0] == kind); f = sum(a[1] for k, a in agg.items() if k[0] == kind)
    if t: print(f'{kind:7s} {t*1e3:8.1f} ms  {f/t/1e12:6.1f} TFLOP/s')
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:40]:
    print(f'{k[0]:7s} B{k[1]} {k[2]}x{k[3]} {k[4]}->{k[5]} k{k[6]}s{k[7]}  n={a[0]:2d} {a[2]*1e3:7.2f} ms {a[1]/a[2]/1e12:6.1f} TF')
