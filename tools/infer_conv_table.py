#!/usr/bin/env python3
"""Per-shape table of the conv launches of one inference forward (HIP events).  usage: infer_conv_table.py [batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402
from somi_amd.configs import SOMI_ANCHORS, fill_state, somi_cfg  # noqa: E402
from somi_amd.model import Model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda')
model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).to(dev).eval()
x = torch.randint(0, 256, (B, 3, 640, 640), dtype=torch.uint8, device=dev)
with torch.no_grad():
    for _ in range(3):
        model(x)
    ops.PROFILE = prof = []
    model(x)
    torch.cuda.synchronize()
    ops.PROFILE = None
by = {}
for name, fl, e0, e1, shp in prof:
    v = by.setdefault(shp, [0, 0.0, 0.0, name])
    v[0] += 1; v[1] += fl; v[2] += e0.elapsed_time(e1) * 1e-3
tot = sum(v[2] for v in by.values())
print(f'batch {B}: {len(prof)} conv launches, {tot * 1e3:.2f} ms, {sum(v[1] for v in by.values()) / tot / 1e12:.1f} TFLOP/s')
for shp, (n, fl, s, name) in sorted(by.items(), key=lambda kv: -kv[1][2])[:25]:
    Bq, H, W, cin, cout, k, st, ps = shp
    print(f'B{Bq} {H}x{W} {cin}->{cout} k{k}s{st}{" ps" if ps else ""}  n={n:2d}  {s * 1e3:7.3f} ms  {s / n * 1e6:7.1f} us/launch  {fl / s / 1e12:6.1f} TF  {name.split("<")[1]}')
