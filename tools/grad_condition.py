#!/usr/bin/env python3
"""Where do full-size parameter gradients sit against their own conditioning?  (diagnostic behind tests/test_train_gpu.py's
conditioning-aware bound; uses the CPU oracle as the checker, so it is a tool, not product code.)

For every parameter: error of the HIP gradient and of the fp32 CPU oracle's gradient against the fp64 oracle, in units of
2^-24 * S with S = sum |terms| of that gradient measured in the fp64 pass (oracle.somi_ref.testing.AbsTermSums), beside the plain
relative error and the condition number S / |g|.
usage: grad_condition.py [size=1280] [nc=3] [batch=2] [dcn=1] [seed=6]"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402

from oracle.somi_ref import Model as OModel  # noqa: E402
from oracle.somi_ref.loss import ComputeLoss as OLoss  # noqa: E402
from oracle.somi_ref.testing import (SOMI_ANCHORS, AbsTermSums, conditioned_errors, fill_state, noise_scaled_errors, somi_cfg,  # noqa: E402
                                     synthetic_batch, HYP_VISDRONE)
from somi_amd.loss import ComputeLoss  # noqa: E402
from somi_amd.model import Model  # noqa: E402

size, nc, B, dcn, seed = (int(v) for v in (sys.argv[1:] + ['1280', '3', '2', '1', '6'][len(sys.argv) - 1:])[:5])
torch.set_num_threads(16)
cfg = somi_cfg(1.0, 1.0, nc=nc, anchors=SOMI_ANCHORS, dcn=bool(dcn))
ref = fill_state(OModel(cfg), seed)
mine = Model(cfg)
mine.load_state_dict(ref.state_dict())
ref.hyp = mine.hyp = dict(HYP_VISDRONE)
ref64 = copy.deepcopy(ref).double()
imgs, targets = synthetic_batch(B, size, nc=nc, seed=14)
mine = mine.cuda().train()
lm, _ = ComputeLoss(mine)(mine(imgs.cuda()), targets.cuda())
lm.backward()
torch.cuda.synchronize()
ref64.train()
with AbsTermSums(ref64) as cond:
    l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
    l64.backward()
ref.train()
l32, _ = OLoss(ref)(ref(imgs.float() / 255), targets)
l32.backward()
print(f'loss hip {lm.item():.6f} fp32 {l32.item():.6f} fp64 {l64.item():.6f}')
g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
ch = {t[0]: t for t in conditioned_errors([(n, p.grad) for n, p in mine.named_parameters() if p.grad is not None], g64, cond.sums)}
cc = {t[0]: t for t in conditioned_errors([(n, p.grad) for n, p in ref.named_parameters() if p.grad is not None], g64, cond.sums)}
print(f'{len(ch)} parameters with a measured S; {len(g64) - len(ch)} without')


def c_req(t):
    return t[1]


rows = sorted(ch, key=lambda n: -ch[n][2])
print('worst 25 by relative error of the HIP gradient:   name | rel hip | rel fp32cpu | cond S/|g| | c_req hip | c_req cpu')
for n in rows[:25]:
    h, c = ch[n], cc[n]
    print(f'{n:58s} {h[2]:9.2e} {c[2]:9.2e} {h[3]:9.1e} {h[1]:10.1f} {c[1]:10.1f}')
uh = sorted((c_req(ch[n]), n) for n in ch)
uc = sorted((c_req(cc[n]), n) for n in cc)
print('largest required c (units of 2^-24 S beyond the 1e-3 relative bar): HIP', [(round(v), n) for v, n in uh[-6:]])
print('                                                                   CPU', [(round(v), n) for v, n in uc[-6:]])
import statistics  # noqa: E402
print('relative error population: HIP median %.2e q90 %.2e max %.2e | fp32 CPU median %.2e q90 %.2e max %.2e' % (
    statistics.median(t[2] for t in ch.values()), sorted(t[2] for t in ch.values())[int(0.9 * len(ch))], max(t[2] for t in ch.values()),
    statistics.median(t[2] for t in cc.values()), sorted(t[2] for t in cc.values())[int(0.9 * len(cc))], max(t[2] for t in cc.values())))

# candidate bar: e_hip(p) <= K * (u + c0 * 2^-24 * cond(p)), u = the fp32 CPU path's own error level on well-conditioned parameters
import torch as _t  # noqa: E402
well = sorted(t[2] for t in cc.values() if t[3] < 100)
u50, u90, u99 = well[len(well) // 2], well[int(0.9 * len(well))], well[min(len(well) - 1, int(0.99 * len(well)))]
print(f'fp32 CPU error on well-conditioned parameters (cond < 100, {len(well)} of them): median {u50:.2e} q90 {u90:.2e} q99 {u99:.2e} max {well[-1]:.2e}')
for c0 in (16, 64, 256):
    ratios = sorted(((ch[n][2] / (u90 + c0 * 2.0 ** -24 * ch[n][3]), n) for n in ch), reverse=True)
    rc = sorted(((cc[n][2] / (u90 + c0 * 2.0 ** -24 * cc[n][3]), n) for n in cc), reverse=True)
    print(f'c0 {c0}: worst e / (u90 + c0 eps cond): HIP', [(round(v, 2), n) for v, n in ratios[:5]], '| CPU', [(round(v, 2), n) for v, n in rc[:3]])

# noise-scaled view: r = max|dg| / max(|g| + Q), Q = root-sum-square of the gradient's terms
rss = cond.rss
nh = {t[0]: t for t in noise_scaled_errors([(n, p.grad) for n, p in mine.named_parameters() if p.grad is not None], g64, rss)}
nc = {t[0]: t for t in noise_scaled_errors([(n, p.grad) for n, p in ref.named_parameters() if p.grad is not None], g64, rss)}
rh, rc = sorted(t[1] for t in nh.values()), sorted(t[1] for t in nc.values())
q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]  # noqa: E731
print(f'noise-scaled error r: HIP median {q(rh, .5):.2e} q90 {q(rh, .9):.2e} q99 {q(rh, .99):.2e} max {rh[-1]:.2e} | fp32 CPU median {q(rc, .5):.2e} '
      f'q90 {q(rc, .9):.2e} q99 {q(rc, .99):.2e} max {rc[-1]:.2e}')
u = q(rc, .9)
worst = sorted(((nh[n][1] / u, n, nh[n][1], nc[n][1], nh[n][2]) for n in nh), reverse=True)[:15]
print('worst r_hip / u90(cpu):')
for w in worst:
    print(f'  {w[1]:58s} x{w[0]:6.2f}   r_hip {w[2]:.2e} r_cpu {w[3]:.2e} rel_hip {w[4]:.2e}')
