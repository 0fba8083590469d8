#!/usr/bin/env python3
"""Times the BatchNorm training kernels (statistics, affine + activation, backward reduce / apply) on single tensors of the
step's largest shapes and prints the achieved TB/s.  usage: bn_probe.py  (runs on the MI355X)"""
import os
import sys

import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'yolo-somi_amd'))
from somi_amd import ops
d = torch.device('cuda')
def timeit(fn, warm=3, iters=20):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for (B, H, C) in ((32, 160, 128), (32, 80, 256), (32, 40, 512)):
    y = torch.randn(B, H, H, C, device=d); dz = torch.randn_like(y); dy = torch.empty_like(y)
    g, b = torch.rand(C, device=d) + 0.5, torch.randn(C, device=d)
    mean, rstd, scale, shift = ops.bn_stats(y, C, 0, g, b, 1e-3, 0.03)
    dg, db = torch.zeros(C, device=d), torch.zeros(C, device=d)
    t_bwd = timeit(lambda: ops.bn_act_backward(dz, 0, y, 0, C, mean, rstd, scale, shift, 'silu', 0, True, dy, 0, dg, db))
    t_st = timeit(lambda: ops.bn_stats(y, C, 0, g, b, 1e-3, 0.03))
    z = torch.empty_like(y)
    t_af = timeit(lambda: ops.chan_affine_act(y, C, 0, scale, shift, 'silu', 0, z))
    nb = y.numel() * 4
    print(f'B{B} {H}x{H} C{C}: bn_act_backward {t_bwd:.1f} us ({5*nb/t_bwd/1e6:.2f} TB/s over 5 passes)  bn_stats {t_st:.1f} us ({nb/t_st/1e6:.2f} TB/s)  affine {t_af:.1f} us ({2*nb/t_af/1e6:.2f} TB/s)')
