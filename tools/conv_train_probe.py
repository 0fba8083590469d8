#!/usr/bin/env python3
"""Times forward / dgrad / wgrad of one conv shape on the MI355X.  usage: conv_train_probe.py B H Cin Cout k s"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402

B, H, Cin, Cout, k, s = (int(v) for v in sys.argv[1:7])
d = torch.device('cuda')
p = k // 2
x = torch.randn(B, H, H, Cin, device=d)
w = torch.randn(Cout, k * k * Cin, device=d) * 0.05
wt = torch.randn(Cin, k * k * Cout, device=d) * 0.05
y = ops.conv2d_nhwc(x, w, None, kh=k, kw=k, stride=s, pad=p)
dy = torch.randn_like(y)
fl = 2.0 * B * y.shape[1] * y.shape[2] * Cout * Cin * k * k


def t(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for name, fn in (('fwd', lambda: ops.conv2d_nhwc(x, w, None, kh=k, kw=k, stride=s, pad=p, act='silu')),
                 ('dgrad', lambda: ops.conv2d_dgrad_nhwc(dy, wt, B=B, H=H, W=H, cin=Cin, kh=k, kw=k, stride=s, pad=p)),
                 ('wgrad', lambda: ops.conv2d_wgrad_nhwc(x, dy, kh=k, kw=k, stride=s, pad=p))):
    dt = t(fn)
    print(f'{name:6s} B{B} {H}x{H} {Cin}->{Cout} k{k}s{s}: {dt * 1e6:9.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s', flush=True)
