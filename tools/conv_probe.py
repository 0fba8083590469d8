#!/usr/bin/env python3
"""Launch one conv shape repeatedly (for rocprofv3 --pmc passes).  usage: conv_probe.py B H Cin Cout k s [reps] [modulate]
PROBE_MODE=wgrad probes the weight-gradient kernel of the same layer instead."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402

B, H, Cin, Cout, k, s = (int(v) for v in sys.argv[1:7])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 5
mod = len(sys.argv) > 8
d = torch.device('cuda')
x = torch.randn(B, H, H, Cin, device=d)
w = torch.randn(Cout, k * k * Cin, device=d) * 0.05
b = torch.randn(Cout, device=d)
if os.environ.get('PROBE_MODE') == 'wgrad':
    Ho = ops.conv_out_size(H, k, s, k // 2)
    dy = torch.randn(B, Ho, Ho, Cout, device=d)
    for _ in range(reps):
        ops.conv2d_wgrad_nhwc(x, dy, kh=k, kw=k, stride=s, pad=k // 2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_wgrad_nhwc(x, dy, kh=k, kw=k, stride=s, pad=k // 2)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f'wgrad B{B} {H}x{H} {Cin}->{Cout} k{k}s{s}: {t*1e6:.1f} us  {2.0 * B * Ho * Ho * Cout * Cin * k * k / t / 1e12:.1f} TFLOP/s')
    sys.exit(0)
kw = dict(a_chan_scale=torch.rand(B, Cin, device=d), a_pix_scale=torch.rand(B, H, H, device=d)) if mod else {}
if os.environ.get('PROBE_STATS'):                            # BatchNorm partial sums from the epilogue (training forward)
    kw['bn_stats'] = {'pivot': torch.zeros(Cout, device=d)}
    b = None
act = 'none' if os.environ.get('PROBE_STATS') else 'silu'
for _ in range(reps):
    y = ops.conv2d_nhwc(x, w, b, kh=k, kw=k, stride=s, pad=k // 2, act=act, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    y = ops.conv2d_nhwc(x, w, b, kh=k, kw=k, stride=s, pad=k // 2, act=act, **kw)
e1.record()
torch.cuda.synchronize()
t = e0.elapsed_time(e1) / reps * 1e-3
fl = 2.0 * B * y.shape[1] * y.shape[2] * Cout * Cin * k * k
print(f'conv B{B} {H}x{H} {Cin}->{Cout} k{k}s{s} mod={mod}: {t*1e6:.1f} us  {fl/t/1e12:.1f} TFLOP/s')
