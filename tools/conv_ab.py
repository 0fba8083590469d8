#!/usr/bin/env python3
"""Forward / dgrad / wgrad TFLOP/s over the representative conv shapes of the batch-32 640x640 training step (HIP events).
Run once per library build to compare kernels A/B:  SOMI_HIP_LIB=<other .so> python tools/conv_ab.py [tag]
Prints one JSON line per (shape, kind) and a FLOP-weighted summary (weights = launches of that shape per step)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd import ops  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get('SOMI_HIP_LIB', 'libsomi_hip.so'))
# (B, H, Cin, Cout, k, s, launches per step)  - the 3x3 bottleneck convs and the 1x1 cv1/cv2 layers of the yolov5l-SOMI graph
SHAPES = [(32, 160, 128, 128, 3, 1, 6), (32, 80, 256, 256, 3, 1, 12), (32, 40, 512, 512, 3, 1, 18), (32, 20, 1024, 1024, 3, 1, 6),
          (32, 160, 128, 128, 1, 1, 4), (32, 80, 256, 256, 1, 1, 8), (32, 80, 640, 256, 1, 1, 2), (32, 40, 512, 512, 1, 1, 8),
          (32, 160, 64, 128, 3, 2, 1), (32, 80, 256, 512, 3, 2, 2), (32, 320, 64, 64, 3, 1, 2)]
d = torch.device('cuda')
if os.environ.get('CONV_AB_N'):                                  # first N shapes only (kernel experiments)
    SHAPES = SHAPES[:int(os.environ['CONV_AB_N'])]
KINDS = os.environ.get('CONV_AB_KINDS', 'fwd,dgrad,wgrad').split(',')
ops.CONV_PREC = ops.PREC[os.environ.get('CONV_AB_PREC') or None]    # bf16 / bf16x3: the opt-in reduced-precision kernels


def t(fn, reps=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


tot = {'fwd': [0.0, 0.0], 'dgrad': [0.0, 0.0], 'wgrad': [0.0, 0.0]}
for B, H, Cin, Cout, k, s, n in SHAPES:
    p = k // 2
    x = torch.randn(B, H, H, Cin, device=d)
    w = torch.randn(Cout, k * k * Cin, device=d) * 0.05
    wt = torch.randn(Cin, k * k * Cout, device=d) * 0.05
    y = ops.conv2d_nhwc(x, w, None, kh=k, kw=k, stride=s, pad=p)
    dy = torch.randn_like(y)
    fl = 2.0 * B * y.shape[1] * y.shape[2] * Cout * Cin * k * k
    for kind, fn in (('fwd', lambda: ops.conv2d_nhwc(x, w, None, kh=k, kw=k, stride=s, pad=p, act='silu')),
                     ('dgrad', lambda: ops.conv2d_dgrad_nhwc(dy, wt, B=B, H=H, W=H, cin=Cin, kh=k, kw=k, stride=s, pad=p)),
                     ('wgrad', lambda: ops.conv2d_wgrad_nhwc(x, dy, kh=k, kw=k, stride=s, pad=p))):
        if kind not in KINDS:
            continue
        dt = t(fn)
        tot[kind][0] += fl * n
        tot[kind][1] += dt * n
        print(json.dumps({'lib': tag, 'kind': kind, 'shape': f'B{B} {H}x{H} {Cin}->{Cout} k{k}s{s}', 'us': round(dt * 1e6, 1),
                          'TFLOPs': round(fl / dt / 1e12, 1)}), flush=True)
    del x, w, wt, y, dy
print(json.dumps({'lib': tag, 'summary_TFLOPs': {k: round(v[0] / v[1] / 1e12, 1) for k, v in tot.items() if v[1]},
                  'all': round(sum(v[0] for v in tot.values()) / sum(v[1] for v in tot.values()) / 1e12, 1)}), flush=True)
