"""HBM rate of the BatchNorm / activation sweeps of the training step, kernel by kernel, at the step's own layer shapes (batch 32, 640^2).
Each timed call walks fresh buffers (a ring of tensors larger than the 256 MB Infinity Cache) so nothing is served from cache.

    python tools/bn_bw_probe.py            -> one JSON line per (kernel, shape): ms, algorithmic GB, TB/s
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'yolo-somi_amd'))
from somi_amd import ops  # noqa: E402

SHAPES = [(32, 320, 320, 64), (32, 160, 160, 128), (32, 160, 160, 64), (32, 80, 80, 256), (32, 80, 80, 128), (32, 40, 40, 512),
          (32, 40, 40, 256), (32, 20, 20, 1024), (32, 20, 20, 512)]


def timed(fn, ring, reps=12):
    for i in range(2):
        fn(i % ring)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i % ring)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device('cuda:0')
    for B, H, W, C in SHAPES:
        nbytes = B * H * W * C * 4
        ring = max(2, min(8, (1 << 30) // nbytes))
        xs = [torch.randn(B, H, W, C, device=dev) for _ in range(ring)]
        dzs = [torch.randn(B, H, W, C, device=dev) for _ in range(ring)]
        outs = [torch.empty(B, H, W, C, device=dev) for _ in range(ring)]
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        st = ops.bn_stats(xs[0], C, 0, g, b, 1e-3, 0.03, rm, rv)
        dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        rows = {
            'bn_stats (1 read)': (lambda i: ops.bn_stats(xs[i], C, 0, g, b, 1e-3, 0.03, rm, rv), 1),
            'chan_affine_act (1 read + 1 write)': (lambda i: ops.chan_affine_act(xs[i], C, 0, st[2], st[3], 'silu', 0, outs[i]), 2),
            'bn_act_backward (4 reads + 1 write)': (lambda i: ops.bn_act_backward(dzs[i], 0, xs[i], 0, C, *st, 'silu', 0, True, outs[i], 0, dg, db), 5),
            'add (2 reads + 1 write)': (lambda i: ops.add_(xs[i], 0, dzs[i], 0, C, outs[i], 0), 3),
        }
        for name, (fn, passes) in rows.items():
            ms = timed(fn, ring)
            print(json.dumps({'kernel': name, 'shape': f'N{B} {H}x{W} C{C}', 'ms': round(ms, 4), 'GB': round(passes * nbytes / 1e9, 3),
                              'TBps': round(passes * nbytes / ms / 1e9, 3)}), flush=True)
        del xs, dzs, outs
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
