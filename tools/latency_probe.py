#!/usr/bin/env python3
"""Inference latency at small batch (the reference's utils/get_FPS.py protocol: warm-up, then timed forwards with a sync around
each): eager launches vs hipGraph replay (somi_amd.graph.GraphedModel).  usage: latency_probe.py [batch ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'yolo-somi_amd'))
import torch  # noqa: E402

from somi_amd.configs import SOMI_ANCHORS, fill_state, somi_cfg  # noqa: E402
from somi_amd.graph import GraphedModel  # noqa: E402
from somi_amd.model import Model  # noqa: E402


def fps(fn, x, warm=20, iters=100):
    for _ in range(warm):
        fn(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(iters):
        fn(x)
        torch.cuda.synchronize()                                 # get_FPS.py:92-101 syncs around every forward
    return (time.time() - t0) / iters


def main():
    dev = torch.device('cuda')
    model = fill_state(Model(somi_cfg(1.0, 1.0, anchors=SOMI_ANCHORS)), 1).to(dev).eval()
    for B in [int(v) for v in sys.argv[1:]] or [1, 4, 16]:
        x = torch.randint(0, 256, (B, 3, 640, 640), dtype=torch.uint8, device=dev)
        with torch.no_grad():
            te = fps(lambda t: model(t), x)
        g = GraphedModel(model, x)
        tg = fps(g, x)
        with torch.no_grad():
            z0 = model(x)[0]
        z1 = g(x)[0]
        torch.cuda.synchronize()
        print(json.dumps({'batch': B, 'eager_ms': round(te * 1e3, 2), 'graph_ms': round(tg * 1e3, 2), 'eager_fps': round(B / te, 1),
                          'graph_fps': round(B / tg, 1), 'max_abs_diff': float((z0 - z1).abs().max())}), flush=True)


if __name__ == '__main__':
    main()
