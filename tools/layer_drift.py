#!/usr/bin/env python3
"""Per-layer drift probe: train-mode forward of the SOMI graph on the HIP path and on the fp32 CPU oracle, both against the fp64 CPU
oracle, layer by layer (max |difference| / max |fp64 value| of every layer's output).  Shows WHERE the HIP path leaves the fp32 CPU
path's error band.  Usage: python tools/layer_drift.py [--size 1280] [--batch 2] [--nc 3] [--width 1.0] [--eval]"""
import argparse
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn as nn  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=1280)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--nc', type=int, default=3)
    ap.add_argument('--width', type=float, default=1.0)
    ap.add_argument('--depth', type=float, default=1.0)
    ap.add_argument('--seed', type=int, default=6)
    ap.add_argument('--eval', action='store_true')
    ap.add_argument('--dcn', action='store_true')
    a = ap.parse_args()
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.testing import SOMI_ANCHORS, fill_state, somi_cfg, synthetic_batch
    from somi_amd import blocks as B
    from somi_amd import ops
    from somi_amd.model import Model
    torch.set_num_threads(16)
    cfg = somi_cfg(a.width, a.depth, nc=a.nc, anchors=SOMI_ANCHORS, dcn=a.dcn)
    ref = fill_state(OModel(cfg), a.seed)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref64 = copy.deepcopy(ref).double()
    imgs, _ = synthetic_batch(a.batch, a.size, nc=a.nc, seed=14)
    for m in (ref, ref64, mine):
        m.train(not a.eval)
    mine = mine.cuda()

    def oracle_layers(model, x):
        outs, y = [], []
        with torch.no_grad():
            for m in model.model:
                if m.f != -1:
                    x = y[m.f] if isinstance(m.f, int) else [x if j == -1 else y[j] for j in m.f]
                if isinstance(m, nn.Upsample):
                    m.recompute_scale_factor = False
                x = m(x.contiguous() if torch.is_tensor(x) else x)
                y.append(x if m.i in model.save else None)
                outs.append(x)
        return outs
    o64 = oracle_layers(ref64, imgs.double() / 255)
    o32 = oracle_layers(ref, imgs.float() / 255)
    with torch.no_grad():
        act = B.Act(ops.image_to_nhwc4(imgs.cuda().contiguous(), scale=1.0 / 255.0), 0, 3)
        y = []
        print(f'{"layer":>5} {"type":<16} {"HIP vs fp64":>12} {"fp32 CPU vs fp64":>17} {"ratio":>7}')
        for m in mine.model:
            if m.f != -1:
                act = y[m.f] if isinstance(m.f, int) else [act if j == -1 else y[j] for j in m.f]
            act = m(act)
            y.append(act if m.i in mine.save else None)
            w64, w32 = o64[m.i], o32[m.i]
            if isinstance(act, B.Act):
                t = act.t[..., act.coff:act.coff + act.c]
                if act.up:
                    continue                                     # a view flag; the consumer expands it
                got = [t.permute(0, 3, 1, 2).cpu().double()]
                w64, w32 = [w64], [w32]
            else:
                raws = act[1] if isinstance(act, tuple) else act
                got = [r.cpu().double() for r in raws]
                w64, w32 = (list(w[1]) if isinstance(w, tuple) else list(w) for w in (w64, w32))
            em = max((g - w).abs().max().item() / (w.abs().max().item() + 1e-30) for g, w in zip(got, w64))
            eo = max((v.double() - w).abs().max().item() / (w.abs().max().item() + 1e-30) for v, w in zip(w32, w64))
            print(f'{m.i:>5} {m.type:<16} {em:12.2e} {eo:17.2e} {em / max(eo, 1e-30):7.1f}')


if __name__ == '__main__':
    main()
