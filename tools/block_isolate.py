#!/usr/bin/env python3
"""Capture-and-isolate: where is the error of a full-size gradient GENERATED?

One whole-graph fp64 oracle pass (forward + loss + backward) records, for every top-level layer, its inputs and the gradient of its
outputs.  Every layer that owns parameters is then run ALONE on those captured tensors rounded to fp32 - the fp64 oracle block, the fp32
CPU oracle block and the HIP block see bit-identical inputs, output gradients and weights - and each parameter gradient, the input
gradient and the output are compared with the fp64 block's: what a block shows here is error it generates itself; what it shows only in
the whole-graph pass arrived from upstream.  Per parameter the tool prints the isolated relative error of both fp32 paths, the error
beyond 1e-3 in units of one fp32 rounding of the gradient's own terms (c_req, AbsTermSums) and the whole-graph relative errors beside it.

The CPU oracle is the checker here: a tool, not product code.
usage: block_isolate.py [--size 1280] [--nc 3] [--batch 2] [--dcn 1] [--seed 6] [--layers 30,36,...] [--out FILE]"""
import argparse
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402


from tests.isolate import capture, hip_alone, oracle_alone  # noqa: E402


def rel(a, b):
    b = b.double()
    return (a.double() - b).abs().max().item() / (b.abs().max().item() + 1e-300)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=1280)
    ap.add_argument('--nc', type=int, default=3)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--dcn', type=int, default=1)
    ap.add_argument('--seed', type=int, default=6)
    ap.add_argument('--width', type=float, default=1.0)
    ap.add_argument('--depth', type=float, default=1.0)
    ap.add_argument('--layers', default='')
    ap.add_argument('--out', default='')
    ap.add_argument('--threads', type=int, default=16)
    ap.add_argument('--gamma', type=float, default=1.0, help='scale every BatchNorm weight (0.25: the well-conditioned fill of the tests)')
    ap.add_argument('--no-hip', action='store_true', help='dry run of the oracle side on a box without a GPU')
    a = ap.parse_args()
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, conditioned_errors, fill_state, somi_cfg, synthetic_batch, HYP_VISDRONE
    from somi_amd.loss import ComputeLoss
    from somi_amd.model import Model
    torch.set_num_threads(a.threads)
    fh = open(a.out, 'w') if a.out else None

    def say(*s):
        line = ' '.join(str(v) for v in s)
        print(line, flush=True)
        if fh:
            fh.write(line + '\n')
            fh.flush()
    t0 = time.time()
    cfg = somi_cfg(a.width, a.depth, nc=a.nc, anchors=SOMI_ANCHORS, dcn=bool(a.dcn))
    ref = fill_state(OModel(cfg), a.seed)
    if a.gamma != 1.0:
        with torch.no_grad():
            for m_ in ref.modules():
                if isinstance(m_, torch.nn.BatchNorm2d):
                    m_.weight.mul_(a.gamma)
    mine = Model(cfg)
    mine.load_state_dict(ref.state_dict())
    ref.hyp = mine.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double()
    imgs, targets = synthetic_batch(a.batch, a.size, nc=a.nc, seed=14)
    ref.train(), ref64.train()
    full_hip = {}
    if not a.no_hip:                                             # whole-graph passes: HIP, fp64 (captured), fp32 CPU
        mine = mine.cuda().train()
        lm, _ = ComputeLoss(mine)(mine(imgs.cuda()), targets.cuda())
        lm.backward()
        torch.cuda.synchronize()
        full_hip = {n: p.grad.detach().cpu() for n, p in mine.named_parameters() if p.grad is not None}
        say(f'# whole-graph HIP pass done, loss {lm.item():.6f}  [{time.time() - t0:.0f}s]')

    def run64():
        l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
        l64.backward()
        say(f'# whole-graph fp64 pass done, loss {l64.item():.6f}  [{time.time() - t0:.0f}s]')
    cap = capture(ref64, run64)
    full64 = {n: p.grad.detach().clone() for n, p in ref64.named_parameters() if p.grad is not None}
    l32, _ = OLoss(ref)(ref(imgs.float() / 255), targets)
    l32.backward()
    full32 = {n: p.grad.detach().clone() for n, p in ref.named_parameters() if p.grad is not None}
    say(f'# whole-graph fp32 CPU pass done, loss {l32.item():.6f}  [{time.time() - t0:.0f}s]')
    want = [int(v) for v in a.layers.split(',') if v] or [m.i for m in ref64.model if any(True for _ in m.parameters())]
    say('# layer type | param | isolated: rel_hip rel_cpu32 c_req_hip c_req_cpu32 cond | whole graph: rel_hip rel_cpu32')
    for i in want:
        rec = cap[i]
        if any(d is None for d in rec['dy']):
            say(f'{i:3d} {ref64.model[i].type}: no output gradient captured, skipped')
            continue
        o64, dx64, g64, sums = oracle_alone(ref64.model[i], rec, torch.float64, with_sums=True)
        o32, dx32, g32, _ = oracle_alone(ref.model[i], rec, torch.float32)
        oh, dxh, gh = (o32, dx32, g32) if a.no_hip else hip_alone(mine.model[i], rec)
        typ = ref64.model[i].type
        e_out = (max(rel(x, y) for x, y in zip(oh, o64)), max(rel(x, y) for x, y in zip(o32, o64)))
        pairs = [(h, c, w) for h, c, w in zip(dxh, dx32, dx64) if w is not None] if dxh else []
        e_dx = (max((rel(h, w) for h, c, w in pairs), default=0.0), max((rel(c, w) for h, c, w in pairs), default=0.0))
        say(f'{i:3d} {typ:<15} OUTPUT rel_hip {e_out[0]:.2e} rel_cpu32 {e_out[1]:.2e} | DX rel_hip {e_dx[0]:.2e} rel_cpu32 {e_dx[1]:.2e}  [{time.time() - t0:.0f}s]')
        ch = {t[0]: t for t in conditioned_errors(list(gh.items()), g64, sums)}
        cc = {t[0]: t for t in conditioned_errors(list(g32.items()), g64, sums)}
        for n in g64:
            full = f'model.{i}.{n}'
            rh = rel(gh[n], g64[n]) if n in gh else float('nan')
            rc = rel(g32[n], g64[n])
            crh = ch[n][1] if n in ch else float('nan')
            crc = cc[n][1] if n in cc else float('nan')
            cond = ch[n][3] if n in ch else float('nan')
            fhp = rel(full_hip[full], full64[full]) if full in full_hip else float('nan')
            fcp = rel(full32[full], full64[full])
            flag = ' <<<' if (rh > 1e-3 and crh > 16) else ''
            say(f'{i:3d} {typ:<15} {n:<46} {rh:9.2e} {rc:9.2e} {crh:9.1f} {crc:9.1f} {cond:8.1e} | {fhp:9.2e} {fcp:9.2e}{flag}')
        del o64, dx64, g64, o32, dx32, g32, oh, dxh, gh
        torch.cuda.empty_cache()
    say(f'# done [{time.time() - t0:.0f}s]')


if __name__ == '__main__':
    main()
