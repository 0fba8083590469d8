#!/usr/bin/env python3
"""How heavy-tailed is a full-size gradient's response to ONE fp32 rounding?  (run once, summary committed under profiles/.)

The 1280x1280 nc-3 DCN graph with fill_state's weights is chaotic: both fp32 paths sit percent-level away from the fp64 oracle.  This tool
measures the graph's own condition with NOTHING of the implementation under test in the calibration: the fp32 CPU ORACLE is run again
`--twins` times with every weight moved by one fp32 rounding (x (1 +- 2^-23), random signs), and the distance of each twin from the unperturbed
CPU run is expressed, per parameter, in the units the test uses: r = max |dg| / max (|g64| + Q).  The same is done for the HIP path.  Printed:
the distribution over parameters and twins of  s(p) / u90  (u90 = the CPU path's 90th-percentile distance from fp64 - the test's unit),
i.e. how many u90 a single rounding moves one parameter - the number a fixed per-parameter factor has to cover - and where the HIP path's
actual distance from fp64 sits in it.  The CPU oracle is the checker: a tool, not product code.
usage: twin_distribution.py [--size 1280] [--nc 3] [--batch 2] [--twins 16] [--out FILE]"""
import argparse
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'yolo-somi_amd')):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', type=int, default=1280)
    ap.add_argument('--nc', type=int, default=3)
    ap.add_argument('--batch', type=int, default=2)
    ap.add_argument('--seed', type=int, default=6)
    ap.add_argument('--twins', type=int, default=16)
    ap.add_argument('--out', default='')
    ap.add_argument('--threads', type=int, default=16)
    a = ap.parse_args()
    from oracle.somi_ref import Model as OModel
    from oracle.somi_ref.loss import ComputeLoss as OLoss
    from oracle.somi_ref.testing import SOMI_ANCHORS, AbsTermSums, fill_state, noise_scaled_errors, somi_cfg, synthetic_batch, HYP_VISDRONE
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
    cfg = somi_cfg(1.0, 1.0, nc=a.nc, anchors=SOMI_ANCHORS, dcn=True)
    ref = fill_state(OModel(cfg), a.seed)
    state0 = copy.deepcopy(ref.state_dict())
    ref.hyp = dict(HYP_VISDRONE)
    ref64 = copy.deepcopy(ref).double()
    imgs, targets = synthetic_batch(a.batch, a.size, nc=a.nc, seed=14)
    ref64.train()
    with AbsTermSums(ref64) as cond:
        l64, _ = OLoss(ref64)(ref64(imgs.double() / 255), targets.double())
        l64.backward()
    g64 = {n: p.grad for n, p in ref64.named_parameters() if p.grad is not None}
    rss = cond.rss
    say(f'# {a.size}x{a.size} nc {a.nc} batch {a.batch} DCN graph, fill_state seed {a.seed}; fp64 pass done [{time.time() - t0:.0f}s]')

    def perturb(model, t):
        g = torch.Generator().manual_seed(99 + t)
        with torch.no_grad():
            for p in model.parameters():
                p.mul_(1.0 + (torch.randint(0, 2, p.shape, generator=g).to(p.dtype) * 2 - 1) * 2.0 ** -23)

    def cpu_run(t=None):
        m = OModel(cfg)
        m.load_state_dict(state0)
        if t is not None:
            perturb(m, t)
        m.hyp = dict(HYP_VISDRONE)
        m.train()
        OLoss(m)(m(imgs.float() / 255), targets)[0].backward()
        return [(n, p.grad.detach()) for n, p in m.named_parameters() if p.grad is not None]

    def hip_run(t=None):
        m = Model(cfg)
        m.load_state_dict(state0)
        if t is not None:
            perturb(m, t)
        m.hyp = dict(HYP_VISDRONE)
        m = m.cuda().train()
        ComputeLoss(m)(m(imgs.cuda()), targets.cuda())[0].backward()
        torch.cuda.synchronize()
        out = [(n, p.grad.detach().cpu()) for n, p in m.named_parameters() if p.grad is not None]
        del m
        torch.cuda.empty_cache()
        return out

    def q(v, f):
        v = sorted(v)
        return v[min(len(v) - 1, int(f * len(v)))]
    if os.environ.get('SOMI_TOOL_NO_HIP') == '1':                # dry run of the tool's own logic on a box without a GPU
        hip_run = cpu_run
    cpu0, hip0 = cpu_run(), hip_run()
    r_cpu = {t[0]: t[1] for t in noise_scaled_errors(cpu0, g64, rss)}
    r_hip = {t[0]: t[1] for t in noise_scaled_errors(hip0, g64, rss)}
    names = list(r_cpu)
    u90 = q(list(r_cpu.values()), 0.9)
    say(f'distance from fp64 (r units): fp32 CPU median {q(list(r_cpu.values()), .5):.2e} q90 {u90:.2e} max {max(r_cpu.values()):.2e} | '
        f'HIP median {q(list(r_hip.values()), .5):.2e} q90 {q(list(r_hip.values()), .9):.2e} max {max(r_hip.values()):.2e}  [{time.time() - t0:.0f}s]')
    s_cpu, s_hip = {n: [] for n in names}, {n: [] for n in names}
    for t in range(a.twins):
        for n, r, _ in noise_scaled_errors(cpu0, g64, rss, against=dict(cpu_run(t))):
            s_cpu[n].append(r)
        for n, r, _ in noise_scaled_errors(hip0, g64, rss, against=dict(hip_run(t))):
            s_hip[n].append(r)
        say(f'twin {t}: fp32 CPU vs its one-rounding twin: median {q([v[-1] for v in s_cpu.values()], .5):.2e} q90 {q([v[-1] for v in s_cpu.values()], .9):.2e} '
            f'max {max(v[-1] for v in s_cpu.values()):.2e} (x{max(v[-1] for v in s_cpu.values()) / u90:.1f} u90) | HIP vs its twin: median '
            f'{q([v[-1] for v in s_hip.values()], .5):.2e} q90 {q([v[-1] for v in s_hip.values()], .9):.2e} max {max(v[-1] for v in s_hip.values()):.2e}'
            f'  [{time.time() - t0:.0f}s]')
    # per twin: the largest single-parameter response in units of u90 - what ONE rounding can do to ONE parameter of this graph
    worst_cpu = sorted(max(s_cpu[n][t] for n in names) / u90 for t in range(a.twins))
    worst_hip = sorted(max(s_hip[n][t] for n in names) / u90 for t in range(a.twins))
    say(f'largest single-parameter response per twin, in u90: fp32 CPU twins {[round(v, 1) for v in worst_cpu]}')
    say(f'                                                    HIP twins      {[round(v, 1) for v in worst_hip]}')
    allc = sorted(v / u90 for n in names for v in s_cpu[n])
    allh = sorted(v / u90 for n in names for v in s_hip[n])
    for lab, v in (('fp32 CPU twins', allc), ('HIP twins', allh)):
        say(f'{lab}: all (parameter, twin) responses / u90: median {q(v, .5):.2f} q90 {q(v, .9):.2f} q99 {q(v, .99):.2f} q99.9 {q(v, .999):.2f} max {v[-1]:.2f}; '
            f'share beyond 6: {sum(x > 6 for x in v) / len(v):.2e}, beyond 12: {sum(x > 12 for x in v) / len(v):.2e}')
    # the test's per-parameter quantity: r_hip(p) / max(u90, r_cpu(p)) - and the same with the CPU twins' largest response as the yardstick
    ratio = sorted(((r_hip[n] / max(u90, r_cpu[n]), n) for n in names), reverse=True)
    say('HIP distance from fp64 per parameter / max(u90, the CPU path\'s own distance): worst ' +
        ', '.join(f'{n} x{v:.1f}' for v, n in ratio[:6]) + f'; beyond 6: {sum(v > 6 for v, _ in ratio)}, beyond 12: {sum(v > 12 for v, _ in ratio)} of {len(ratio)}')
    ratio2 = sorted(((r_hip[n] / max(u90, r_cpu[n], max(s_cpu[n])), n) for n in names), reverse=True)
    say('... / max(u90, CPU distance, largest response of that parameter over the CPU twins): worst ' +
        ', '.join(f'{n} x{v:.1f}' for v, n in ratio2[:6]))
    say(f'# done [{time.time() - t0:.0f}s]')


if __name__ == '__main__':
    main()
