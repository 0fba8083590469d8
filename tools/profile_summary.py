#!/usr/bin/env python3
"""Condense a tools/profile_bench.sh output directory into a small text summary for profiles/ (per-kernel time table,
per-kernel mean HBM traffic and MFMA utilisation).  usage: profile_summary.py <dir> > profiles/<name>.txt"""
import csv
import glob
import sys
from collections import defaultdict

d = sys.argv[1]


def first(pattern):
    fs = sorted(glob.glob(pattern, recursive=True))
    return fs[0] if fs else None


infer = len(sys.argv) <= 2                                          # called without a traffic file for the inference profile
print('# rocprofv3 --kernel-trace --stats : python bench.py ' + ('--mode infer --batch 128 --steps 5 --warmup 2 --no-cpu-baseline' if infer else
                                                              '--steps 3 --warmup 1 --no-cpu-baseline --no-infer'))
f = first(f'{d}/stats/**/*_kernel_stats.csv')
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(f'# total kernel time {tot / 1e6:.1f} ms over ' + ('7 steps (2 warm-up + 5)' if infer else '4 steps (1 warm-up + 3)'))
print(f'{"kernel":100s} {"calls":>6s} {"total_ms":>10s} {"avg_us":>10s} {"pct":>6s}')
for r in rows[:60]:
    print(f'{r["Name"][:100]:100s} {r["Calls"]:>6s} {float(r["TotalDurationNs"]) / 1e6:10.2f} {float(r["AverageNs"]) / 1e3:10.1f} '
          f'{float(r["Percentage"]):6.1f}')
print()
acc = defaultdict(lambda: defaultdict(list))
for sub in ('fetch', 'write', 'sq'):
    for f in glob.glob(f'{d}/{sub}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name'][:80]][r['Counter_Name']].append(float(r['Counter_Value']))
            if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':            # wall time of the same dispatches: sustained clock = GUI cycles / 8 XCDs / time
                acc[r['Kernel_Name'][:80]]['_ns'].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
print('# PMC passes (separate runs, counters only): mean per dispatch.  FETCH_SIZE / WRITE_SIZE are in KiB;')
print('# hbm_read_MB applies the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read).')
print('# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE / 8); GHz = GRBM_GUI_ACTIVE / 8 / wall time of the dispatches (sq pass).')
print(f'{"kernel":82s} {"n":>5s} {"hbm_read_MB":>12s} {"hbm_write_MB":>12s} {"mfma_busy":>10s} {"GHz":>6s}')
order = sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', [0])) - sum(kv[1].get('FETCH_SIZE', [0])))
for k, c in order[:16]:
    n = len(c.get('FETCH_SIZE', c.get('WRITE_SIZE', [0])))
    rd = 2 * sum(c.get('FETCH_SIZE', [0])) / max(n, 1) * 1024 / 1e6
    wr = sum(c.get('WRITE_SIZE', [0])) / max(len(c.get('WRITE_SIZE', [0])), 1) * 1024 / 1e6
    busy = ghz = ''
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in c and sum(c['GRBM_GUI_ACTIVE']) > 0:
        busy = f'{sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) / 8 * 1024):.3f}'
        ghz = f'{sum(c["GRBM_GUI_ACTIVE"]) / 8 / max(sum(c["_ns"]), 1.0):.2f}'
    print(f'{k:82s} {n:5d} {rd:12.1f} {wr:12.1f} {busy:>10s} {ghz:>6s}')

if len(sys.argv) > 2:
    import json
    kern = {}
    for k, c in acc.items():
        if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c:
            rd = 2 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024
            wr = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024
            kern[k] = {'hbm_bytes_per_launch': round(rd + wr), 'read_bytes': round(rd), 'write_bytes': round(wr),
                       'launches_sampled': len(c['FETCH_SIZE'])}
    json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer`; '
                         'FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half of a wide coalesced read)',
               'kernels': kern}, open(sys.argv[2], 'w'), indent=1)
