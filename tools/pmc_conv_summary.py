#!/usr/bin/env python3
"""Summarise tools/pmc_conv_round4.sh: per conv kernel variant and launch geometry (kernel, grid, workgroup) the mean duration, the MFMA-pipe
busy share and the sustained clock, for the isolated probes and for the launches inside the training step.

  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x shader cycles of the dispatch), shader cycles = GRBM_GUI_ACTIVE / 8
              (rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back");
  GHz       = GRBM_GUI_ACTIVE / 8 / wall time of the dispatch;
  TFLOP/s at that clock if the pipe were always busy = 157.3 x GHz / 2.4.
usage: pmc_conv_summary.py <dir>"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
csv.field_size_limit(1 << 30)


def load(name):
    """-> {dispatch id: {'k': kernel, 'grid': .., 'wg': .., 'us': .., counters...}}"""
    rows = {}
    for f in glob.glob(f'{d}/{name}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if 'conv_igemm' not in k and 'conv_wgrad' not in k:
                continue
            e = rows.setdefault(int(r['Dispatch_Id']), {'k': re.sub(r'^void somi::', '', k)[:64], 'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']),
                                                        'us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    return rows


def table(title, a, b):
    print(f'## {title}')
    print(f'{"kernel":66s} {"grid/wg":>12s} {"n":>4s} {"avg_us":>8s} {"mfma_busy":>9s} {"GHz":>6s} {"wait_inst":>9s} {"vmem_lvl":>8s} {"L2 hit":>7s} {"L2 miss/launch":>14s}')
    groups = defaultdict(list)
    for e in a.values():
        groups[(e['k'], e['grid'], e['wg'])].append(e)
    gb = defaultdict(list)
    for e in b.values():
        gb[(e['k'], e['grid'], e['wg'])].append(e)
    tot_busy = tot_cyc = tot_t = tot_gui = 0.0
    for key, es in sorted(groups.items(), key=lambda kv: -sum(x['us'] for x in kv[1])):
        n = len(es)
        us = sum(x['us'] for x in es) / n
        gui = sum(x.get('GRBM_GUI_ACTIVE', 0) for x in es) / 8                  # shader cycles, summed over the group
        busy = sum(x.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) for x in es)
        wave = sum(x.get('SQ_WAVE_CYCLES', 0) for x in es)
        wait = sum(x.get('SQ_WAIT_INST_ANY', 0) for x in es)
        lvl = sum(x.get('SQ_INST_LEVEL_VMEM', 0) for x in es)
        t = sum(x['us'] for x in es) * 1e-6
        hb = gb.get(key, [])
        hit, miss = sum(x.get('TCC_HIT_sum', 0) for x in hb), sum(x.get('TCC_MISS_sum', 0) for x in hb)
        tot_busy += busy; tot_cyc += gui * 1024; tot_t += t; tot_gui += gui
        print(f'{key[0]:66s} {str(key[1] // key[2]) + "x" + str(key[2]):>12s} {n:4d} {us:8.1f} {busy / max(gui * 1024, 1):9.3f} {gui / max(t, 1e-12) / 1e9:6.2f} '
              f'{wait / max(wave, 1):9.3f} {lvl / max(wave, 1):8.2f} {hit / max(hit + miss, 1):7.3f} {miss / max(len(hb), 1):14.0f}')
    if tot_t:
        print(f'{"all conv launches above":66s} {"":>12s} {"":>4s} {"":>8s} {tot_busy / max(tot_cyc, 1):9.3f} {tot_gui / tot_t / 1e9:6.2f}')
    print()


print('# rocprofv3 --kernel-trace --pmc (counters only), tools/pmc_conv_round4.sh; pass A = SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES')
print('# SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE, pass B = TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE.  wait_inst = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES,')
print('# vmem_lvl = SQ_INST_LEVEL_VMEM / SQ_WAVE_CYCLES (vector-memory instructions in flight per wave).  Profiled passes run 2-5 % slower than the bench.')
for name, title in (('iso160', 'isolated: forward 32 x 160x160, 128 -> 128, 3x3 (tools/conv_probe.py, 12 + 12 launches)'),
                    ('iso80', 'isolated: forward 32 x 80x80, 256 -> 256, 3x3'),
                    ('step', 'inside the training step: python bench.py --steps 2 --warmup 1 (3 steps; every conv / wgrad launch, grouped by kernel and grid)')):
    table(title, load(name + '_a'), load(name + '_b'))
