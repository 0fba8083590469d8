#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (counter_collection.csv + kernel_trace.csv) per kernel: mean counter value per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

out = sys.argv[1]
kfilter = sys.argv[2] if len(sys.argv) > 2 else 'conv_igemm'
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(f'{out}/pass*/**/*_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if kfilter in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for f in sorted(glob.glob(f'{out}/pass1/**/*_kernel_trace.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if kfilter in r['Kernel_Name']:
            dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, cs in acc.items():
    print(k, ' dispatches:', len(next(iter(cs.values()))), ' avg_us(pass1):', round(sum(dur[k]) / max(len(dur[k]), 1), 1))
    for c, v in sorted(cs.items()):
        print(f'   {c:32s} {sum(v) / len(v):16.1f}')
