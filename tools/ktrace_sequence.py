#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace CSV into the ORDERED list of kernel launches of the last training step: "<short name> <us>" per line -
to see which kernels surround the small copies / fills (where in the step they are issued).  usage: ktrace_sequence.py <dir> <out.txt> [n_last]"""
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)
d, out = sys.argv[1], sys.argv[2]
n_last = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
rows = []
for f in glob.glob(f'{d}/**/*_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    n = n.replace('somi::', '').replace('at::native::', 'at::')
    return n[:70]


with open(out, 'w') as fh:
    prev_end = None
    for s, e, n in rows[-n_last:]:
        gap = 0.0 if prev_end is None else (s - prev_end) / 1e3
        fh.write(f'{short(n)} {(e - s) / 1e3:.1f} gap {gap:.1f}\n')
        prev_end = e
print(len(rows), 'launches in the trace;', min(n_last, len(rows)), 'written')
