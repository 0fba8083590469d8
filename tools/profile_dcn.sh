#!/bin/bash
# rocprofv3 kernel trace of the DCNv3 kernel bench (forward, windowed backward A+B+C, direct backward); summary -> gpurun_out/prof_dcn/
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rm -rf gpurun_out/prof_dcn && mkdir -p gpurun_out/prof_dcn
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dcn/raw -- python3 tools/kernel_bench.py dcn > gpurun_out/prof_dcn/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_dcn/raw/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open('gpurun_out/prof_dcn/summary.txt', 'w') as out:
    for r in rows:
        if 'dcnv3' in r['Name']:
            line = f"{r['Name'][:90]:<90} calls {r['Calls']:>5} avg_us {float(r['AverageNs'])/1e3:>10.1f} min_us {float(r['MinNs'])/1e3:>10.1f} max_us {float(r['MaxNs'])/1e3:>10.1f}"
            print(line); out.write(line + '\n')
PY
rm -rf gpurun_out/prof_dcn/raw
