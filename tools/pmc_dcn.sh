#!/bin/bash
# Counter passes over the DCNv3 kernels at N32 80x80 C256 (offsets ~ N(0, 0.7)): instruction mix, LDS conflicts, L2 requests, HBM bytes.
# One --pmc group per run (rocprofv3 with --kernel-trace only); summary -> gpurun_out/pmc_dcn/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_dcn; rm -rf $out; mkdir -p $out
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 5 180 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -- python3 tools/dcn_probe.py 32 80 0.7 3 > $out/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY'
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_dcn/p*/**/*_counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'dcnv3' in r['Kernel_Name']:
            res.setdefault(r['Kernel_Name'][:60], collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_dcn/summary.txt', 'w') as out:
    for k, c in res.items():
        if 'FETCH_SIZE' in c:
            c['fetch_MB(x2 gfx950 corr)'] = [2 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024 / 1e6]
        if 'WRITE_SIZE' in c:
            c['write_MB'] = [sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) * 1024 / 1e6]
        line = f"{k}\n    " + '  '.join(f"{n}={sum(v) / len(v):.4g}" for n, v in c.items())
        print(line); out.write(line + '\n')
PY
rm -rf $out/p[0-9]
