#!/bin/bash
# L2-miss traffic (FETCH_SIZE with the gfx950 x2 correction, TCC hit / miss) and time of the 3x3 conv kernel at the two dominant shapes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_conv_fetch; rm -rf $out; mkdir -p $out
for shape in "32 80 128 128 3 1" "32 160 128 128 3 1"; do
  for mode in 0; do
    tag=$(echo $shape | tr ' ' '_')_m$mode
    SOMI_CONV_2D=$mode rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f_$tag -- python3 tools/conv_probe.py $shape 4 > $out/f_$tag.log 2>&1
    SOMI_CONV_2D=$mode rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/t_$tag -- python3 tools/conv_probe.py $shape 4 > $out/t_$tag.log 2>&1
    SOMI_CONV_2D=$mode python3 tools/conv_probe.py $shape 10 > $out/time_$tag.log 2>&1
  done
done
python3 - <<'PY'
import csv, glob, collections
res = collections.OrderedDict()
for f in sorted(glob.glob('gpurun_out/pmc_conv_fetch/*/**/*_counter_collection.csv', recursive=True)):
    tag = f.split('/')[2]
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' in r['Kernel_Name']:
            res.setdefault(tag[2:], collections.defaultdict(list))[r['Counter_Name']].append(float(r['Counter_Value']))
with open('gpurun_out/pmc_conv_fetch/summary.txt', 'w') as out:
    for tag, c in res.items():
        t = open(f'gpurun_out/pmc_conv_fetch/time_{tag}.log').read().strip().splitlines()[-1]
        line = (f"{tag:28s} fetch_MB(x2 corr) {2 * sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) * 1024 / 1e6:8.1f}  "
                f"tcc_hit {sum(c['TCC_HIT_sum']) / max(len(c['TCC_HIT_sum']), 1) / 1e6:7.2f}M  tcc_miss {sum(c['TCC_MISS_sum']) / max(len(c['TCC_MISS_sum']), 1) / 1e6:7.2f}M  | {t}")
        print(line); out.write(line + '\n')
PY
rm -rf $out/f_* $out/t_*
