#!/bin/bash
# Round profile of bench.py on the GPU box: kernel-trace stats + separate PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE).
# usage: tools/profile_bench.sh <outdir>   (run through gpurun; counters are never combined with other trace domains)
set -e
out=${1:-gpurun_out/prof}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer > $out/bench_stats.json 2> $out/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > $out/bench_write.json 2> $out/write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $out/sq -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer > $out/bench_sq.json 2> $out/sq.err
python tools/profile_summary.py $out $out/traffic.json > $out/summary.txt
cp $(find $out/stats -name '*_kernel_stats.csv' | head -1) $out/kernel_stats.csv
# the raw traces are tens of MiB for a training step: keep the condensed files only
rm -rf $out/stats $out/fetch $out/write $out/sq
echo done
