#!/bin/bash
# Round profile on the GPU box (run through gpurun): rocprofv3 kernel-trace stats of the default training bench and of the inference bench at
# batch 128 (BASELINE configs[4]), plus separate PMC passes (FETCH_SIZE / WRITE_SIZE, never combined with other trace domains) for the HBM
# traffic of the training step's kernels.  usage: tools/profile_round.sh <outdir>
set -e
out=${1:-gpurun_out/prof_r02}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
T="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $T > $out/bench_stats.json 2> $out/stats.err
echo "train stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0 > $out/bench_fetch.json 2> $out/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0 > $out/bench_write.json 2> $out/write.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $out/sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0 > $out/bench_sq.json 2> $out/sq.err
echo "train pmc done"
python3 tools/profile_summary.py $out $out/traffic.json > $out/train_summary.txt
cp $(find $out/stats -name '*_kernel_stats.csv' | head -1) $out/train_kernel_stats.csv
rm -rf $out/stats $out/fetch $out/write $out/sq
I="python3 bench.py --mode infer --batch 128 --steps 5 --warmup 2 --no-cpu-baseline --settle-max 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- $I > $out/bench_infer_stats.json 2> $out/infer.err
python3 tools/profile_summary.py $out > $out/infer_summary.txt
cp $(find $out/stats -name '*_kernel_stats.csv' | head -1) $out/infer_kernel_stats.csv
rm -rf $out/stats
echo done
