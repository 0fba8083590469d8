#!/bin/bash
# counter passes of the LDS-DMA conv kernel ALONE and INSIDE the training step, same counters, same box.
# MFMA-pipe busy share and the clock the chip sustains (GRBM_GUI_ACTIVE / 8 / wall time) for both, so that the in-step loss against the
# isolated kernels (126-131 -> 115 TFLOP/s) can be split into clock and everything else.  Counters only with --kernel-trace (pool rule).
# usage: tools/pmc_conv_round4.sh <outdir>
set -e
out=${1:-gpurun_out/r4/conv_pmc}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
PA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"
PB="TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
run() {  # name, pmc, command...
  local name=$1; local pmc=$2; shift 2
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $out/$name -- "$@" > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
  echo "$name done"
}
# isolated: 160x160 128->128 3x3 and 80x80 256->256 3x3 at batch 32 (forward), 12 launches each
run iso160_a "$PA" python3 tools/conv_probe.py 32 160 128 128 3 1 12
run iso160_b "$PB" python3 tools/conv_probe.py 32 160 128 128 3 1 12
run iso80_a "$PA" python3 tools/conv_probe.py 32 80 256 256 3 1 12
run iso80_b "$PB" python3 tools/conv_probe.py 32 80 256 256 3 1 12
# in the step: the default training bench, 1 warm-up + 2 timed steps
T="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0"
run step_a "$PA" $T
run step_b "$PB" $T
python3 tools/pmc_conv_summary.py $out > $out/summary.txt
find $out -name '*.csv' -size +20M -delete 2>/dev/null || true
tail -40 $out/summary.txt
