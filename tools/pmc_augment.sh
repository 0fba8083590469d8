#!/bin/bash
# rocprofv3 passes for the input-pipeline kernel (counters only with --kernel-trace): kernel-trace stats, HBM bytes moved vs the
# algorithmic 6 B / pixel, instruction mix (VALU / SALU / LDS / VMEM) and busy cycles.
# usage: tools/pmc_augment.sh <outdir>
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python tools/kernel_bench.py augment > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python tools/kernel_bench.py augment > $out/pass$i.log 2>&1 || { tail -5 $out/pass$i.log; exit 1; }
done
{
  echo "# rocprofv3 --kernel-trace --stats -- python tools/kernel_bench.py augment"
  head -4 $(find $out/stats -name '*_kernel_stats.csv' | head -1)
  echo
  echo "# rocprofv3 --kernel-trace --pmc <one group per pass> -- python tools/kernel_bench.py augment  (mean per dispatch; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE to be doubled on gfx950)"
  python tools/pmc_summary.py $out augment_kernel
  echo
  grep augment_kernel $out/stats.log
} > $out/summary.txt
rm -rf $out/stats $out/pass1 $out/pass2 $out/pass3 $out/pass4
cat $out/summary.txt
