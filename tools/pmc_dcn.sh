#!/bin/bash
# rocprofv3 counter passes for the DCNv3 kernels (counters only with --kernel-trace): HBM bytes actually moved vs algorithmic bytes.
# usage: tools/pmc_dcn.sh <outdir>
set -e
out=$1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python tools/kernel_bench.py dcn > $out/pass$i.log 2>&1 || { tail -5 $out/pass$i.log; exit 1; }
done
grep dcnv3 $out/pass1.log
