#!/bin/bash
# rocprofv3 counter passes for one conv shape (counters only with --kernel-trace, as the pool requires).
# usage: tools/pmc_conv.sh <outdir> <conv_probe args...>
set -e
out=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p $out
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
P3="FETCH_SIZE"
P4="WRITE_SIZE"
P5="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $out/pass$i -- python tools/conv_probe.py "$@" > $out/pass$i.log 2>&1 || { tail -5 $out/pass$i.log; exit 1; }
done
tail -1 $out/pass1.log
