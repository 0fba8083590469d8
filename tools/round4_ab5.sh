# round 4, late: the spatial-attention weight-gradient kernel in 512-pixel chunks and CBAM's step C inside the BatchNorm backward.
# tests of the touched paths first, then an interleaved same-box A/B (40 timed steps each) and one kernel-stats pass
mkdir -p gpurun_out/r4
python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "spatial_attention_backward or step_c_inside or c2fcbam or attention or whole_model_train_step or bit_repro or isolated_at_320" > gpurun_out/r4/t_ab5.log 2>&1 || { tail -n 30 gpurun_out/r4/t_ab5.log; exit 1; }
tail -n 2 gpurun_out/r4/t_ab5.log
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  $B > gpurun_out/r4/ab5_all_$r.json 2>> gpurun_out/r4/ab5.err
  SOMI_CBAM_FUSED_BN=0 $B > gpurun_out/r4/ab5_nofusedbn_$r.json 2>> gpurun_out/r4/ab5.err
done
python - <<'PY'
import json
for f in ('all_1', 'nofusedbn_1', 'all_2', 'nofusedbn_2'):
    d = json.load(open('gpurun_out/r4/ab5_%s.json' % f))
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), d['roofline']['frac'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/ab5_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer > gpurun_out/r4/ab5_stats.json 2> gpurun_out/r4/ab5_stats.err
python3 tools/profile_summary.py gpurun_out/r4/ab5_stats > gpurun_out/r4/ab5_summary.txt 2>/dev/null || python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r4/ab5_stats/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open('gpurun_out/r4/ab5_summary.txt', 'w') as o:
    for r in rows[:70]:
        o.write('%-100s %6s %10.2f %9.1f\n' % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
PY
cp $(find gpurun_out/r4/ab5_stats -name '*_kernel_stats.csv' | head -1) gpurun_out/r4/ab5_kernel_stats.csv; rm -rf gpurun_out/r4/ab5_stats
grep -n "spatial_attn_bwd_weight\|cbam_b\|bn_act_bwd_.*pooled" gpurun_out/r4/ab5_summary.txt | cut -c1-150
