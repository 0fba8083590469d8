# interleaved same-box A/B, 40 timed steps each: does GPU time removed from the DCNv3 backward / the CBAM pooled pass show in the step?
mkdir -p gpurun_out/r4
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  $B > gpurun_out/r4/ab2_all_$r.json 2>> gpurun_out/r4/ab2.err
  SOMI_DCN_SLAB=1 $B > gpurun_out/r4/ab2_dcnslab_$r.json 2>> gpurun_out/r4/ab2.err
  SOMI_BN_POOLED=0 SOMI_FUSE_POOL=0 SOMI_AMAX_BY_VALUE=0 $B > gpurun_out/r4/ab2_r3cbam_$r.json 2>> gpurun_out/r4/ab2.err
done
python - <<'PY'
import json
for f in ('all_1', 'dcnslab_1', 'r3cbam_1', 'all_2', 'dcnslab_2', 'r3cbam_2'):
    d = json.load(open('gpurun_out/r4/ab2_%s.json' % f))
    k = d['roofline_dcnv3']['kernels']
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), 'dcn bwd us', k['dcnv3_bwd_kernel']['avg_launch_us'])
PY
