# same-box A/B of the round-4 CBAM fusions (run through gpurun): step time with each one switched off
mkdir -p gpurun_out/r4
B="python bench.py --no-cpu-baseline --no-infer --steps 20 --warmup 6"
$B > gpurun_out/r4/ab_all.json 2> gpurun_out/r4/ab.err
SOMI_FUSE_POOL=0 $B > gpurun_out/r4/ab_nofusepool.json 2>> gpurun_out/r4/ab.err
SOMI_AMAX_BY_VALUE=0 $B > gpurun_out/r4/ab_noamax.json 2>> gpurun_out/r4/ab.err
SOMI_BN_POOLED=0 $B > gpurun_out/r4/ab_nobnpooled.json 2>> gpurun_out/r4/ab.err
SOMI_DCN_SLAB=1 $B > gpurun_out/r4/ab_dcnslab.json 2>> gpurun_out/r4/ab.err
$B > gpurun_out/r4/ab_all2.json 2>> gpurun_out/r4/ab.err
python - <<'PY'
import json
for f in ('ab_all', 'ab_nofusepool', 'ab_noamax', 'ab_nobnpooled', 'ab_dcnslab', 'ab_all2'):
    d = json.load(open('gpurun_out/r4/%s.json' % f))
    print(f, d['ms_per_step'], d['roofline']['all_conv_tflops'], d['roofline']['conv_share_of_step'])
PY
