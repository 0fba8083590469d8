# round 4, late: SEAM's squeeze without the normalised tensor (statistics pass + one pooling pass of gelu(u2) with the BatchNorm affine applied to the (B,C) means)
mkdir -p gpurun_out/r4
python -m pytest tests/test_train_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "sppf_seam or whole_model or bit_reproducible or full_width_well or attention_pieces" > gpurun_out/r4/t_ab9.log 2>&1 || { tail -n 30 gpurun_out/r4/t_ab9.log; exit 1; }
tail -n 2 gpurun_out/r4/t_ab9.log
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  $B > gpurun_out/r4/ab9_all_$r.json 2>> gpurun_out/r4/ab9.err
  SOMI_FUSE_POOL=0 $B > gpurun_out/r4/ab9_nofusepool_$r.json 2>> gpurun_out/r4/ab9.err
done
python - <<'PY'
import json
for f in ('all_1', 'nofusepool_1', 'all_2', 'nofusepool_2'):
    d = json.load(open('gpurun_out/r4/ab9_%s.json' % f))
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), d['roofline']['frac'])
PY
