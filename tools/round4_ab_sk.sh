# hybrid stream-K: parity of the conv family, then interleaved same-box step times (40 steps) for the pure and hybrid schedules
mkdir -p gpurun_out/r4
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py -q -m gpu -x -k "conv or whole_model or bit_reproducible or stream or amp" > gpurun_out/r4/t_sk.log 2>&1; tail -3 gpurun_out/r4/t_sk.log
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  SOMI_SK_HYBRID=0 $B > gpurun_out/r4/absk_pure_$r.json 2>> gpurun_out/r4/absk.err
  $B > gpurun_out/r4/absk_hyb4_$r.json 2>> gpurun_out/r4/absk.err
  SOMI_SK_REM_MIN_KT=8 $B > gpurun_out/r4/absk_hyb8_$r.json 2>> gpurun_out/r4/absk.err
  SOMI_SK_REM_MIN_KT=1 $B > gpurun_out/r4/absk_hyb1_$r.json 2>> gpurun_out/r4/absk.err
done
python - <<'PY'
import json
for r in (1, 2):
    for f in ('pure', 'hyb4', 'hyb8', 'hyb1'):
        d = json.load(open('gpurun_out/r4/absk_%s_%d.json' % (f, r)))
        print(f, r, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), 'all conv TF', d['roofline']['all_conv_tflops'], 'dominant', d['roofline']['frac'])
PY
