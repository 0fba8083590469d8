# round 4, late: ODConv - input gradient added in place by the dgrad epilogue, squeeze gradient folded into the producing BatchNorm backward, squeeze average taken by the producing BatchNorm + SiLU pass.
mkdir -p gpurun_out/r4
python -m pytest tests/test_train_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "whole_model or odconv or bit_reproducible or train_step_with_optimizer or full_width_well or accumulation" > gpurun_out/r4/t_ab7.log 2>&1 || { tail -n 30 gpurun_out/r4/t_ab7.log; exit 1; }
tail -n 2 gpurun_out/r4/t_ab7.log
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  $B > gpurun_out/r4/ab7_all_$r.json 2>> gpurun_out/r4/ab7.err
  SOMI_ODCONV_INPLACE=0 $B > gpurun_out/r4/ab7_noinplace_$r.json 2>> gpurun_out/r4/ab7.err
done
python - <<'PY'
import json
for f in ('all_1', 'noinplace_1', 'all_2', 'noinplace_2'):
    d = json.load(open('gpurun_out/r4/ab7_%s.json' % f))
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), d['roofline']['frac'], d['settle'])
PY
