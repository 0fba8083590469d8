# round 4, late: the weight gradients' split reduce on a side stream.  Training tests (gradients, bit reproducibility, the 2-rank rehearsal) first,
# then an interleaved same-box A/B, 40 timed steps each
mkdir -p gpurun_out/r4
python -m pytest tests/test_train_gpu.py -x -q -m gpu -k "whole_model or bit_reproducible or train_step or packed_conv_masters or rehearsal or sync_bn or accumulation or end_to_end or two_ranks or amp_training or multi_scale" > gpurun_out/r4/t_ab6.log 2>&1 || { tail -n 30 gpurun_out/r4/t_ab6.log; exit 1; }
tail -n 2 gpurun_out/r4/t_ab6.log
B="python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8"
for r in 1 2; do
  $B > gpurun_out/r4/ab6_all_$r.json 2>> gpurun_out/r4/ab6.err
  SOMI_SIDE_REDUCE=0 $B > gpurun_out/r4/ab6_noside_$r.json 2>> gpurun_out/r4/ab6.err
done
python - <<'PY'
import json
for f in ('all_1', 'noside_1', 'all_2', 'noside_2'):
    d = json.load(open('gpurun_out/r4/ab6_%s.json' % f))
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), d['roofline']['frac'])
PY
