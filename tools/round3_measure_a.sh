# round-3 measurement, part A (run through gpurun from the repo root): kernel-trace stats + HBM-traffic passes of the training and inference bench
mkdir -p gpurun_out/r3
bash tools/profile_round.sh gpurun_out/prof_r03 > gpurun_out/r3/profile_round.log 2>&1; tail -n 1 gpurun_out/r3/profile_round.log
cp gpurun_out/prof_r03/traffic.json profiles/traffic.json
python bench.py > gpurun_out/r3/bench_line.json 2> gpurun_out/r3/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r3/bench_line.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_tflops'], d['roofline']['traffic'], d['cpu_baseline']['value'], d['cpu_baseline']['iterations'])
PY
