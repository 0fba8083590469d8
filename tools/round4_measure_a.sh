# round-4 measurement, part A (run through gpurun from the repo root): the default bench line (with its configs[4] leg), kernel-trace stats +
# HBM-traffic + MFMA-busy / clock passes of the training bench, the inference profile, DCNv3 backward A/B (coloured vs slab form)
mkdir -p gpurun_out/r4
python bench.py > gpurun_out/r4/bench_line.json 2> gpurun_out/r4/bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_line.json'))
print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['all_conv_tflops'], d.get('configs4'), d['roofline_dcnv3']['kernels'])
PY
python tools/dcn_bwd_probe.py > gpurun_out/r4/dcn_bwd_probe.jsonl 2>&1; SOMI_DCN_SLAB=1 python tools/dcn_bwd_probe.py >> gpurun_out/r4/dcn_bwd_probe.jsonl 2>&1; cat gpurun_out/r4/dcn_bwd_probe.jsonl
bash tools/profile_round.sh gpurun_out/prof_r04 > gpurun_out/r4/profile_round.log 2>&1; tail -n 1 gpurun_out/r4/profile_round.log
