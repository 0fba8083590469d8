# round-4 measurement, part B: smoke, the other bench lines, the north-star backbone figure, the DCNv3 counter passes
mkdir -p gpurun_out/r4
python __graft_entry__.py smoke > gpurun_out/r4/smoke.log 2>&1; tail -n 1 gpurun_out/r4/smoke.log
python bench.py --model somi --no-cpu-baseline > gpurun_out/r4/bench_somi.json 2> gpurun_out/r4/bench_b.err
python bench.py --mode infer --batch 128 > gpurun_out/r4/bench_infer128.json 2>> gpurun_out/r4/bench_b.err
python bench.py --size 1280 --batch 8 --nc 3 --no-cpu-baseline > gpurun_out/r4/bench_uavdt1280_b8.json 2>> gpurun_out/r4/bench_b.err
python bench.py --model yolov5s --batch 2 > gpurun_out/r4/bench_yolov5s_b2.json 2>> gpurun_out/r4/bench_b.err
python bench.py --amp bf16x3 --no-cpu-baseline > gpurun_out/r4/bench_amp_bf16x3.json 2>> gpurun_out/r4/bench_b.err
for f in bench_somi bench_infer128 bench_uavdt1280_b8 bench_yolov5s_b2 bench_amp_bf16x3; do python -c "import json;d=json.load(open('gpurun_out/r4/$f.json'));print('$f',d['value'],d['ms_per_step'],d['roofline']['frac'],d['roofline']['all_conv_tflops'],d.get('latency_ms_per_image'))"; done
python tools/kernel_bench.py dcn backbone > gpurun_out/r4/kernel_bench.jsonl 2> gpurun_out/r4/kernel_bench.err; tail -n 2 gpurun_out/r4/kernel_bench.jsonl | cut -c1-300
bash tools/pmc_dcn.sh > gpurun_out/r4/pmc_dcn.log 2>&1; echo pmc done; cp gpurun_out/pmc_dcn/summary.txt gpurun_out/r4/dcnv3_pmc.txt
