bash tools/profile_round.sh gpurun_out/prof_r02 > gpurun_out/r2_profile_round.log 2>&1; tail -n 1 gpurun_out/r2_profile_round.log
bash tools/profile_dcn.sh > gpurun_out/r2_prof_dcn.log 2>&1
bash tools/pmc_dcn.sh > gpurun_out/r2_pmc_dcn.log 2>&1; echo pmc done
cp gpurun_out/prof_r02/traffic.json profiles/traffic.json
python bench.py > gpurun_out/r2_bench_line.json 2> gpurun_out/r2_bench.err
python bench.py --size 1280 --batch 8 --nc 3 --no-cpu-baseline > gpurun_out/r2_bench_uavdt1280_b8.json 2>> gpurun_out/r2_bench.err
python bench.py --model yolov5s --batch 2 > gpurun_out/r2_bench_yolov5s_b2.json 2>> gpurun_out/r2_bench.err
python bench.py --model somi --no-cpu-baseline > gpurun_out/r2_bench_somi.json 2>> gpurun_out/r2_bench.err
python bench.py --mode infer --batch 128 --no-cpu-baseline > gpurun_out/r2_bench_infer128.json 2>> gpurun_out/r2_bench.err
for f in r2_bench_line r2_bench_uavdt1280_b8 r2_bench_yolov5s_b2 r2_bench_somi r2_bench_infer128; do python -c "import json,sys;d=json.load(open('gpurun_out/$f.json'));print('$f',d['value'],d['ms_per_step'],d.get('roofline',{}).get('frac'),d.get('roofline',{}).get('traffic'),(d.get('roofline_dcnv3') or {}).get('achieved'),(d.get('roofline_dcnv3') or {}).get('share_of_step'))"; done
