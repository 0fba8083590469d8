# round 4, late: SPPF backward as three chained 5x5 gathers (tests, then the kernel times under rocprofv3 and a 40-step run)
mkdir -p gpurun_out/r4
python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_stock_gpu.py -x -q -m gpu -k "sppf or whole_model or bit_reproducible or stock or full_width_well" > gpurun_out/r4/t_ab8.log 2>&1 || { tail -n 30 gpurun_out/r4/t_ab8.log; exit 1; }
tail -n 2 gpurun_out/r4/t_ab8.log
python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8 > gpurun_out/r4/ab8_all_1.json 2>> gpurun_out/r4/ab8.err
python bench.py --no-cpu-baseline --no-infer --steps 40 --warmup 8 > gpurun_out/r4/ab8_all_2.json 2>> gpurun_out/r4/ab8.err
python - <<'PY'
import json
for f in ('all_1', 'all_2'):
    d = json.load(open('gpurun_out/r4/ab8_%s.json' % f))
    print(f, d['ms_per_step'], 'conv ms', round(d['roofline']['conv_share_of_step'] * d['ms_per_step'], 1), d['roofline']['frac'], d['settle'])
PY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4/ab8_stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-infer --settle-max 0 > gpurun_out/r4/ab8_stats.json 2> gpurun_out/r4/ab8_stats.err
cp $(find gpurun_out/r4/ab8_stats -name '*_kernel_stats.csv' | head -1) gpurun_out/r4/ab8_kernel_stats.csv; rm -rf gpurun_out/r4/ab8_stats
grep "sppf\|pool_bwd_add\|add_kernel" gpurun_out/r4/ab8_kernel_stats.csv | cut -c1-200
