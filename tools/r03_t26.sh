mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_slab.py > gpurun_out/r03/t26.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t26.log
python bench.py --workload C4 --steps 500 --warmup 500 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['kernels']['pair_list'])"
python bench.py --workload C4T --steps 200 --warmup 200 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['kernels']['pair_list'])"
