mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q -m gpu -k "wall_crossing or lazy or next_step or pressure or slabs_match or dilute or degenerate or thermostat" > gpurun_out/r03/t25.log 2>&1
echo rc=$?; tail -4 gpurun_out/r03/t25.log
python bench.py --workload C4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['kernels']['collect'], d['kernels']['finalize'])"
