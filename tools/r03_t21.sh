mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slab.py -x -q -m gpu -k "pair_lists or lazy or next_step or any_cell or lists_grow or surk or thermostat_radii or several_waves or grow_their or whole_number" > gpurun_out/r03/t21.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t21.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/i_$name.json 2> gpurun_out/r03/i_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/i_$name.json | head -2; }
b C4T --workload C4T --steps 200 --warmup 200
b M4 --workload M4 --steps 60 --warmup 60
b S40 --workload S40 --steps 200 --warmup 200
