mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or any_cell or lists_grow or surk or thermostat_radii" > gpurun_out/r03/t14.log 2>&1
echo rc=$?; tail -4 gpurun_out/r03/t14.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/o_$name.json 2> gpurun_out/r03/o_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/o_$name.json | head -2; }
b C4 --workload C4 --steps 500 --warmup 500
b C4T --workload C4T --steps 200 --warmup 200
b C3T --workload C3T --steps 200 --warmup 200
b C4_driver --workload C4 --steps 20 --warmup 5
b C4_default --workload C4
