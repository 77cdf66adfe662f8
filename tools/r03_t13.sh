mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or thermostat_radii or surk or potential_families or any_cell or energies_only" > gpurun_out/r03/t13.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t13.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/p_$name.json 2> gpurun_out/r03/p_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/p_$name.json > gpurun_out/r03/p_$name.txt; head -2 gpurun_out/r03/p_$name.txt; grep "aztot: lists recorded" gpurun_out/r03/p_$name.err | tail -1; }
b S40_w1 --workload S40 --steps 200 --warmup 200 --split 1
b S40_w2 --workload S40 --steps 200 --warmup 200 --split 2
b S40_w4 --workload S40 --steps 200 --warmup 200 --split 4
b S4 --workload S4 --steps 500 --warmup 500
