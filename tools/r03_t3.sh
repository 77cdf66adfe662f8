set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or sort_interval or thermostat_radii" > gpurun_out/r03/t3.log 2>&1 || { tail -40 gpurun_out/r03/t3.log; exit 1; }
tail -3 gpurun_out/r03/t3.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 200 --warmup 200 --no-cpu-baseline "$@" > gpurun_out/r03/y_$name.json 2> gpurun_out/r03/y_$name.err; echo "== $name"; python tools/bench_summary.py gpurun_out/r03/y_$name.json | head -2; grep "aztot: lists recorded" gpurun_out/r03/y_$name.err | tail -1; }
b C4 --workload C4
b C4T --workload C4T
b C4T_ph1 --workload C4T --debug 1
b C4T_s40 --workload C4T --skin 0.40
b C4_noskin --workload C4 --skin -1
b C4X --workload C4X
b C3T --workload C3T
