set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or sort_interval or thermostat_radii or family_kernels" > gpurun_out/r03/t4.log 2>&1 || { tail -40 gpurun_out/r03/t4.log; exit 1; }
tail -3 gpurun_out/r03/t4.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 200 --warmup 200 --no-cpu-baseline "$@" > gpurun_out/r03/z_$name.json 2> gpurun_out/r03/z_$name.err; echo "== $name"; python tools/bench_summary.py gpurun_out/r03/z_$name.json > gpurun_out/r03/z_$name.txt; head -2 gpurun_out/r03/z_$name.txt; grep "aztot: lists recorded" gpurun_out/r03/z_$name.err | tail -1; }
b C3T --workload C3T
b C3 --workload C3
b C3T_noskin --workload C3T --skin -1
b C4T_ph2 --workload C4T --debug 2
b M4 --workload M4 --steps 50 --warmup 50
b S40 --workload S40
b S4 --workload S4
b B3 --workload B3 --steps 100 --warmup 100
