mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or any_cell_population or lists_grow" > gpurun_out/r03/t6.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t6.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/w_$name.json 2> gpurun_out/r03/w_$name.err; echo "== $name"; python tools/bench_summary.py gpurun_out/r03/w_$name.json > gpurun_out/r03/w_$name.txt; head -2 gpurun_out/r03/w_$name.txt; }
b C4T --workload C4T --steps 200 --warmup 200
b C4T_ph1 --workload C4T --steps 10 --warmup 20 --debug 1
b C4T_ph2 --workload C4T --steps 10 --warmup 20 --debug 2
b C4T_ph3 --workload C4T --steps 10 --warmup 20 --debug 3
