# quick check of a kernel change: the list tests, then C4 / C4T steady-state lines (gpurun_out/r03b/)
mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or any_cell or lists_grow or whole_number or surk" > gpurun_out/r03b/t.log 2>&1
rc=$?; echo rc=$rc; tail -5 gpurun_out/r03b/t.log
[ $rc -eq 0 ] || exit $rc
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03b/$name.json 2> gpurun_out/r03b/$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03b/$name.json | head -2; grep "aztot: lists recorded" gpurun_out/r03b/$name.err | tail -1; }
b C4 --workload C4 --steps 500 --warmup 500
b C4T --workload C4T --steps 300 --warmup 300
