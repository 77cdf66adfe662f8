mkdir -p gpurun_out/r03b
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or any_cell or lists_grow or whole_number or surk or restart or case_study or sort_interval" > gpurun_out/r03b/t2.log 2>&1
rc=$?; echo rc=$rc; tail -15 gpurun_out/r03b/t2.log
[ $rc -eq 0 ] || exit $rc
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03b/$name.json 2> gpurun_out/r03b/$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03b/$name.json | head -2; grep "run again" gpurun_out/r03b/$name.err | tail -2; }
b C2 --workload C2 --steps 1000 --warmup 1000
b C2np --workload C2 --steps 1000 --warmup 1000 --no-profile
AZTOT_DEBUG=4 b C2np_cleanup --workload C2 --steps 1000 --warmup 1000 --no-profile
b C2T --workload C2T --steps 1000 --warmup 1000 --no-profile
b CS1 --case-study 1 --steps 1000 --warmup 1000 --no-profile
b CS2 --case-study 2 --steps 1000 --warmup 1000 --no-profile
AZTOT_DEBUG=4 b CS2_cleanup --case-study 2 --steps 1000 --warmup 1000 --no-profile
b S4 --workload S4 --steps 500 --warmup 500 --no-profile
