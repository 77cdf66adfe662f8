mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03/full2.log 2>&1
echo rc=$?; tail -15 gpurun_out/r03/full2.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/v_$name.json 2> gpurun_out/r03/v_$name.err; echo "== $name"; python tools/bench_summary.py gpurun_out/r03/v_$name.json > gpurun_out/r03/v_$name.txt; head -2 gpurun_out/r03/v_$name.txt; grep "aztot: lists recorded" gpurun_out/r03/v_$name.err | tail -1; }
b C4 --workload C4 --steps 200 --warmup 200
b C4T --workload C4T --steps 200 --warmup 200
b C3T --workload C3T --steps 200 --warmup 200
b C4X --workload C4X --steps 200 --warmup 200
