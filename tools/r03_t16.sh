mkdir -p gpurun_out/r03
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/k_$name.json 2> gpurun_out/r03/k_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/k_$name.json | head -2; grep "aztot: lists recorded\|pair lists for" gpurun_out/r03/k_$name.err | tail -2; }
b z1 --workload C4 --steps 300 --warmup 300
AZTOT_ZGROUP=2 b z2 --workload C4 --steps 300 --warmup 300
AZTOT_ZGROUP=2 b z2_T --workload C4T --steps 300 --warmup 300
AZTOT_ZGROUP=2 b z2_w1 --workload C4 --steps 300 --warmup 300 --split 1
