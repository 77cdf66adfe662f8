mkdir -p gpurun_out/r03
b() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/h_$name.json 2> gpurun_out/r03/h_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/h_$name.json | head -2; }
b C4T --workload C4T --steps 200 --warmup 200
b M4 --workload M4 --steps 60 --warmup 60
