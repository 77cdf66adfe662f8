mkdir -p gpurun_out/r03
b() { name=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/l_$name.json 2> gpurun_out/r03/l_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/l_$name.json | head -2; }
b real --workload C4 --steps 40 --warmup 100
b fake_tile --workload C4 --steps 40 --warmup 100 --debug 4
b fake_tile_noconf --workload C4 --steps 40 --warmup 100 --debug 12
