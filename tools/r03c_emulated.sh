# one rank of P on one GPU (loopback halo), 64^3-cell lattice C4L: gpurun_out/r03c/emu_*.json
mkdir -p gpurun_out/r03c
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03c/emu_$name.json 2> gpurun_out/r03c/emu_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03c/emu_$name.json | head -2; grep "interval up to" gpurun_out/r03c/emu_$name.err | tail -1; }
b C4L_single --workload C4L --cell-size 9.176 --steps 500 --warmup 500
for p in 2 4 8; do b C4L_rank_of_$p --workload C4L --cell-size 9.176 --emulate-ranks $p --steps 500 --warmup 500; done
b C4LT_single --workload C4LT --cell-size 9.176 --steps 300 --warmup 300
b C4LT_rank_of_8 --workload C4LT --cell-size 9.176 --emulate-ranks 8 --steps 300 --warmup 300
