mkdir -p gpurun_out/r03
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/u_$name.json 2> gpurun_out/r03/u_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/u_$name.json > gpurun_out/r03/u_$name.txt; head -2 gpurun_out/r03/u_$name.txt; grep "aztot: lists recorded\|interval up to" gpurun_out/r03/u_$name.err | tail -2; }
b C4L --workload C4L --cell-size 9.176 --steps 200 --warmup 300
b C4L_r8 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 200 --warmup 300
b C4L_r4 --workload C4L --cell-size 9.176 --emulate-ranks 4 --steps 200 --warmup 300
b C4L_r2 --workload C4L --cell-size 9.176 --emulate-ranks 2 --steps 200 --warmup 300
b C4_r7_old --workload C4 --skin -1 --emulate-ranks 7 --steps 200 --warmup 300
python -m pytest tests/test_gpu_slab.py -q -m gpu -k torch_in_the -s 2>&1 | grep "RCCL self-test"
