mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or sort_interval or thermostat_radii or any_cell or lists_grow or restart or whole_number or family_kernels or surk" > gpurun_out/r03/t10.log 2>&1
echo rc=$?; tail -12 gpurun_out/r03/t10.log
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/s_$name.json 2> gpurun_out/r03/s_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/s_$name.json > gpurun_out/r03/s_$name.txt; head -2 gpurun_out/r03/s_$name.txt; grep "aztot: lists recorded" gpurun_out/r03/s_$name.err | tail -1; }
b C4T_w1 --workload C4T --steps 200 --warmup 200 --split 1
b C4T_w2 --workload C4T --steps 200 --warmup 200 --split 2
b S40_w1 --workload S40 --steps 200 --warmup 200 --split 1
b S40_w2 --workload S40 --steps 200 --warmup 200 --split 2
b S40_w4 --workload S40 --steps 200 --warmup 200 --split 4
b M4_w1 --workload M4 --steps 50 --warmup 50 --split 1
b M4_w2 --workload M4 --steps 50 --warmup 50 --split 2
b M4_w4 --workload M4 --steps 50 --warmup 50 --split 4
b C4L_r8_w1 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 200 --warmup 300 --split 1
b C4L_r8_w2 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 200 --warmup 300 --split 2
b C4L_r8_w4 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 200 --warmup 300 --split 4
b C2_w1 --workload C2 --steps 500 --warmup 500 --split 1
b C2_w2 --workload C2 --steps 500 --warmup 500 --split 2
b C2_w4 --workload C2 --steps 500 --warmup 500 --split 4
