mkdir -p gpurun_out/r03
AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --case-study 2 --steps 1000 --warmup 1000 > gpurun_out/r03/final_CS2.json 2> gpurun_out/r03/final_CS2.err; echo "== CS2 rc=$?"; grep "aztot:" gpurun_out/r03/final_CS2.err | grep -v "longest step" | tail -4 | cut -c1-250; python tools/bench_summary.py gpurun_out/r03/final_CS2.json | head -2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "pair_lists or any_cell or lists_grow or case_study or surk or dense" > gpurun_out/r03/t24.log 2>&1
echo rc=$?; tail -4 gpurun_out/r03/t24.log
