mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_slab.py -x -q -m gpu -k "whole_number or any_cell_population or lists_grow or restart or pressure or cli_reproduces or disagree or torch_in_the_process or slabs_match or overlapped" > gpurun_out/r03/t5.log 2>&1
echo rc=$?; tail -40 gpurun_out/r03/t5.log
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "thermalised" > gpurun_out/r03/t5b.log 2>&1
echo rc=$?; tail -30 gpurun_out/r03/t5b.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03/b5.json 2> gpurun_out/r03/b5.err; echo rc=$?; tail -5 gpurun_out/r03/b5.err; python tools/bench_summary.py gpurun_out/r03/b5.json
