mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_slab.py -x -q -m gpu -k "heating or disagree or slabs_match" > gpurun_out/r03/t17.log 2>&1
echo rc=$?; tail -15 gpurun_out/r03/t17.log
