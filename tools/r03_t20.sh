mkdir -p gpurun_out/r03
timeout -k 10 900 python -m pytest tests/test_gpu_slab.py -x -q -m gpu -k "several_waves or grow_their_lists" > gpurun_out/r03/t20.log 2>&1
echo rc=$?; tail -25 gpurun_out/r03/t20.log
