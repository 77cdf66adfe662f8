mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03/full.log 2>&1
echo rc=$?
tail -30 gpurun_out/r03/full.log
