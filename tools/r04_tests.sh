mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/tests_$1.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/r04/tests_$1.log
