# usage: bash tools/r04_run.sh <tag> [pytest -k expression | "all" | "none"]  - GPU tests, then the driver-window / default bench lines of C4 and the call overhead of C2
T=$1; K=${2:-all}
mkdir -p gpurun_out/r04
if [ "$K" = all ]; then timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/tests_$T.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r04/tests_$T.log
elif [ "$K" != none ]; then timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "$K" > gpurun_out/r04/tests_$T.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r04/tests_$T.log; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/${T}_driver.json 2> gpurun_out/r04/${T}_driver.err; echo "driver rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/${T}_default.json 2> gpurun_out/r04/${T}_default.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --workload C2 --steps 1000 --warmup 1000 --no-cpu-baseline > gpurun_out/r04/${T}_C2.json 2> gpurun_out/r04/${T}_C2.err; echo "C2 rc=$?"
python tools/r04_summary.py gpurun_out/r04/${T}_driver.json gpurun_out/r04/${T}_default.json gpurun_out/r04/${T}_C2.json
