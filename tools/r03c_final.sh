# Round 3, second session: the evidence behind DESIGN.md / README.md on the build in the tree (gpurun_out/r03c/, copied into profiles/ by tools/r03c_collect.sh)
mkdir -p gpurun_out/r03c
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03c/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03c/final_tests.log
for W in C4 C4T C3; do bash tools/profile_run.sh $W > gpurun_out/prof_$W.log 2>&1; echo "profile $W rc=$?"; done
for W in C4T C3T C2T; do timeout -k 10 600 python tools/soak_check.py $W > gpurun_out/r03c/soak_$W.txt 2>&1; echo "soak $W rc=$?"; tail -4 gpurun_out/r03c/soak_$W.txt; done
timeout -k 10 600 python tools/long_run.py C4T 30000 > gpurun_out/r03c/long_run_C4T.txt 2>&1; echo "long run rc=$?"; tail -3 gpurun_out/r03c/long_run_C4T.txt
timeout -k 10 600 python tools/long_run.py C2T 100000 > gpurun_out/r03c/long_run_C2T.txt 2>&1; echo "long run C2T rc=$?"; tail -3 gpurun_out/r03c/long_run_C2T.txt
