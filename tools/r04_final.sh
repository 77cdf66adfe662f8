# Round 4: the evidence behind DESIGN.md / README.md on the build in the tree (gpurun_out/r04/, copied into profiles/ by tools/r04_collect.sh)
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/tests_final.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r04/tests_final.log
for W in C4 C4T C3; do bash tools/profile_run.sh $W > gpurun_out/prof_$W.log 2>&1; echo "profile $W rc=$?"; done
