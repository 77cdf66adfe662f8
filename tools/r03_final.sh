# the round's final evidence on the build in the tree: GPU tests, every workload, rocprofv3 summaries, soak runs, rank emulation
mkdir -p gpurun_out/r03
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03/final_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r03/final_tests.log
bash tools/r03_numbers.sh > gpurun_out/r03/numbers.log 2>&1; grep -A1 "^==" gpurun_out/r03/numbers.log | grep -v "^--" | paste - - | cut -c1-130
bash tools/r03_prof.sh > gpurun_out/r03/prof.log 2>&1; grep "rc=\|^== " gpurun_out/r03/prof.log | cut -c1-100
