mkdir -p gpurun_out/r04
timeout -k 10 500 python tools/long_run.py C4T 20000 > gpurun_out/r04/long_run_C4T.txt 2>&1; echo "long C4T rc=$?"; tail -3 gpurun_out/r04/long_run_C4T.txt
timeout -k 10 300 python tools/long_run.py C2T 100000 > gpurun_out/r04/long_run_C2T.txt 2>&1; echo "long C2T rc=$?"; tail -3 gpurun_out/r04/long_run_C2T.txt
for W in C4T C3T C2T; do timeout -k 10 500 python tools/soak_check.py $W > gpurun_out/r04/soak_$W.txt 2>&1; echo "soak $W rc=$?"; tail -4 gpurun_out/r04/soak_$W.txt; done
