mkdir -p gpurun_out/r04
for m in 1.3 1.15 1.05; do AZTOT_MARGIN=$m timeout -k 10 500 python tools/long_run.py C4T 20000 > gpurun_out/r04/margin_${m}_C4T.txt 2>&1; echo "margin $m rc=$?"; tail -2 gpurun_out/r04/margin_${m}_C4T.txt; done
for m in 1.15 1.05; do AZTOT_MARGIN=$m timeout -k 10 500 python tools/long_run.py C3T 10000 > gpurun_out/r04/margin_${m}_C3T.txt 2>&1; echo "margin $m C3T rc=$?"; tail -2 gpurun_out/r04/margin_${m}_C3T.txt; done
