set -e
mkdir -p gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pair_lists or lazy or next_step or energies_only or sort_interval or thermostat_radii" > gpurun_out/r03/t1.log 2>&1 || { tail -40 gpurun_out/r03/t1.log; exit 1; }
tail -3 gpurun_out/r03/t1.log
for w in C4 C4T; do
  AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --workload $w --steps 200 --warmup 200 --no-cpu-baseline > gpurun_out/r03/n1_$w.json 2> gpurun_out/r03/n1_$w.err
  python tools/bench_summary.py gpurun_out/r03/n1_$w.json; grep "aztot:" gpurun_out/r03/n1_$w.err | tail -4
done
