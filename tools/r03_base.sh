set -e
mkdir -p gpurun_out/r03
for w in C4 C4T C4X C3T; do
  python bench.py --workload $w --steps 200 --warmup 200 --no-cpu-baseline > gpurun_out/r03/base_$w.json 2> gpurun_out/r03/base_$w.err
  echo "$w done"; python tools/bench_summary.py gpurun_out/r03/base_$w.json || true
done
