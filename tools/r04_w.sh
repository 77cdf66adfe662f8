# C4 steady state with W waves per cell forced: bash tools/r04_w.sh 1 2
mkdir -p gpurun_out/r04
for w in "$@"; do
  timeout -k 10 300 python bench.py --workload C4 --split $w --steps 600 --warmup 600 --no-cpu-baseline --no-steady > gpurun_out/r04/w_${w}_C4.json 2> gpurun_out/r04/w_${w}_C4.err; echo "$w rc=$?"
done
for w in "$@"; do python tools/r04_summary.py gpurun_out/r04/w_${w}_C4.json; done
