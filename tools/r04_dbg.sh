# C4 steady state under AZTOT_DEBUG bits: bash tools/r04_dbg.sh <bits> [<bits> ...]
mkdir -p gpurun_out/r04
for d in "$@"; do
  AZTOT_DEBUG=$d timeout -k 10 300 python bench.py --workload C4 --steps 600 --warmup 600 --no-cpu-baseline --no-steady > gpurun_out/r04/dbg_${d}_C4.json 2> gpurun_out/r04/dbg_${d}_C4.err; echo "$d rc=$?"
done
for d in "$@"; do python tools/r04_summary.py gpurun_out/r04/dbg_${d}_C4.json; done
