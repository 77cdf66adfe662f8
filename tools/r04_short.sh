# the first steps of C4 (lattice at rest) under library builds: per-kernel times of a short run.   bash tools/r04_short.sh name ...
mkdir -p gpurun_out/r04
for n in "$@"; do
  L=""; [ "$n" != tree ] && L="$PWD/ab/$n.so"
  AZTOT_LIB=$L timeout -k 10 300 python bench.py --workload C4 --steps 60 --warmup 20 --no-cpu-baseline --no-steady > gpurun_out/r04/short_${n}.json 2> gpurun_out/r04/short_${n}.err; echo "$n rc=$?"
done
for n in "$@"; do python tools/r04_summary.py gpurun_out/r04/short_${n}.json; done
