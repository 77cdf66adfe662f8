# A/B of library builds on the steady state of C4 (and C4T): ab/<name>.so against the tree's library.  usage: bash tools/r04_ab.sh name1 name2 ...
mkdir -p gpurun_out/r04
for n in "$@"; do
  L=""; [ "$n" != tree ] && L="$PWD/ab/$n.so"
  for W in C4 C4T; do
    AZTOT_LIB=$L timeout -k 10 300 python bench.py --workload $W --steps 600 --warmup 600 --no-cpu-baseline --no-steady > gpurun_out/r04/ab_${n}_$W.json 2> gpurun_out/r04/ab_${n}_$W.err; echo "$n $W rc=$?"
  done
done
for n in "$@"; do python tools/r04_summary.py gpurun_out/r04/ab_${n}_C4.json gpurun_out/r04/ab_${n}_C4T.json; done
