# interleaved A/B on one box (C4 steady state, twice each): bash tools/r04_ab2.sh nameA nameB
mkdir -p gpurun_out/r04
for rep in 1 2; do for n in "$@"; do
  L=""; [ "$n" != tree ] && L="$PWD/ab/$n.so"
  AZTOT_LIB=$L timeout -k 10 300 python bench.py --workload C4 --steps 600 --warmup 600 --no-cpu-baseline --no-steady > gpurun_out/r04/ab2_${n}_$rep.json 2> gpurun_out/r04/ab2_${n}_$rep.err; echo "$n $rep rc=$?"
done; done
for rep in 1 2; do for n in "$@"; do python tools/r04_summary.py gpurun_out/r04/ab2_${n}_$rep.json; done; done
