# A/B of library builds on the launch-bound cases (fused next step): bash tools/r04_ab_small.sh name1 name2 ...
mkdir -p gpurun_out/r04
run() { n=$1; tag=$2; shift 2; L=""; [ "$n" != tree ] && L="$PWD/ab/$n.so"
  AZTOT_LIB=$L timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-steady > gpurun_out/r04/abs_${n}_$tag.json 2> gpurun_out/r04/abs_${n}_$tag.err; echo "$n $tag rc=$?"; }
for n in "$@"; do
  run $n C2 --workload C2 --steps 4000 --warmup 2000
  run $n C2T --workload C2T --steps 4000 --warmup 2000
  run $n R8 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 1000 --warmup 500
  run $n R4 --workload C4L --cell-size 9.176 --emulate-ranks 4 --steps 500 --warmup 500
done
for n in "$@"; do for t in C2 C2T R8 R4; do python tools/r04_summary.py gpurun_out/r04/abs_${n}_$t.json; done; done
