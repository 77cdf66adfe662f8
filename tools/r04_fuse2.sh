mkdir -p gpurun_out/r04
run() { n=$1; lib=$2; shift; shift; L=""; [ "$lib" != tree ] && L="$PWD/ab/$lib.so"; AZTOT_LIB=$L timeout -k 10 300 python bench.py --workload C4 --steps 400 --warmup 400 --no-cpu-baseline --no-steady "$@" > gpurun_out/r04/fuse2_$n.json 2> gpurun_out/r04/fuse2_$n.err; echo "$n rc=$?"; python tools/r04_summary.py gpurun_out/r04/fuse2_$n.json | cut -c1-200; }
run plain tree
run fused tree --debug 262144
run fused_nof nof --debug 262144
run fused_nof_nor0 nof_nor0 --debug 262144
