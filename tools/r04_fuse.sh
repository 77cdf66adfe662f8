mkdir -p gpurun_out/r04
run() { n=$1; shift; timeout -k 10 300 python bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-steady "$@" > gpurun_out/r04/fuse_$n.json 2> gpurun_out/r04/fuse_$n.err; echo "$n rc=$?"; python tools/r04_summary.py gpurun_out/r04/fuse_$n.json | cut -c1-330; }
run C4_plain --workload C4
run C4_fused --workload C4 --debug 262144
run C4T_plain --workload C4T
run C4T_fused --workload C4T --debug 262144
