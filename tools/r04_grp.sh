mkdir -p gpurun_out/r04
run() { n=$1; g=$2; shift; shift; AZTOT_CELLS_PER_GROUP=$g timeout -k 10 300 python bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-steady "$@" > gpurun_out/r04/grp_$n.json 2> gpurun_out/r04/grp_$n.err; echo "$n rc=$?"; python tools/r04_summary.py gpurun_out/r04/grp_$n.json | cut -c1-260; }
run C4_g1 1 --workload C4
run C4_g2 2 --workload C4
run C4_g4 4 --workload C4
run C4T_g1 1 --workload C4T
run C4T_g2 2 --workload C4T
run C4T_g4 4 --workload C4T
run C3_g1 1 --workload C3
run C3_g2 2 --workload C3
run C3_g4 4 --workload C3
