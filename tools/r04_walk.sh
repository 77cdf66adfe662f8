mkdir -p gpurun_out/r04
run() { n=$1; shift; timeout -k 10 300 python bench.py --steps 400 --warmup 400 --no-cpu-baseline --no-steady "$@" > gpurun_out/r04/walk_$n.json 2> gpurun_out/r04/walk_$n.err; echo "$n rc=$?"; python tools/r04_summary.py gpurun_out/r04/walk_$n.json | cut -c1-250; }
run C4_x --workload C4 --debug 536870912
run C4_y --workload C4
run C4T_x --workload C4T --debug 536870912
run C4T_y --workload C4T
run C3_x --workload C3 --debug 536870912
run C3_y --workload C3
