set -e
mkdir -p gpurun_out/r03
run() { # name, env..., args
  name=$1; shift
  env "$@" > /dev/null 2>&1 || true
}
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --steps 200 --warmup 200 --no-cpu-baseline "$@" > gpurun_out/r03/x_$name.json 2> gpurun_out/r03/x_$name.err; echo "== $name"; python tools/bench_summary.py gpurun_out/r03/x_$name.json | head -2; grep "aztot: lists recorded" gpurun_out/r03/x_$name.err | tail -1; }
b C4_noskin --workload C4 --skin -1
AZTOT_CAND_CAP=320 b C4_noskin_c320 --workload C4 --skin -1
b C4T_ph1 --workload C4T --debug 1
b C4T_ph2 --workload C4T --debug 2
AZTOT_CAND_CAP=512 AZTOT_ITER_CAP=48 b C4T_big --workload C4T
AZTOT_CAND_CAP=512 AZTOT_ITER_CAP=48 b C4T_big_s30 --workload C4T --skin 0.30
AZTOT_CAND_CAP=512 AZTOT_ITER_CAP=48 b C4T_big_s40 --workload C4T --skin 0.40
AZTOT_CAND_CAP=576 AZTOT_ITER_CAP=56 b C4T_big_s55 --workload C4T --skin 0.55
