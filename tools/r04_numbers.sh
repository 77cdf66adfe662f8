# every workload of README.md on the build in the tree: gpurun_out/r04/final_<name>.json
mkdir -p gpurun_out/r04
b() { name=$1; shift; AZTOT_VERBOSE=1 timeout -k 10 400 python bench.py "$@" > gpurun_out/r04/final_$name.json 2> gpurun_out/r04/final_$name.err; echo "== $name rc=$?"; python tools/r04_summary.py gpurun_out/r04/final_$name.json | cut -c1-300; }
b C4 --workload C4
b C4_driver --workload C4 --steps 20 --warmup 5
b C4_long --workload C4 --steps 1000 --warmup 1000 --no-cpu-baseline
b C4_every --workload C4 --sort-every 1 --no-cpu-baseline --no-steady
b C4T --workload C4T --steps 200 --warmup 200 --no-cpu-baseline
b C4X --workload C4X --steps 200 --warmup 200 --no-cpu-baseline
b C3 --workload C3 --steps 200 --warmup 200 --no-cpu-baseline
b C3T --workload C3T --steps 200 --warmup 200 --no-cpu-baseline
b C2 --workload C2 --steps 1000 --warmup 1000 --no-cpu-baseline
b C2T --workload C2T --steps 1000 --warmup 1000 --no-cpu-baseline
b C1 --workload C1 --steps 1000 --warmup 1000 --no-cpu-baseline
b S4 --workload S4 --steps 500 --warmup 500 --no-cpu-baseline
b S40 --workload S40 --steps 200 --warmup 200 --no-cpu-baseline
b M4 --workload M4 --steps 100 --warmup 100 --no-cpu-baseline
b B3 --workload B3 --steps 200 --warmup 200 --no-cpu-baseline
b E2 --workload E2 --steps 100 --warmup 100 --no-cpu-baseline
b CS1 --case-study 1 --steps 1000 --warmup 1000
b CS2 --case-study 2 --steps 1000 --warmup 1000
