# round-3 evidence: rocprofv3 summaries for C4 / C4T / C3, soak runs, emulated ranks -> gpurun_out/ (copied into profiles/ by tools/make_profiles.py afterwards)
mkdir -p gpurun_out/r03
for W in C4 C4T C3; do bash tools/profile_run.sh $W > gpurun_out/prof_$W.log 2>&1; echo "profile $W rc=$?"; done
for W in C4T C3T C2T; do timeout -k 10 600 python tools/soak_check.py $W > gpurun_out/r03/soak_$W.txt 2>&1; echo "soak $W rc=$?"; tail -4 gpurun_out/r03/soak_$W.txt; done
b() { name=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > gpurun_out/r03/emu_$name.json 2> gpurun_out/r03/emu_$name.err; echo "== $name rc=$?"; python tools/bench_summary.py gpurun_out/r03/emu_$name.json | head -2; }
b C4L_single --workload C4L --cell-size 9.176 --steps 500 --warmup 500
b C4L_rank_of_2 --workload C4L --cell-size 9.176 --emulate-ranks 2 --steps 500 --warmup 500
b C4L_rank_of_4 --workload C4L --cell-size 9.176 --emulate-ranks 4 --steps 500 --warmup 500
b C4L_rank_of_8 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 500 --warmup 500
