mkdir -p gpurun_out/r04
bash tools/r04_numbers.sh
e() { n=$1; shift; timeout -k 10 300 python bench.py --no-cpu-baseline --no-steady "$@" > gpurun_out/r04/emu_final_$n.json 2> gpurun_out/r04/emu_final_$n.err; echo "== emu $n rc=$?"; python tools/r04_summary.py gpurun_out/r04/emu_final_$n.json | cut -c1-200; }
e C4L_single --workload C4L --cell-size 9.176 --steps 500 --warmup 500
for p in 2 4 8; do e C4L_rank_of_$p --workload C4L --cell-size 9.176 --emulate-ranks $p --steps 500 --warmup 500; done
e C4LT_single --workload C4LT --cell-size 9.176 --steps 300 --warmup 300
e C4LT_rank_of_8 --workload C4LT --cell-size 9.176 --emulate-ranks 8 --steps 300 --warmup 300
timeout -k 10 300 python tools/fuzz_many.py 500 20 > gpurun_out/r04/fuzz_final.txt 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/r04/fuzz_final.txt
