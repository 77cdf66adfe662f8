# emulated slab ranks on the 64^3-cell lattice (C4L / C4LT, cell 9.176 A: 40 cell layers = 8 ranks x 5): gpurun_out/r04/emu_<tag>_*.json
T=$1
mkdir -p gpurun_out/r04
for n in 8; do timeout -k 10 300 python bench.py --workload C4L --cell-size 9.176 --emulate-ranks $n --steps 500 --warmup 500 --no-cpu-baseline > gpurun_out/r04/emu_${T}_C4L_rank_of_$n.json 2> gpurun_out/r04/emu_${T}_C4L_rank_of_$n.err; echo "rank of $n rc=$?"; done
timeout -k 10 300 python bench.py --workload C4LT --cell-size 9.176 --emulate-ranks 8 --steps 500 --warmup 500 --no-cpu-baseline > gpurun_out/r04/emu_${T}_C4LT_rank_of_8.json 2> gpurun_out/r04/emu_${T}_C4LT_rank_of_8.err; echo "C4LT rank of 8 rc=$?"
python tools/r04_summary.py gpurun_out/r04/emu_${T}_C4L_rank_of_8.json gpurun_out/r04/emu_${T}_C4LT_rank_of_8.json
