mkdir -p gpurun_out/r04
run() { n=$1; lib=$2; cpw=$3; L=""; [ "$lib" != tree ] && L="$PWD/ab/$lib.so"; AZTOT_CELLS_PER_WAVE=$cpw AZTOT_LIB=$L timeout -k 10 300 python bench.py --workload C4 --steps 400 --warmup 400 --no-cpu-baseline --no-steady > gpurun_out/r04/cpw_$n.json 2> gpurun_out/r04/cpw_$n.err; echo "$n rc=$?"; python tools/r04_summary.py gpurun_out/r04/cpw_$n.json | cut -c1-230; }
run base1 base 1
run tree1 tree 1
run tree3 tree 3
run tree6 tree 6
run tree14 tree 14
run cap1 cpw_cap 1
run cap3 cpw_cap 3
run cap6 cpw_cap 6
run cap14 cpw_cap 14
