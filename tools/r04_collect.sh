# copy the evidence of tools/r04_numbers.sh, tools/profile_run.sh, tools/r04_emu.sh and the soak / long runs from gpurun_out/ (scratch) into profiles/ (tracked); tag r04
set -e
for W in C4 C4T C3; do T=r04; [ $W != C4 ] && T=r04_$W; [ -d gpurun_out/prof_$W/stats ] && python tools/make_profiles.py $T gpurun_out/prof_$W/stats gpurun_out/prof_$W/fetch gpurun_out/prof_$W/write $W 1 > /dev/null; done
for f in gpurun_out/r04/final_*.json; do n=$(basename $f .json); n=${n#final_}; cp $f profiles/r04_${n}_bench_line.json; done
for f in gpurun_out/r04/emu_final_*.json; do [ -f $f ] && cp $f profiles/r04_emulated_$(basename $f .json | sed 's/emu_final_//').json; done
for W in C4T C3T C2T; do [ -f gpurun_out/r04/soak_$W.txt ] && cp gpurun_out/r04/soak_$W.txt profiles/r04_soak_$W.txt; done
[ -f gpurun_out/r04/long_run_C4T.txt ] && cp gpurun_out/r04/long_run_C4T.txt profiles/r04_long_run_C4T.txt
[ -f gpurun_out/r04/long_run_C2T.txt ] && cp gpurun_out/r04/long_run_C2T.txt profiles/r04_long_run_C2T.txt
[ -f gpurun_out/r04/tests_final.log ] && cp gpurun_out/r04/tests_final.log profiles/r04_gpu_tests.log
for p in p1 p2 p3; do [ -f gpurun_out/r04/pmc_a_$p.txt ] && cp gpurun_out/r04/pmc_a_$p.txt profiles/r04_pmc_where_waves_wait_$p.txt; done
ls profiles | grep r04 | wc -l
