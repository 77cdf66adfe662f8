# copy the evidence of tools/r03c_final.sh, tools/r03_numbers.sh and tools/r03c_emulated.sh from gpurun_out/ (scratch) into profiles/ (tracked); tag r03c
set -e
for W in C4 C4T C3; do T=r03c; [ $W != C4 ] && T=r03c_$W; python tools/make_profiles.py $T gpurun_out/prof_$W/stats gpurun_out/prof_$W/fetch gpurun_out/prof_$W/write $W 1 > /dev/null; done
for n in C4L_single C4L_rank_of_2 C4L_rank_of_4 C4L_rank_of_8 C4LT_single C4LT_rank_of_8; do cp gpurun_out/r03c/emu_$n.json profiles/r03c_emulated_$n.json; done
for W in C4T C3T C2T; do cp gpurun_out/r03c/soak_$W.txt profiles/r03c_soak_$W.txt; done
cp gpurun_out/r03c/long_run_C4T.txt profiles/r03c_long_run_C4T_30000_steps.txt
cp gpurun_out/r03c/long_run_C2T.txt profiles/r03c_long_run_C2T_100000_steps.txt
cp gpurun_out/r03c/final_C4.json profiles/r03c_bench_line.json
cp gpurun_out/r03c/final_C4_driver.json profiles/r03c_bench_line_steps20_warmup5.json
cp gpurun_out/r03c/final_C4_long.json profiles/r03c_bench_line_steps1000_warmup1000.json
for W in C4T C3 C3T C4X C2 C2T S40 M4 B3; do cp gpurun_out/r03c/final_$W.json profiles/r03c_${W}_bench_line.json; done
cp gpurun_out/r03c/final_CS1.json profiles/r03c_case_study_1_bench_line.json
cp gpurun_out/r03c/final_CS2.json profiles/r03c_case_study_2_bench_line.json
cp gpurun_out/r03c/final_tests.log profiles/r03c_gpu_tests.log
