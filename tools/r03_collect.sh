# copy the evidence of tools/r03_final.sh from gpurun_out/ (scratch) into profiles/ (tracked)
set -e
for W in C4 C4T C3; do T=r03; [ $W != C4 ] && T=r03_$W; python tools/make_profiles.py $T gpurun_out/prof_$W/stats gpurun_out/prof_$W/fetch gpurun_out/prof_$W/write $W 1 > /dev/null; done
for n in single rank_of_2 rank_of_4 rank_of_8; do cp gpurun_out/r03/emu_C4L_$n.json profiles/r03_emulated_C4L_$n.json; done
for W in C4T C3T C2T; do cp gpurun_out/r03/soak_$W.txt profiles/r03_soak_$W.txt; done
cp gpurun_out/r03/final_C4.json profiles/r03_bench_line.json
cp gpurun_out/r03/final_C4_driver.json profiles/r03_bench_line_steps20_warmup5.json
cp gpurun_out/r03/final_C4_long.json profiles/r03_bench_line_steps1000_warmup1000.json
for W in C4T C3 C3T C4X S40 M4; do cp gpurun_out/r03/final_$W.json profiles/r03_${W}_bench_line.json; done
cp gpurun_out/r03/final_CS1.json profiles/r03_case_study_1_bench_line.json
cp gpurun_out/r03/final_CS2.json profiles/r03_case_study_2_bench_line.json
cp gpurun_out/r03/final_tests.log profiles/r03_gpu_tests.log
