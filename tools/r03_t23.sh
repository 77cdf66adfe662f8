mkdir -p gpurun_out/r03
for k in 1 2; do
AZTOT_VERBOSE=1 timeout -k 10 300 python bench.py --case-study $k --steps 1000 --warmup 1000 > gpurun_out/r03/final_CS$k.json 2> gpurun_out/r03/final_CS$k.err; echo "== CS$k rc=$?"; tail -2 gpurun_out/r03/final_CS$k.err | cut -c1-200; python tools/bench_summary.py gpurun_out/r03/final_CS$k.json | head -2
done
