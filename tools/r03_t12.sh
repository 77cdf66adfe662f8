mkdir -p gpurun_out/r03
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for w in 1 4; do
O=$R/gpurun_out/r03/sqS40_w$w; rm -rf $O; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O -- python3 $R/bench.py --workload S40 --steps 20 --warmup 40 --no-cpu-baseline --no-profile --split $w > $O/bench_line.json 2> $O/err.log
python3 $R/tools/pmc_summary.py $O k_pair_list > $R/gpurun_out/r03/sqS40_w$w.txt
find $O -name "*.csv" -size +4M -delete
echo "== W=$w"; cat $R/gpurun_out/r03/sqS40_w$w.txt
done
