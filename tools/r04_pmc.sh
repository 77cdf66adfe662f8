# where do the waves of k_pair_list wait?  Two --pmc passes (no tracing beside them), summaries into gpurun_out/r04/pmc_<tag>_*.txt
T=${1:-a}; W=${2:-C4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04/pmc_$T
rm -rf $O && mkdir -p $O/p1 $O/p2 $O/p3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 $R/bench.py --workload $W --steps 30 --warmup 30 --no-cpu-baseline --no-profile --no-steady > $O/p1/line.json 2> $O/p1/err.log; echo "p1 rc=$?"
rocprofv3 --pmc TA_BUSY_avr TD_TD_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE --output-format csv -d $O/p2 -- python3 $R/bench.py --workload $W --steps 30 --warmup 30 --no-cpu-baseline --no-profile --no-steady > $O/p2/line.json 2> $O/p2/err.log; echo "p2 rc=$?"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --output-format csv -d $O/p3 -- python3 $R/bench.py --workload $W --steps 30 --warmup 30 --no-cpu-baseline --no-profile --no-steady > $O/p3/line.json 2> $O/p3/err.log; echo "p3 rc=$?"
cd $R
for p in p1 p2 p3; do python3 tools/pmc_summary.py $O/$p k_pair_list > gpurun_out/r04/pmc_${T}_$p.txt; python3 tools/pmc_summary.py $O/$p k_build >> gpurun_out/r04/pmc_${T}_$p.txt; python3 tools/pmc_summary.py $O/$p k_integrate_plain >> gpurun_out/r04/pmc_${T}_$p.txt; cat gpurun_out/r04/pmc_${T}_$p.txt; done
find $O -name "*.csv" -size +4M -delete
