#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's numbers on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_run.sh [workload]
# Four separate runs as MI355X_MICROARCH.md prescribes: kernel trace + stats; FETCH_SIZE; WRITE_SIZE; SQ counters.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
W=${1:-C4}
O=$R/gpurun_out/prof_$W
rm -rf $O && mkdir -p $O/stats $O/fetch $O/write $O/sq $O/fp64
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --workload $W --steps 200 --warmup 100 --no-cpu-baseline --no-steady > $O/stats/bench_line.json 2> $O/stats/err.log
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --workload $W --steps 40 --warmup 40 --no-cpu-baseline --no-profile --no-steady > $O/fetch/bench_line.json 2> $O/fetch/err.log
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --workload $W --steps 40 --warmup 40 --no-cpu-baseline --no-profile --no-steady > $O/write/bench_line.json 2> $O/write/err.log
echo "write done"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq -- python3 $R/bench.py --workload $W --steps 40 --warmup 40 --no-cpu-baseline --no-profile --no-steady > $O/sq/bench_line.json 2> $O/sq/err.log
echo "sq done"
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $O/fp64 -- python3 $R/bench.py --workload $W --steps 40 --warmup 40 --no-cpu-baseline --no-profile --no-steady > $O/fp64/bench_line.json 2> $O/fp64/err.log
echo "fp64 done"
python3 $R/tools/pmc_summary.py $O/sq > $O/sq_summary.txt
python3 $R/tools/pmc_summary.py $O/fp64 > $O/fp64_summary.txt
find $O -name "*.csv" -size +8M -delete      # the merge-back limit is 64 MiB: keep the summaries, drop bulky traces
ls -la $O/stats | head
