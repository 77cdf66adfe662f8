#!/bin/bash
# SQ counters of the pair kernel with a debug switch:  bash tools/sq_dbg.sh <debug-bits> [workload]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$1; W=${2:-C4}
O=$R/gpurun_out/sqd_$D
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O -- python3 $R/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-profile --debug $D > $O/bench_line.json 2> $O/err.log
python3 $R/tools/pmc_summary.py $O k_pair > $R/gpurun_out/sqd_$D.txt
find $O -name "*.csv" -size +4M -delete
cat $R/gpurun_out/sqd_$D.txt
