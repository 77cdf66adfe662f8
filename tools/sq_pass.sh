#!/bin/bash
# SQ counter pass of bench.py for one build of the library:  bash tools/sq_pass.sh <tag> [path/to/libaztot.so] [workload]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; LIB=${2:-}; W=${3:-C4}
O=$R/gpurun_out/sq_$TAG
rm -rf $O && mkdir -p $O/a $O/b
[ -n "$LIB" ] && export AZTOT_LIB=$LIB
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -- python3 $R/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $O/a/bench_line.json 2> $O/a/err.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d $O/b -- python3 $R/bench.py --workload $W --steps 6 --warmup 2 --no-cpu-baseline --no-profile > $O/b/bench_line.json 2> $O/b/err.log
python3 $R/tools/pmc_summary.py $O k_ > $R/gpurun_out/sq_$TAG.txt
find $O -name "*.csv" -size +4M -delete
cat $R/gpurun_out/sq_$TAG.txt
