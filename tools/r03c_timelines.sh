# kernel timelines (rocprofv3 --kernel-trace) of the launch-bound cases: an emulated rank of 8, the 40 000-atom box C2 and case study 1 -> gpurun_out/r03c/timeline_*.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r03c
cd /tmp && export TMPDIR=/tmp
t() { name=$1; shift; rm -rf $R/gpurun_out/tl_$name; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$name -- python3 $R/bench.py --no-cpu-baseline --no-profile --no-graph "$@" > /dev/null 2> $R/gpurun_out/tl_$name.err; python3 $R/tools/timeline.py $R/gpurun_out/tl_$name 600 > $R/gpurun_out/r03c/timeline_$name.txt; echo "== $name"; head -8 $R/gpurun_out/r03c/timeline_$name.txt; find $R/gpurun_out/tl_$name -name "*.csv" -size +4M -delete; }
t C4L_rank_of_8 --workload C4L --cell-size 9.176 --emulate-ranks 8 --steps 300 --warmup 600
t C2 --workload C2 --steps 300 --warmup 600
t case_study_1 --case-study 1 --steps 300 --warmup 600
