mkdir -p gpurun_out/r03
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03/t28.log 2>&1
echo rc=$?; tail -5 gpurun_out/r03/t28.log
for i in 1 2 3; do python bench.py --workload C4 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('driver', round(d['ms_per_step'],4), round(d['timed_window']['ms_per_step_as_measured'],4), d['config']['sort_violations'])"; done
for i in 1 2; do python bench.py --workload C4 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('default', round(d['ms_per_step'],4), round(d['timed_window']['ms_per_step_as_measured'],4), d['config']['sort_violations'])"; done
python bench.py --workload B3 --steps 400 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('B3 from start', round(d['ms_per_step'],4), d['config']['sort_interval'], d['config']['sort_violations'])"
