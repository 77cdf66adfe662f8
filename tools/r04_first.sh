# round 4, first contact: the GPU suite with the new control plane / launcher, then the baseline numbers (driver window, default window, call overhead)
mkdir -p gpurun_out/r04
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/tests_first.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r04/tests_first.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04/base_driver.json 2> gpurun_out/r04/base_driver.err; echo "driver rc=$?"
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r04/base_default.json 2> gpurun_out/r04/base_default.err; echo "default rc=$?"
timeout -k 10 300 python bench.py --workload C2 --steps 1000 --warmup 1000 --no-cpu-baseline > gpurun_out/r04/base_C2.json 2> gpurun_out/r04/base_C2.err; echo "C2 rc=$?"
python - <<'PY'
import json
for n in ("base_driver", "base_default", "base_C2"):
    try:
        d = json.loads(open("gpurun_out/r04/%s.json" % n).read().strip().splitlines()[-1])
        print(n, d["value"], d["ms_per_step"], d.get("steady_state"), d.get("call_overhead"), d["roofline"]["avg_launch_us"], d["roofline"]["traffic"], d["roofline"].get("counters_stale"))
    except Exception as ex:
        print(n, "failed", ex)
PY
