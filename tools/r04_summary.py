"""one line per bench record: value, ms/step, steady state, call overhead, pair kernel, thermalised"""
import json, sys
for n in sys.argv[1:]:
    try:
        d = json.loads(open(n).read().strip().splitlines()[-1])
        ss, co, th, k = d.get("steady_state") or {}, d.get("call_overhead") or {}, d.get("thermalised") or {}, d.get("kernels") or {}
        print("%s: %.1f ns/day %.4f ms/step | steady %s (K %s) | step(1) %s (x%s) step(stat) %s | %s %.1f us frac %.4f | T: %s K=%s | kernels %s" % (
            n.split("/")[-1], d["value"], d["ms_per_step"], ("%.4f" % ss["ms_per_step"]) if ss else None, ss.get("sort_interval"),
            ("%.4f" % co["step1_ms_per_step"]) if co else None, ("%.2f" % co["step1_over_long_call"]) if co else None, ("%.4f" % co["step_stat_ms_per_step"]) if co else None,
            d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], ("%.4f" % th["ms_per_step"]) if isinstance(th, dict) and th else th, th.get("sort_interval") if isinstance(th, dict) else None,
            {a: round(b["avg_us"], 1) for a, b in k.items()}))
    except Exception as ex:
        print(n, "failed:", repr(ex))
