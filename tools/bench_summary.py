import json, sys
for f in sys.argv[1:]:
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "ns/day %.1f  ms/step %.4f  Matom-steps/s %.1f" % (d["value"], d["ms_per_step"], d["matom_steps_per_s"]))
    if "kernels" in d:
        print("   ", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items() if v["calls"]})
    if "roofline" in d:
        r = d["roofline"]; print("    roofline", r["kernel"], "%.1f GB/s frac %.4f" % (r["achieved"], r["frac"]))
    if "cpu_baseline" in d:
        print("    cpu", d["cpu_baseline"])
