"""Long single-engine run: energy conservation, sort interval, violations and unlisted cells of the default engine over tens of thousands of steps.
    python tools/long_run.py [workload] [steps]"""
import sys, time
sys.path.insert(0, '.')
from aztotmd_amd import api, inputs
w = sys.argv[1] if len(sys.argv) > 1 else "C4T"
total = int(sys.argv[2]) if len(sys.argv) > 2 else 30000
e = api.Engine(api.Model.from_case(inputs.config(w)))
e.step(200)
s0 = e.stats()
t0 = time.time()
done = 0
print("workload", w, "engTot after 200 steps %.6f" % s0["engTot"])
while done < total:
    n = min(5000, total - done)
    e.step(n); done += n
    s = e.stats()
    print("step %6d  K %2d  violations %d  cells without list %d  rebuilds %d  T %.2f K  engTot %.6f  drift %.2e" %
          (s["step"], s["sort_interval"], s["sort_violations"], s["cells_without_list"], s["rebuilds"], s["temperature"], s["engTot"], (s["engTot"] - s0["engTot"]) / abs(s0["engTot"])), flush=True)
print("wall %.1f s for %d steps: %.4f ms/step" % (time.time() - t0, total, (time.time() - t0) / total * 1e3))
