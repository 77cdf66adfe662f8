"""How the kernels' durations develop over an engine's first steps (HIP events on the engine's stream): is the gap between the driver's 20-step window and the
steady state the hardware warming up or something the engine does?      python tools/warmup_curve.py [workload] [calls] [steps per call]"""
import sys, time
sys.path.insert(0, '.')
from aztotmd_amd import api, inputs
w = sys.argv[1] if len(sys.argv) > 1 else "C4"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n = int(sys.argv[3]) if len(sys.argv) > 3 else 25
e = api.Engine(api.Model.from_case(inputs.config(w)), initial_forces=1)
e.step(5)
e.set_profile(1)
t0 = time.time()
for c in range(calls):
    e.reset_kernel_times()
    e.step(n)
    k = e.kernel_times()
    us = lambda name: 1e3 * k[name]["ms"] / max(k[name]["calls"], 1) if name in k else 0.0
    print("steps %5d  t %.3f s  pair_list %.1f us  integrate1 %.1f  cleanup %.1f  K %d" % (5 + (c + 1) * n, time.time() - t0, us("pair_list"), us("integrate1"), us("pair_cleanup"), e.stats()["sort_interval"]), flush=True)
