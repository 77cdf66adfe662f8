"""Experiment (not a benchmark mode): the driver's window - 5 warm-up steps, 20 timed steps on C4 in a fresh process - with the GPU kept busy by UNRELATED work
(a 40 000-atom engine stepping) for a while right before it.  If the slow steps 11-20 of profiles/r04_C4_first_600_steps_pair_list_us.txt belong to the GPU's
power management after the onset of load, a GPU that is already under load when the window starts does not show them.
      python tools/r04_preheat.py <milliseconds of unrelated load>"""
import sys, time
sys.path.insert(0, '.')
from aztotmd_amd import api, inputs

ms = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
big = api.Engine(api.Model.from_case(inputs.config("C4")), use_graph=1, profile=0, device=0, initial_forces=1)
small = api.Engine(api.Model.from_case(inputs.config("C2T")), use_graph=0, profile=0, device=0, initial_forces=1)
small.step(50); small.sync(); api.device_synchronize(0)
time.sleep(0.05)                                   # (the idle gap a fresh process has between set-up and its first step)
t0 = time.perf_counter(); n = 0
while (time.perf_counter() - t0) * 1e3 < ms:
    small.step(200); small.sync(); n += 200
big.step(5); big.sync(); api.device_synchronize(0)
t1 = time.perf_counter()
big.step(20); big.sync(); api.device_synchronize(0)
t2 = time.perf_counter()
print("unrelated load %.0f ms (%d steps of C2T): C4 window of 20 steps behind 5 of warm-up: %.4f ms/step" % (ms, n, (t2 - t1) * 1e3 / 20), flush=True)
