"""Stress of the window-run-again machinery on one GPU (more seeds than tests/test_gpu_call_patterns.py::test_windows_run_again_on_random_systems runs):
an engine that runs its plain steps without the clean-up launch and repairs violations from snapshots (debug 8192: interval held at 16 steps whatever the
speeds) against one with the launch behind every step (debug 8192 | 4) and against the every-step schedule.      python tools/stress_repair.py [seeds]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_gpu_call_patterns import run_stress

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bad = 0
for seed in range(n_seeds):
    r = run_stress(seed)
    ok = r["a_b"] < 1e-9 and r["a_every_step"] < 1e-7 and r["energies"] < 1e-8 and r["steps"][0] == r["steps"][1]
    bad += 0 if ok else 1
    print(r, "ok" if ok else "MISMATCH", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
