"""More seeds of tests/test_gpu_call_patterns.py::test_random_call_patterns than the suite runs (the suite's six found a real bug in round 4).
    python tools/fuzz_many.py first_seed n_seeds"""
import sys, traceback
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_call_patterns as T
first, n = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, first + n):
    for name in ("liquid", "radiative", "fennell", "bonded", "nose"):
        try:
            T.test_random_call_patterns(name, seed)
            print("ok  ", name, seed, flush=True)
        except Exception as ex:   # noqa: BLE001
            bad += 1
            print("FAIL", name, seed, repr(ex)[:600], flush=True)
            traceback.print_exc(limit=3)
print("failures:", bad)
sys.exit(1 if bad else 0)
