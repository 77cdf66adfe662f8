"""The call-pattern fuzz of tests/test_gpu_call_patterns.py at the size the bench runs (C4T / C3T, 1 000 188 atoms): the default engine (adaptive interval,
pair lists, no clean-up launch, 100 MB snapshots, deferred call ends) against the every-step schedule under random calls.    python tools/fuzz_big.py [workload] [seed]"""
import sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from aztotmd_amd import api, inputs
from util import rel_err
w = sys.argv[1] if len(sys.argv) > 1 else "C4T"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
case = inputs.config(w)
a = api.Engine(api.Model.from_case(case))
b = api.Engine(api.Model.from_case(case), sort_every=1)
total, bad = 0, 0
t0 = time.time()
def check(tag):
    global bad
    sa, sb = a.stats(), b.stats()
    xa, xb = a.state(("x", "vx", "fx")), b.state(("x", "vx", "fx"))
    e = max(rel_err(xa[k], xb[k]) for k in ("x", "vx", "fx"))
    de = max(abs(sa[k] - sb[k]) / abs(sb[k]) for k in ("engTot", "engKin", "engVdW") if abs(sb[k]) > 0)
    ok = e < 1e-8 and de < 1e-9 and sa["step"] == sb["step"] and sa["posCross"] == sb["posCross"] and sa["negCross"] == sb["negCross"]
    bad += 0 if ok else 1
    print("%s step %d K %d violations %d rebuilds %d  state %.1e energies %.1e  %s" % (tag, sa["step"], sa["sort_interval"], sa["sort_violations"], sa["rebuilds"], e, de, "ok" if ok else "MISMATCH"), flush=True)
while total < 400:
    op = rng.choice(["step", "step", "step1", "stats", "forces", "heat", "restart"])
    if op == "step":
        n = int(rng.choice([2, 5, 13, 34, 89]))
        a.step(n); b.step(n); total += n
    elif op == "step1":
        n = int(rng.integers(1, 12))
        for _ in range(n):
            a.step(1); b.step(1)
        total += n
    elif op == "stats":
        check("stats  ")
    elif op == "forces":
        a.forces(); b.forces()
    elif op == "heat":
        f = float(rng.uniform(0.95, 1.15))
        for e in (a, b):
            s = e.state(("vx", "vy", "vz")); e.set_state(**{k: s[k] * f for k in ("vx", "vy", "vz")})
    elif op == "restart":
        for e in (a, b):
            s = e.state(("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz")); c = e.clock()
            e.set_state(**s and {k: s[k] for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz")}); e.set_clock(**c)
    print(op, total, flush=True)
check("end    ")
print("wall %.1f s, mismatches %d" % (time.time() - t0, bad))
sys.exit(1 if bad else 0)
