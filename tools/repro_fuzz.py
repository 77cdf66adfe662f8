import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import test_gpu_call_patterns as T
from aztotmd_amd import api
name, seed = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(900 + seed)
case = T.systems(name)
engs = [api.Engine(api.Model.from_case(case)), api.Engine(api.Model.from_case(case), debug=T.SETTLE_EVERY_CALL | T.ALWAYS_CLEANUP), api.Engine(api.Model.from_case(case), sort_every=1)]
total = 0
def show(tag):
    for i, e in enumerate(engs):
        st = e.stats()
        print(tag, i, "step", st["step"], "K", st["sort_interval"], "viol", st["sort_violations"], "rebuilds", st["rebuilds"], " ".join("%s=%.6f" % (k, st[k]) for k in ("engTot", "engKin", "engVdW", "engCoul")), flush=True)
while total < 700:
    op = rng.choice(["step", "step", "step", "step1", "step1", "stats", "state", "forces", "heat", "restart", "clock"])
    if op == "step":
        n = int(rng.choice([2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 300], p=[0.12, 0.12, 0.12, 0.12, 0.12, 0.1, 0.1, 0.06, 0.05, 0.04, 0.03, 0.02]))
        for e in engs: e.step(n)
        total += n
        print("step", n, total)
    elif op == "step1":
        n = int(rng.integers(1, 25))
        for e in engs:
            for _ in range(n): e.step(1)
        total += n
        print("step1 x", n, total)
    elif op == "stats":
        show("stats@%d" % total)
    elif op == "state":
        [e.state(("x", "vx", "fx")) for e in engs]; print("state")
    elif op == "forces":
        for e in engs: e.forces()
        print("forces")
    elif op == "heat":
        f = float(rng.uniform(0.9, 1.2))
        for e in engs:
            s = e.state(("vx", "vy", "vz")); e.set_state(**{k: s[k] * f for k in ("vx", "vy", "vz")})
        print("heat", f)
    elif op == "restart":
        for e in engs:
            s = e.state(); c = e.clock()
            e.set_state(**{k: s[k] for k in T.KEYS if not np.isnan(s[k]).any()}); e.set_clock(**c)
        print("restart")
    elif op == "clock":
        for e in engs: c = e.clock()
        print("clock", c["step"])
show("end")
