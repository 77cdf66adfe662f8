"""Long run cut into irregular aztot_step calls: the default engine (lazy re-sort, pair lists, fusion where it applies) against the every-step schedule.
    python tools/soak_check.py [workload]"""
import sys, time, numpy as np
sys.path.insert(0, '.')
from aztotmd_amd import api, inputs
case = inputs.config(sys.argv[1] if len(sys.argv) > 1 else "C4")
m = api.Model.from_case(case)
a = api.Engine(m, initial_forces=1)
b = api.Engine(m, initial_forces=1, sort_every=1)
rng = np.random.default_rng(3)
tot = 0
e0 = None
t0 = time.time()
for n in [5, 20, 1, 1, 300, 7, 64, 1000, 3, 16, 600, 33]:
    a.step(int(n)); b.step(int(n)); tot += n
    sa, sb = a.stats(), b.stats()
    if e0 is None: e0 = sb["engTot"]
    print(tot, "K", sa["sort_interval"], "viol", sa["sort_violations"], "lists", sa["pair_lists"], sa["cells_without_list"],
          "engTot a %.10f b %.10f rel %.2e drift %.2e" % (sa["engTot"], sb["engTot"], abs(sa["engTot"]-sb["engTot"])/abs(sb["engTot"]), (sb["engTot"]-e0)/abs(e0)), flush=True)
xa, xb = a.state(("x","vx","fx")), b.state(("x","vx","fx"))
for k in ("x","vx","fx"):
    d = np.abs(xa[k]-xb[k]); print(k, "max abs diff", d.max(), "rel", d.max()/np.abs(xb[k]).max())
print("wall", time.time()-t0)
