"""Stress of the window-run-again machinery on one GPU (tests/util.py's random systems with thermostats, equilibration schedules, bonds and angles): an engine
that runs its plain steps without the clean-up launch and repairs violations from snapshots (debug 8192: interval held at `cap` steps whatever the speeds)
against one with the launch behind every step (debug 8192 | 4) and against the every-step schedule.      python tools/stress_repair.py [seeds]"""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from aztotmd_amd import api
from aztotmd_amd import inputs
from util import add_random_dynamics, rel_err

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bad = 0
for seed in range(n_seeds):
    rng = np.random.default_rng(seed)
    # a 7^3-cell lattice (five cells of rc + skin per axis: the smallest box the lazy schedule runs on), hot enough to leave the slack within the held interval
    kw = dict(a=5.4, seed=100 + seed, rc=6.5, cell_list=6.9, vel_T=float(rng.uniform(4000.0, 12000.0)))
    if seed % 3 == 1:
        kw.update(charges=(0.2, -0.2), elec="fenn", r_real=6.5)
    grid = (7, 7, 7)
    if seed % 3 == 2:                               # dense enough for nearest neighbours to be bonded (bonds + angles straddle cells; the sort remaps their indices)
        kw.update(a=4.05, vel_T=float(rng.uniform(1500.0, 4000.0)))
        grid = (9, 9, 9)
    case = add_random_dynamics(inputs.lj_case(grid, **kw), seed)
    case["dt"] = 0.002
    calls = [int(v) for v in rng.integers(1, 40, size=6)]
    try:
        a = api.Engine(api.Model.from_case(case), sort_every=16, debug=8192)
        b = api.Engine(api.Model.from_case(case), sort_every=16, debug=8192 | 4)
        c = api.Engine(api.Model.from_case(case), sort_every=1)
    except api.AztotError as ex:
        print("seed", seed, "skipped:", str(ex)[:80]); continue
    for n in calls:
        a.step(n); b.step(n); c.step(n)
    sa, sb, sc, sta, stb, stc = a.state(), b.state(), c.state(), a.stats(), b.stats(), c.stats()
    keys = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")
    eab = max([rel_err(sa[k], sb[k]) for k in keys if np.abs(np.nan_to_num(sb[k])).max() > 0] or [float('nan')])
    eac = max([rel_err(sa[k], sc[k]) for k in keys if np.abs(np.nan_to_num(sc[k])).max() > 0] or [float('nan')])
    een = max(abs(sta[k] - stc[k]) / (abs(stc[k]) + 1e-300) for k in ("engTot", "engKin", "engVdW") if abs(stc[k]) > 0)
    ok = eab < 1e-9 and eac < 1e-7 and een < 1e-8 and sta["step"] == stc["step"]
    bad += 0 if ok else 1
    print("seed %2d  tstat %d nEq %d bonds %s  lazy %d lists %d  violations %d/%d  a-b %.1e  a-every-step %.1e  energies %.1e  %s" %
          (seed, case.get("tstat_type", 0), case.get("nEq", 0), "yes" if case.get("bonds") is not None and len(case.get("bonds")) else "no", sta["sort_interval"] > 1, sta["pair_lists"], sta["sort_violations"], stb["sort_violations"], eab, eac, een,
           "ok" if ok else "MISMATCH"), flush=True)
    a.close(); b.close(); c.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
