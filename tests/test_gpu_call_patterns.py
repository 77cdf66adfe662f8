"""The engine's host state machine under arbitrary call patterns: the reference's loop is per step with every quantity visible after
every kernel (main.cu:281-410, main.cpp:89-142), so whatever a caller does between two steps - read statistics, read or overwrite the state, recompute
the forces, restart the clock - the trajectory must be the one the every-step schedule produces.  Engines compared here:

  default     adaptive lazy re-sort, pair lists, no clean-up launch where the engine runs optimistically, end of a call deferred (Engine::settle)
  eager       the same with every call settled on the spot (AZTOT_DEBUG bit 268435456) and the clean-up launch always in place (bit 4)
  every-step  cells rebuilt on every step (`sort_every = 1`: the reference's schedule, main.cu:300-326)

plus the windows-run-again stress of tools/stress_repair.py as a test (random systems, sort interval held above what the speeds allow).
"""
import numpy as np
import pytest

from aztotmd_amd import api, inputs
from util import add_random_dynamics, rel_err

pytestmark = pytest.mark.gpu
KEYS = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")
SETTLE_EVERY_CALL = 268435456
ALWAYS_CLEANUP = 4
FIXED_INTERVAL = 8192


def systems(name):
    if name == "liquid":          # C2T-like: a thermal LJ liquid whose interval opens up to a dozen steps
        return inputs.lj_case((10, 10, 10), a=5.4, seed=41, vel_T=120.0)
    if name == "radiative":       # radiative thermostat with an equilibration schedule (case study 1's kind of run, dense)
        return inputs.lj_case((9, 9, 9), a=5.3, seed=42, rc=6.5, cell_list=6.5, T=200.0, tstat="radi", vel_T=150.0, radii=[(2.73, 4.731, 0.2)], nEq=40, freqEq=8)
    if name == "fennell":         # two charged species, Fennell / DSF electrostatics
        return inputs.lj_case((10, 10, 10), a=5.3, seed=43, charges=(0.2, -0.2), elec="fenn", vel_T=200.0)
    if name == "bonded":          # bent triatomics: bonds + angles on top of LJ + Fennell (every call ends synchronously there)
        return inputs.molecular_case((8, 8, 8), seed=44, charges=(-0.2, 0.1), elec="fenn", vel_T=300.0)
    if name == "nose":            # Nose-Hoover with equilibration rescaling
        c = inputs.lj_case((9, 9, 9), a=5.3, seed=45, rc=6.5, cell_list=6.5, T=140.0, vel_T=100.0)
        c.update(tstat_type=1, tau=0.08, nEq=30, freqEq=6)
        return c
    raise KeyError(name)


def compare(engs, tol_state, tol_energy, what):
    states = [e.state() for e in engs]
    stats = [e.stats() for e in engs]
    ref, sref = states[-1], stats[-1]
    for s, st in zip(states[:-1], stats[:-1]):
        assert st["step"] == sref["step"], (what, st["step"], sref["step"])
        for k in KEYS:
            if np.isnan(ref[k]).all() or np.abs(np.nan_to_num(ref[k])).max() == 0:
                continue
            assert rel_err(s[k], ref[k]) < tol_state, (what, k, rel_err(s[k], ref[k]))
        for k in ("engTot", "engKin", "engVdW", "engCoul", "engTemp", "engBond", "engAngle"):
            if abs(sref[k]) > 0:
                assert abs(st[k] - sref[k]) <= tol_energy * abs(sref[k]) + 1e-12, (what, k, st[k], sref[k])
        assert st["posCross"] == sref["posCross"] and st["negCross"] == sref["negCross"], what
    return stats


@pytest.mark.parametrize("name,seed", [("liquid", 1), ("liquid", 2), ("radiative", 3), ("fennell", 4), ("bonded", 5), ("nose", 6)])
def test_random_call_patterns(name, seed):
    """random aztot_step sizes (1 ... 300, mostly small) interleaved with get_stats, md_to_host, aztot_forces, set_state (a heating kick through new
    velocities, a restart through the state read back), get_clock / set_clock - the default engine, the eagerly settled one and the every-step schedule
    stay on one trajectory (summation order aside: the default engines re-sort at other steps)."""
    rng = np.random.default_rng(900 + seed)
    case = systems(name)
    engs = [api.Engine(api.Model.from_case(case)),
            api.Engine(api.Model.from_case(case), debug=SETTLE_EVERY_CALL | ALWAYS_CLEANUP),
            api.Engine(api.Model.from_case(case), sort_every=1)]
    total = 0
    log = []
    while total < 700:
        op = rng.choice(["step", "step", "step", "step1", "step1", "stats", "state", "forces", "heat", "restart", "clock"])
        log.append(op)
        if op == "step":
            n = int(rng.choice([2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 300], p=[0.12, 0.12, 0.12, 0.12, 0.12, 0.1, 0.1, 0.06, 0.05, 0.04, 0.03, 0.02]))
            for e in engs:
                e.step(n)
            total += n
        elif op == "step1":       # a caller that couples something to every step: many calls of one step, nothing read in between
            n = int(rng.integers(1, 25))
            for e in engs:
                for _ in range(n):
                    e.step(1)
            total += n
        elif op == "stats":
            compare(engs, 1e-8, 1e-8, (name, seed, total, tuple(log[-6:])))
        elif op == "state":
            s = [e.state(("x", "vx", "fx")) for e in engs]
            for k in ("x", "vx", "fx"):
                assert rel_err(s[0][k], s[2][k]) < 1e-8 and rel_err(s[1][k], s[2][k]) < 1e-8, (name, seed, total, k)
        elif op == "forces":      # aztot_forces between two calls: forces of the current positions, nothing else moves
            before = [e.state(("x", "vx")) for e in engs]
            for e in engs:
                e.forces()
            after = [e.state(("x", "vx")) for e in engs]
            for b, a in zip(before, after):
                assert np.array_equal(b["x"], a["x"]) and np.array_equal(b["vx"], a["vx"])
        elif op == "heat":        # new velocities through aztot_set_state: the interval measured on the old ones is forgotten
            f = float(rng.uniform(0.9, 1.2))
            for e in engs:
                s = e.state(("vx", "vy", "vz"))
                e.set_state(**{k: s[k] * f for k in ("vx", "vy", "vz")})
        elif op == "restart":     # the whole dynamic state read back and written again: an exact restart in place
            for e in engs:
                s = e.state()
                c = e.clock()
                e.set_state(**{k: s[k] for k in KEYS if not np.isnan(s[k]).any()})
                e.set_clock(**c)
        elif op == "clock":
            for e in engs:
                c = e.clock()
                assert c["step"] == total
    stats = compare(engs, 1e-8, 1e-8, (name, seed, "end", tuple(log[-6:])))
    assert stats[2]["sort_interval"] == 1
    for e in engs:
        e.close()


@pytest.mark.parametrize("name", ["liquid", "fennell", "radiative"])
def test_deferred_end_of_call_is_bit_identical(name):
    """One GPU: a call may return before its end (second half-kick, statistics, look) has happened (Engine::settle_now).  With the sort interval held fixed
    (so that both engines rebuild at the same steps) a loop of single-step calls with nothing read in between must leave, bit for bit, the state of the
    engine that settles every call - and of one long call."""
    case = systems(name)
    kw = dict(sort_every=8, debug=FIXED_INTERVAL | ALWAYS_CLEANUP)
    a = api.Engine(api.Model.from_case(case), **kw)
    b = api.Engine(api.Model.from_case(case), sort_every=8, debug=FIXED_INTERVAL | ALWAYS_CLEANUP | SETTLE_EVERY_CALL)
    c = api.Engine(api.Model.from_case(case), **kw)
    a.step(8); b.step(8); c.step(8)                     # (the first call measures and records the first lists on all three)
    for _ in range(37):
        a.step(1)
        b.step(1)
    c.step(37)
    sa, sb, sc = a.state(), b.state(), c.state()
    sta, stb, stc = a.stats(), b.stats(), c.stats()
    for k in KEYS:
        assert np.array_equal(sa[k], sb[k], equal_nan=True), (name, k, rel_err(sa[k], sb[k]))
        assert np.array_equal(sa[k], sc[k], equal_nan=True), (name, k, rel_err(sa[k], sc[k]))
    for k in ("engTot", "engKin", "engVdW", "engCoul", "engTemp"):
        assert sta[k] == stb[k] == stc[k], (name, k, sta[k], stb[k], stc[k])
    assert sta["step"] == stb["step"] == stc["step"] == 45
    # ... and reads in the middle of such a loop change nothing either
    for i in range(20):
        a.step(1); b.step(1)
        if i % 3 == 0:
            a.stats()
        if i % 7 == 0:
            a.state(("x",))
    for k in KEYS:
        assert np.array_equal(a.state()[k], b.state()[k], equal_nan=True), (name, k)


@pytest.mark.parametrize("name", ["liquid", "fennell"])
def test_statistics_after_a_call_served_entirely_by_replayed_cycles(name):
    """Found by the fuzz above (round 4): aztot_forces launches the staging kernel, which books its energies into more partial-sum slots than the list
    kernel; a following call made of whole replayed cycles (hipGraphs: no host code per step) then summed those stale slots into the statistics. A graph slot
    now carries the host's notes of its last step (Engine::LaunchNotes)."""
    case = systems(name)
    a = api.Engine(api.Model.from_case(case), sort_every=8, debug=FIXED_INTERVAL)
    b = api.Engine(api.Model.from_case(case), sort_every=1)
    for e in (a, b):
        e.step(8)
        e.step(16)              # (the cycles are captured by now)
        e.forces()              # k_pair_tile: several waves per cell, each with a slot of its own
        e.step(32)              # four whole cycles of 8: replays only
    sa, sb = a.stats(), b.stats()
    assert sa["step"] == sb["step"] == 56
    for k in ("engTot", "engKin", "engVdW", "engCoul"):
        if abs(sb[k]) > 0:
            assert abs(sa[k] - sb[k]) <= 1e-9 * abs(sb[k]), (name, k, sa[k], sb[k])
    xa, xb = a.state(("x", "vx", "fx")), b.state(("x", "vx", "fx"))
    for k in ("x", "vx", "fx"):
        assert rel_err(xa[k], xb[k]) < 1e-9, (name, k)


def stress_case(seed):
    """tools/stress_repair.py's systems: a lattice of five to nine cells of rc + skin per axis, hot enough to leave the slack within the held interval;
    every third with Fennell charges, every third dense enough for nearest neighbours to be bonded; random thermostat / equilibration schedule on top"""
    rng = np.random.default_rng(seed)
    kw = dict(a=5.4, seed=100 + seed, rc=6.5, cell_list=6.9, vel_T=float(rng.uniform(4000.0, 12000.0)))
    if seed % 3 == 1:
        kw.update(charges=(0.2, -0.2), elec="fenn", r_real=6.5)
    grid = (7, 7, 7)
    if seed % 3 == 2:
        kw.update(a=4.05, vel_T=float(rng.uniform(1500.0, 4000.0)))
        grid = (9, 9, 9)
    case = add_random_dynamics(inputs.lj_case(grid, **kw), seed)
    case["dt"] = 0.002
    return case, [int(v) for v in rng.integers(1, 40, size=6)]


def run_stress(seed):
    case, calls = stress_case(seed)
    a = api.Engine(api.Model.from_case(case), sort_every=16, debug=FIXED_INTERVAL)                     # no clean-up launch: violations repaired from snapshots
    b = api.Engine(api.Model.from_case(case), sort_every=16, debug=FIXED_INTERVAL | ALWAYS_CLEANUP)    # the launch behind every step
    c = api.Engine(api.Model.from_case(case), sort_every=1)
    for n in calls:
        a.step(n); b.step(n); c.step(n)
    sa, sb, sc, sta, stb, stc = a.state(), b.state(), c.state(), a.stats(), b.stats(), c.stats()
    eab = max([rel_err(sa[k], sb[k]) for k in KEYS if np.abs(np.nan_to_num(sb[k])).max() > 0] or [float("nan")])
    eac = max([rel_err(sa[k], sc[k]) for k in KEYS if np.abs(np.nan_to_num(sc[k])).max() > 0] or [float("nan")])
    een = max(abs(sta[k] - stc[k]) / (abs(stc[k]) + 1e-300) for k in ("engTot", "engKin", "engVdW") if abs(stc[k]) > 0)
    out = {"seed": seed, "tstat": case.get("tstat_type", 0), "nEq": case.get("nEq", 0), "bonds": bool(case.get("bonds") is not None and len(case.get("bonds"))),
           "lists": sta["pair_lists"], "violations": (sta["sort_violations"], stb["sort_violations"]), "a_b": eab, "a_every_step": eac, "energies": een,
           "steps": (sta["step"], stc["step"])}
    for e in (a, b, c):
        e.close()
    return out


@pytest.mark.parametrize("seed", range(8))
def test_windows_run_again_on_random_systems(seed):
    """the snapshot / run-the-window-again repair on random systems (NVE, Nose-Hoover, radiative thermostat, equilibration schedules, Fennell charges,
    bonds + angles) with the interval held at 16 steps and random call sizes: equal to the engine that keeps the clean-up launch and to the every-step schedule"""
    r = run_stress(seed)
    assert r["steps"][0] == r["steps"][1]
    assert r["a_b"] < 1e-9 and r["a_every_step"] < 1e-7 and r["energies"] < 1e-8, r
