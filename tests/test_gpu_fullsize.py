"""GPU parity at the sizes bench.py runs (BASELINE configs 3 and 4: 1 000 188 atoms, 74 088 cells) and on the reference's two
shipped example inputs (configs 1 and 5), through the C ABI.

At 1 M atoms the engine takes code paths no small test reaches: the multi-workgroup prefix sum (k_scan_totals + k_scan_apply, above
16 384 cells; reference: calc_firstAtomInCell cuSort.cu:130-143), the deferred second half-kick (more than 262 144 atoms; reference:
verlet_2stage cuMDfunc.cu:521-600 / integrate2 integrators.cpp:486-531) and the 8-XCD workgroup -> cell mapping on 74 088 workgroups.
The checker is the reference's own serial code (oracle/_ref/ref_driver, main.cpp:89-142) when the binary travelled to this machine,
else the oracle (bit-identical to it on every golden fixture).
"""
import os

import numpy as np
import pytest

from aztotmd_amd import api, inputs
from oracle import oracle, parse
from util import FRC, VEL, case_from_parsed, materialise_case_study, per_atom_err, rel_err

pytestmark = pytest.mark.gpu
XVF = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz")


def cpu_steps(case, nsteps):
    """(per-atom state, energies, who computed it) after `nsteps` from F = 0 on the serial CPU path."""
    if oracle.ref_available():
        c = dict(case)
        c.update(nsteps=nsteps, dump=[nsteps], init_forces=0, use_clist=1, center_box=0)
        d = oracle.run_ref(c, timeout=900)["dumps"][nsteps]
        return d, {"engVdW": d["engVdW"], "engCoul": d["engElec3"], "engKin": d["engKin"], "engTot": d["engTot"]}, "reference binary"
    o = oracle.Oracle(case)
    o.step(nsteps)                       # forces start at 0, as on the reference's GPU path (sys_init.cpp:551-553)
    st = o.stats()
    return o.state(), {"engVdW": st["engVdW"], "engCoul": st["engElec3"], "engKin": st["engKin"], "engTot": st["engTot"]}, "oracle"


def host_cell_table(s, box, dims):
    """count_cell (cuSort.cu:114-128): cell = floor(x * cRevSize) per axis, in double; then the exclusive prefix sum."""
    idx = []
    for k, key in enumerate(("x", "y", "z")):
        c = np.floor(s[key] * (dims[k] / box[k])).astype(np.int64) % dims[k]
        idx.append(c)
    cell = (idx[0] * dims[1] + idx[1]) * dims[2] + idx[2]
    hist = np.bincount(cell, minlength=dims[0] * dims[1] * dims[2])
    start = np.concatenate([[0], np.cumsum(hist)])
    return cell, start


def check_cell_table(e, s, box):
    dims, start, ids = e.cell_table()
    cell, ref_start = host_cell_table(s, box, dims)
    assert np.array_equal(start, ref_start), "cell offsets differ from the host prefix sum"
    assert np.array_equal(np.sort(ids), np.arange(len(ids))), "every atom sits in exactly one slot"
    slot_cell = np.repeat(np.arange(len(start) - 1), np.diff(start))
    assert np.array_equal(cell[ids], slot_cell), "an atom sits in a slot of the wrong cell"
    same = slot_cell[1:] == slot_cell[:-1]
    assert np.all(ids[1:][same] > ids[:-1][same]), "atoms of one cell are not in id order"
    return dims


@pytest.mark.parametrize("name", ["C4", "C3"])
def test_full_size_steps_match_the_serial_cpu_path(name):
    """BASELINE configs 4 and 3 at their full 1 000 188 atoms: 3 steps from F = 0, per-atom x / v / f to 1e-9 (north star), energies
    to 1e-11, plus the device's cell table against a host prefix sum."""
    case = inputs.config(name)
    n = len(case["types"])
    assert n == 1000188
    ref, eref, who = cpu_steps(case, 3)
    # the reference's schedule: cells rebuilt every step; debug bit 8388608: hipGraph replay although the engine would launch a system of this size eagerly
    e = api.Engine(api.Model.from_case(case), initial_forces=0, sort_every=1, debug=8388608)
    e.step(3)
    s, st = e.state(), e.stats()
    assert st["n_cells"] == 42 ** 3 and st["n_cells"] > 16384           # the multi-workgroup scan is the one that ran
    for k in XVF:
        assert rel_err(s[k], ref[k]) < 1e-9, (name, who, k, rel_err(s[k], ref[k]))
    for k in ("fx", "fy", "fz"):
        assert rel_err(s[k], ref[k]) < 1e-11, (name, who, k, rel_err(s[k], ref[k]))
    # ... and atom by atom: every one of the 1 000 188 forces (velocities) within 1e-9 of ITS OWN magnitude (floor: a thousandth of the rms)
    assert per_atom_err(s, ref, FRC) < 1e-9, (name, who, per_atom_err(s, ref, FRC))
    assert per_atom_err(s, ref, VEL) < 1e-9, (name, who, per_atom_err(s, ref, VEL))
    for k, v in eref.items():
        assert abs(st[k] - v) <= 1e-11 * abs(v) + 1e-12, (name, who, k, st[k], v)
    assert st["pairs_dropped"] == 0
    for k in ("fx", "fy", "fz"):
        assert abs(s[k].sum()) < 1e-8                                    # Newton 3 over 1 M atoms
    check_cell_table(e, s, case["box"])
    # the deferred half-kick path must give the same trajectory as k_integrate2 every step (debug bit 128), bit for bit
    e2 = api.Engine(api.Model.from_case(case), initial_forces=0, debug=128, use_graph=0, sort_every=1)
    e2.step(3)
    s2 = e2.state()
    for k in XVF:
        assert np.array_equal(s[k], s2[k]), k
    e2.close()
    # and more steps, graph-replayed in pairs, still agree with eager launches
    e.step(5)
    e3 = api.Engine(api.Model.from_case(case), initial_forces=0, use_graph=0, sort_every=1)
    e3.step(8)
    s, s3 = e.state(), e3.state()
    for k in ("x", "vx", "fx"):
        assert np.array_equal(s[k], s3[k]), k


@pytest.mark.parametrize("name", ["C4", "C3"])
def test_lazy_resort_at_full_size(name):
    """the default schedule (cells rebuilt only when an atom could have left the slack between the stencil's reach and the cut-off) against the
    every-step schedule on the 1 M-atom boxes: 40 steps in four calls, per-atom x / v / f to 1e-10, energies to 1e-11, equal wall counters;
    the interval actually opens up (> 1) and no violation occurs."""
    case = inputs.config(name)
    a = api.Engine(api.Model.from_case(case))
    b = api.Engine(api.Model.from_case(case), sort_every=1)
    for n in (10, 10, 15, 5):
        a.step(n); b.step(n)
    sa, sb, sta, stb = a.state(), b.state(), a.stats(), b.stats()
    assert sta["sort_interval"] > 1 and stb["sort_interval"] == 1 and sta["sort_violations"] == 0
    for k in XVF:
        assert rel_err(sa[k], sb[k]) < 1e-10, (k, rel_err(sa[k], sb[k]))
    for k in ("engVdW", "engCoul", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-11 * abs(stb[k]) + 1e-12, (k, sta[k], stb[k])
    for k in ("posCross", "negCross"):
        assert sta[k] == stb[k]
    assert max(np.abs(sa[k]).max() for k in ("x", "y", "z")) < max(case["box"]) and min(sa[k].min() for k in ("x", "y", "z")) >= 0.0


def test_slack_violation_at_full_size():
    """1 000 188 atoms at 3 000 K with the interval held at 16 steps (debug bit 8192) although the atoms use up the 0.05 A slack in two or three: atoms
    DO leave the slack between two rebuilds, the list kernel stands down, and the clean-up launch (one residency's worth of workgroups striding over all
    74 088 cells) stages every cell with the wider stencil.  Same trajectory as the every-step schedule, summation order aside."""
    case = inputs.lj_case((63, 63, 63), seed=20240502, vel_T=3000.0)
    a = api.Engine(api.Model.from_case(case), sort_every=16, debug=8192)
    b = api.Engine(api.Model.from_case(case), sort_every=1)
    for n in (10, 22, 5):
        a.step(n); b.step(n)
    sa, sb, sta, stb = a.state(), b.state(), a.stats(), b.stats()
    assert sta["sort_interval"] == 16 and sta["sort_violations"] > 0 and sta["pair_lists"] == 1, sta
    for k in XVF:
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
    for k in ("engVdW", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-10 * abs(stb[k]) + 1e-12, (k, sta[k], stb[k])
    for k in ("posCross", "negCross"):
        assert sta[k] == stb[k]


@pytest.mark.parametrize("cell", [2.2, 3.1])
def test_multi_workgroup_scan_on_a_fine_grid(cell):
    """40 000 atoms (BASELINE config 2) on 2.2 A / 3.1 A cells: 52 x 52 x 65 = 175 760 cells, most of them empty - k_scan_totals +
    k_scan_apply (172 chunks) against the host prefix sum, forces against the oracle."""
    case = inputs.config("C2")
    e = api.Engine(api.Model.from_case(case), cell_size=cell)
    s, st = e.state(), e.stats()
    assert st["n_cells"] > 16384
    dims = check_cell_table(e, s, case["box"])
    assert dims[0] * dims[1] * dims[2] == st["n_cells"]
    o = oracle.Oracle(case)
    o.forces(1)
    so = o.state()
    for k in ("fx", "fy", "fz"):
        assert rel_err(s[k], so[k]) < 1e-11, (k, rel_err(s[k], so[k]))
    e.step(7)
    s = e.state()
    check_cell_table(e, s, case["box"])
    o.step(7)
    for k in XVF:
        assert rel_err(s[k], o.state()[k]) < 1e-9, k


def test_forces_call_leaves_wall_counters_alone():
    """aztot_forces after aztot_step collects the energies only: wall momenta, crossing counts and dropped pairs of the previous
    step window must not be added a second time (k_finalize)."""
    case = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, vel_T=3000.0)
    e = api.Engine(api.Model.from_case(case))
    e.step(40)
    a = e.stats()
    assert sum(a["posCross"]) + sum(a["negCross"]) > 20
    e.forces()
    b = e.stats()
    for k in ("posCross", "negCross", "posMom", "negMom", "pairs_dropped"):
        assert a[k] == b[k], (k, a[k], b[k])
    e.forces()
    e.step(25)
    c = e.stats()
    o = oracle.Oracle(case)
    o.forces(1)
    o.step(65)
    so = o.stats()
    assert c["negCross"] + c["posCross"] == [so["cross"][k] for k in (0, 2, 4, 1, 3, 5)]
    mom = [so["momXn"], so["momYn"], so["momZn"], so["momXp"], so["momYp"], so["momZp"]]
    assert rel_err(c["negMom"] + c["posMom"], mom) < 1e-9


def test_set_state_after_graph_replay_keeps_thermostat_state_attached():
    """U / radius set AFTER the step graph was captured: the sort must start carrying them (the captured graph had the old
    carry mode baked in and is rebuilt)."""
    case = inputs.lj_case((6, 6, 6), a=5.3, seed=17, rc=7.0, cell_list=7.0, vel_T=600.0)
    n = len(case["types"])
    e = api.Engine(api.Model.from_case(case), use_graph=1)
    e.step(6)                                    # captures + replays
    U = np.arange(n, dtype=float) + 0.25
    rad = 2.0 + np.arange(n, dtype=float) / n
    e.set_state(U=U, radius=rad)
    e.step(40)                                   # atoms change cells: a stale graph would scramble U against the ids
    s = e.state()
    assert np.array_equal(s["U"], U) and np.array_equal(s["radius"], rad)


# ---------------------------------------------------------------------------------------------------
# the reference's shipped example inputs, read by aztot_init_md from the four files
# ---------------------------------------------------------------------------------------------------
def test_case_study_1_through_the_hip_path(tmp_path):
    """BASELINE config 1 verbatim: 40 000 Ar in a 1141.5 A box, 'temperature 298 radi', 'cell_list 85', 'elec fenn' demoted to none
    (no charged species).  20 steps against the oracle: forces are exactly 0 (no pair inside 4 A), the thermostat moves v, U, radius."""
    d = materialise_case_study(1, str(tmp_path / "cs1"), nstep=20)
    m = api.Model.from_dir(d)
    assert int(m.query("n_atoms")[0]) == 40000 and int(m.query("tstat_type")[0]) == 2 and int(m.query("elec_type")[0]) == 0
    e = api.Engine(m)
    case = case_from_parsed(parse.parse_dir(d))
    o = oracle.Oracle(case)
    o.forces(1)
    s0 = e.state()
    assert np.array_equal(s0["radius"], o.state()["rad"])               # initial radii: same table, same draw per atom id
    e.step(20)
    o.step(20)
    s, so, st, sto = e.state(), o.state(), e.stats(), o.stats()
    assert st["n_cells"] == 13 ** 3
    assert all(np.all(s[k] == 0.0) for k in ("fx", "fy", "fz")) and st["engVdW"] == 0.0
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        assert rel_err(s[k], so[k]) < 1e-9, (k, rel_err(s[k], so[k]))
    assert rel_err(s["U"], so["U"]) < 1e-9
    # field.txt of case study 1 has no 'radii' section: radA = radB = mxEng = 0, so the radius law 0 / (0 - min(U, 0)) is 0 / 0 for every
    # atom - on both sides (nothing reads the radii of a Lennard-Jones run)
    assert np.all(np.isnan(s["radius"])) and np.all(np.isnan(so["rad"]))
    for a, b in (("engKin", "engKin"), ("engTemp", "engTemp"), ("engTot", "engTot")):
        assert abs(st[a] - sto[b]) <= 1e-10 * abs(sto[b]), (a, st[a], sto[b])
    assert st["negCross"] + st["posCross"] == [sto["cross"][k] for k in (0, 2, 4, 1, 3, 5)]


def test_case_study_2_through_the_hip_path(tmp_path):
    """BASELINE config 5 verbatim: 4 000 atoms, radius-dependent 'surk' potential rc 6.0 on 2.7 A cells (12^3 cells of 2.92 A, 7^3
    stencil), radiative thermostat at 500 K + equilibration rescaling.  The radii the thermostat writes feed the next step's forces."""
    d = materialise_case_study(2, str(tmp_path / "cs2"), nstep=30)
    m = api.Model.from_dir(d)
    e = api.Engine(m)
    p = parse.parse_dir(d)
    assert p["vdw"][0][0]["type"] == 7 and p["cell_list"] == 2.7
    case = case_from_parsed(p)
    case["freqEq"] = 5                      # the file's eqfreq 2500 would never fire in a 30-step test ...
    case["nEq"] = 20
    e.close()
    e = api.Engine(api.Model.from_case(case))          # ... so the schedule is shortened; everything else is the file's
    o = oracle.Oracle(case)
    o.forces(1)
    s, so = e.state(), o.state()
    for k in ("fx", "fy", "fz"):
        assert rel_err(s[k], so[k]) < 1e-11, k
    assert e.stats()["n_cells"] == 12 ** 3
    e.step(30)
    o.step(30)
    s, so, st, sto = e.state(), o.state(), e.stats(), o.stats()
    for k in XVF:
        assert rel_err(s[k], so[k]) < 1e-8, (k, rel_err(s[k], so[k]))
    assert rel_err(s["U"], so["U"]) < 1e-8 and rel_err(s["radius"], so["rad"]) < 1e-9
    for a, b in (("engVdW", "engVdW"), ("engKin", "engKin"), ("engTemp", "engTemp"), ("engTot", "engTot")):
        assert abs(st[a] - sto[b]) <= 1e-9 * abs(sto[b]) + 1e-10, (a, st[a], sto[b])
    # the files themselves (eqfreq 2500 untouched) run too and agree with the array entry point for the first steps
    e1 = api.Engine(m)
    e2 = api.Engine(api.Model.from_case(case_from_parsed(p)))
    e1.step(12); e2.step(12)
    for k in ("x", "vx", "fx", "U", "radius"):
        assert np.array_equal(e1.state()[k], e2.state()[k]), k


@pytest.mark.parametrize("name", ["C4T", "C3T"])
def test_thermalised_liquid_at_full_size(name):
    """BASELINE configs 4 and 3 with Maxwell velocities at argon's 85 K (`init_vel gaus`; the reference rebuilds its cell list every step, main.cu:300-326, so
    its cost does not depend on temperature - the lazy schedule's does): the DEFAULT engine (automatic skin, pair lists, adaptive interval - capped at 4 steps
    here so that 18 steps of the serial reference span the measuring phase and two and a half sort intervals) against the reference's serial code at all
    1 000 188 atoms, per-atom x / v / f to 1e-9 (north star), energies to 1e-11."""
    case = inputs.config(name)
    ref, eref, who = cpu_steps(case, 18)
    e = api.Engine(api.Model.from_case(case), initial_forces=0, sort_every=4)
    e.step(8)                   # the engine measures the atoms' speed (interval 1)
    r0 = e.stats()["rebuilds"]
    e.step(10)                  # interval 4: rebuilds at two or three of these steps, lists walked on the others
    s, st = e.state(), e.stats()
    assert st["sort_interval"] == 4 and st["pair_lists"] == 1 and st["sort_violations"] == 0 and st["cells_without_list"] == 0, st
    assert 2 <= st["rebuilds"] - r0 <= 3 and 0.25 < st["skin"] < 0.5 and st["n_cells"] < 42 ** 3, st
    assert 30.0 < st["temperature"] < 90.0, st["temperature"]
    for k in XVF:
        assert rel_err(s[k], ref[k]) < 1e-9, (name, who, k, rel_err(s[k], ref[k]))
    # atom by atom (every force and velocity against its own magnitude), 18 steps into a thermal trajectory
    assert per_atom_err(s, ref, FRC) < 1e-9, (name, who, per_atom_err(s, ref, FRC))
    assert per_atom_err(s, ref, VEL) < 1e-9, (name, who, per_atom_err(s, ref, VEL))
    for k, v in eref.items():
        assert abs(st[k] - v) <= 1e-11 * abs(v) + 1e-12, (name, who, k, st[k], v)
    assert st["pairs_dropped"] == 0
    # and the run goes on at the interval the speeds allow: a thermalised liquid keeps its lists for ten steps and more
    f = api.Engine(api.Model.from_case(case))
    for n in (10, 40, 40):
        f.step(n)
    stf = f.stats()
    assert stf["sort_interval"] >= 10 and stf["sort_violations"] == 0 and stf["pair_lists"] == 1 and stf["cells_without_list"] == 0, stf


def test_random_call_pattern_at_full_size():
    """the call-pattern fuzz (tests/test_gpu_call_patterns.py) at the size the bench runs: C4T, 1 000 188 atoms - the default engine (adaptive interval, pair lists,
    no clean-up launch with 100 MB snapshots, call ends deferred to the next look or read) against the every-step schedule under random aztot_step sizes,
    single-step loops, statistics reads, aztot_forces, a heating kick and an in-place restart"""
    rng = np.random.default_rng(7)
    case = inputs.config("C4T")
    a = api.Engine(api.Model.from_case(case))
    b = api.Engine(api.Model.from_case(case), sort_every=1)
    total = 0
    while total < 260:
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
            sa, sb = a.stats(), b.stats()
            for k in ("engTot", "engKin", "engVdW"):
                assert abs(sa[k] - sb[k]) <= 1e-10 * abs(sb[k]), (total, k, sa[k], sb[k])
        elif op == "forces":
            a.forces(); b.forces()
        elif op == "heat":
            f = float(rng.uniform(0.95, 1.15))
            for e in (a, b):
                s = e.state(("vx", "vy", "vz"))
                e.set_state(**{k: s[k] * f for k in ("vx", "vy", "vz")})
        else:
            for e in (a, b):
                s = e.state(XVF)
                c = e.clock()
                e.set_state(**{k: s[k] for k in XVF})
                e.set_clock(**c)
    sa, sb, xa, xb = a.stats(), b.stats(), a.state(XVF), b.state(XVF)
    assert sa["step"] == sb["step"] == total and sa["sort_interval"] > 1 and sb["sort_interval"] == 1
    for k in XVF:
        assert rel_err(xa[k], xb[k]) < 1e-9, (k, rel_err(xa[k], xb[k]))
    assert per_atom_err(xa, xb, FRC) < 1e-8
    for k in ("engTot", "engKin", "engVdW"):
        assert abs(sa[k] - sb[k]) <= 1e-10 * abs(sb[k]), (k, sa[k], sb[k])
    assert sa["posCross"] == sb["posCross"] and sa["negCross"] == sb["negCross"]
