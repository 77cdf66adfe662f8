"""GPU parity tests (run with -m gpu on an MI355X): the HIP hot path, called through the C ABI, against the
CPU oracle (oracle/, a restatement of the reference's serial path pinned to the reference's own compiled code)
and against the committed golden vectors that the reference binary generated.

Tolerances (fp64): forces/positions/velocities <= 1e-11 array-relative after a force evaluation, <= 1e-9 after
tens of steps (north star: 1e-9); energies <= 1e-12 relative.  Differences come only from summation order and
FMA contraction; the arithmetic per pair follows the serial reference.
"""
import os

import numpy as np
import pytest

from aztotmd_amd import api, inputs
from oracle import oracle
from util import FRC, VEL, add_random_dynamics, family_with_coulomb, mixed_case, per_atom_err, random_case, rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EKEYS = ("engVdW", "engElec3", "engKin", "engTot", "engElecField", "Temp", "momXn", "momXp", "momYn", "momYp", "momZn", "momZp")
FKEYS = ("fx", "fy", "fz")


def engine(case, **kw):
    return api.Engine(api.Model.from_case(case), **kw)


def check_forces(case, tol=1e-11, **kw):
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    e = engine(case, **kw)
    s, st = e.state(), e.stats()
    for k in FKEYS:
        assert rel_err(s[k], so[k]) < tol, (k, rel_err(s[k], so[k]))
    assert per_atom_err(s, so, FRC) < 1e-9, per_atom_err(s, so, FRC)        # every atom's force against its own magnitude (north star: 1e-9)
    assert abs(st["engVdW"] - sto["engVdW"]) <= 1e-12 * abs(sto["engVdW"]) + 1e-14
    assert abs(st["engCoul"] - sto["engElec3"]) <= 1e-12 * abs(sto["engElec3"]) + 1e-14
    return e, o


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("name", ["F1", "F2", "F3"])
def test_initial_forces_match_oracle(name, variant):
    check_forces(inputs.config(name), pair_variant=variant)


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("pot", ["buck", "bmhs", "p746", "elin", "einv", "lnjs+dir", "lnjs+fenn+field"])
def test_potential_families(pot, variant):
    check_forces(mixed_case(pot), pair_variant=variant)


@pytest.mark.parametrize("elec", ["dir", "fenn", "ewald"])
@pytest.mark.parametrize("family", ["lnjs", "buck", "p746", "bmhs"])
def test_family_kernels_with_coulomb(family, elec):
    """the specialised tile kernels (one potential family x none/direct/Fennell/Ewald, branch-free bodies, erfcx fit, own exp and
    rsqrt) against the oracle's libm arithmetic, and bit-for-bit energy agreement is NOT expected: 1e-11 forces, 1e-12 energies."""
    case = family_with_coulomb(family, elec)
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    for variant in (2, 1):
        e = engine(case, pair_variant=variant)
        s, st = e.state(), e.stats()
        for k in FKEYS:
            assert rel_err(s[k], so[k]) < 1e-11, (variant, k, rel_err(s[k], so[k]))
        assert abs(st["engVdW"] - sto["engVdW"]) <= 1e-12 * abs(sto["engVdW"]) + 1e-14
        assert abs(st["engCoul"] - sto["engElec3"]) <= 1e-12 * abs(sto["engElec3"]) + 1e-14
    e.step(20)
    o.step(20)
    for k in ("x", "vx", "fx"):
        assert rel_err(e.state()[k], o.state()[k]) < 1e-9, k


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("cell", [2.2, 3.3, 4.7, 13.0])
def test_cells_smaller_or_larger_than_cutoff(cell, variant):
    """control.txt 'cell_list' below the cut-off (case study 2: 2.7 vs rc 6.0) widens the stencil; above it, coarsens."""
    check_forces(inputs.config("F1"), pair_variant=variant, cell_size=cell)


def test_golden_trajectory_F1():
    """50 steps against the reference binary's own trajectory (tests/golden/F1_lj.npz)."""
    z = np.load(os.path.join(G, "F1_lj.npz"))
    case = inputs.config("F1")
    for k in ("x", "y", "z"):
        assert np.array_equal(case[k], z["in_" + k])
    e = engine(case)
    done = 0
    for st in (0, 1, 10, 50):
        e.step(st - done)
        done = st
        s, stt = e.state(), e.stats()
        ref = dict(zip(EKEYS, z["e_%d" % st].tolist()))
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
            assert rel_err(s[k], z["%s_%d" % (k, st)]) < 1e-10, (st, k, rel_err(s[k], z["%s_%d" % (k, st)]))
        zz = {k: z["%s_%d" % (k, st)] for k in FRC + VEL}
        assert per_atom_err(s, zz, FRC) < 1e-9, (st, per_atom_err(s, zz, FRC))
        if st:
            assert per_atom_err(s, zz, VEL) < 1e-9, (st, per_atom_err(s, zz, VEL))
        assert abs(stt["engVdW"] - ref["engVdW"]) < 1e-12 * abs(ref["engVdW"])
        if st:
            assert abs(stt["engKin"] - ref["engKin"]) < 1e-11 * abs(ref["engKin"])
            assert abs(stt["engTot"] - ref["engTot"]) < 1e-12 * abs(ref["engTot"])
            assert abs(stt["temperature"] - ref["Temp"]) < 1e-11 * abs(ref["Temp"])


@pytest.mark.parametrize("name,kw", [("F2_lj", {}), ("F3_fennel", dict(charges=(0.2, -0.2), elec="fenn"))])
def test_golden_trajectory_4000(name, kw):
    z = np.load(os.path.join(G, name + ".npz"))
    case = inputs.lj_case((10, 10, 10), a=5.26, seed=12345, **kw)
    e = engine(case)
    s = e.state()
    for k in FKEYS:
        assert rel_err(s[k], z[k + "_0"]) < 1e-11
    e.step(50)
    s, st = e.state(), e.stats()
    ref = dict(zip(EKEYS, z["e_50"].tolist()))
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(s[k], z[k + "_50"]) < 1e-9, (k, rel_err(s[k], z[k + "_50"]))
    zz = {k: z[k + "_50"] for k in FRC + VEL}
    assert per_atom_err(s, zz, FRC) < 1e-9 and per_atom_err(s, zz, VEL) < 1e-9, (per_atom_err(s, zz, FRC), per_atom_err(s, zz, VEL))
    for a, b in (("engVdW", "engVdW"), ("engCoul", "engElec3"), ("engKin", "engKin"), ("engTot", "engTot")):
        assert abs(st[a] - ref[b]) <= 1e-11 * abs(ref[b]) + 1e-13, (a, st[a], ref[b])


def test_golden_equilibration_scaling():
    """nequil 20 / eqfreq 5: the velocity rescaling branch of integrate2 (integrators.cpp:511-522)."""
    z = np.load(os.path.join(G, "F1_tscale.npz"))
    case = inputs.config("F1")
    case.update(nEq=20, freqEq=5, T=85.0)
    for k in ("vx", "vy", "vz"):
        case[k] = z["in_" + k]
    e = engine(case)
    e.step(5)
    s, st = e.state(), e.stats()
    ref = dict(zip(EKEYS, z["e_5"].tolist()))
    assert abs(st["engKin"] - ref["engKin"]) < 1e-12 * ref["engKin"]
    for k in ("vx", "x", "fx"):
        assert rel_err(s[k], z[k + "_5"]) < 1e-10
    e.step(25)
    s, st = e.state(), e.stats()
    ref = dict(zip(EKEYS, z["e_30"].tolist()))
    for k in ("vx", "vy", "vz", "x", "fz"):
        assert rel_err(s[k], z[k + "_30"]) < 1e-9
    assert abs(st["engTot"] - ref["engTot"]) < 1e-11 * abs(ref["engTot"])


def test_golden_nose_hoover():
    """Nose-Hoover thermostat + equilibration rescaling against the reference binary's trajectory (tests/golden/F1_nose.npz)."""
    z = np.load(os.path.join(G, "F1_nose.npz"))
    case = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, T=120.0, vel_T=80.0)
    case.update(tstat_type=1, tau=0.05, nEq=10, freqEq=5)
    for k in ("vx", "x"):
        assert np.array_equal(case[k], z["in_" + k])
    e = engine(case)
    done = 0
    for st in (1, 10, 40):
        e.step(st - done)
        done = st
        stt = e.stats()
        ref = dict(zip(EKEYS, z["e_%d" % st].tolist()))
        assert abs(stt["engKin"] - ref["engKin"]) < 1e-11 * abs(ref["engKin"]), (st, stt["engKin"], ref["engKin"])
        assert abs(stt["engTot"] - ref["engTot"]) < 1e-11 * abs(ref["engTot"])
        if ("x_%d" % st) in z:
            s = e.state()
            for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
                assert rel_err(s[k], z["%s_%d" % (k, st)]) < 1e-9, (st, k)
    o = oracle.Oracle(case)
    o.forces(0)
    o.step(40)
    assert abs(e.stats()["nose_chit"] - o.stats()["chit"]) < 1e-10 * abs(o.stats()["chit"])
    assert abs(e.stats()["nose_conint"] - o.stats()["conint"]) < 1e-10 * abs(o.stats()["conint"])


# ---------------------------------------------------------------------------------------------------
# bonds + angles ("next" row f2)
# ---------------------------------------------------------------------------------------------------
BKEYS = EKEYS + ("engBond", "engAngle")


@pytest.mark.parametrize("kw", [{}, dict(charges=(-0.2, 0.1), elec="fenn"), dict(cell_list=2.6)])
def test_bonded_forces_match_oracle(kw):
    """aztot_forces on a molecular liquid: pair + bond + angle forces and the two bonded energies vs the oracle."""
    case = inputs.molecular_case((8, 8, 9), seed=3, **kw)
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    e = engine(case, cell_size=kw.get("cell_list", 0.0))
    s0 = e.state()
    o2 = oracle.Oracle(case)
    o2.forces(2)                                  # the state after init: pair forces only (sys_init.cpp:1181-1184)
    for k in FKEYS:
        assert rel_err(s0[k], o2.state()[k]) < 1e-11, k
    assert e.stats()["engBond"] == 0.0 and e.stats()["engAngle"] == 0.0
    e.forces()
    s, st = e.state(), e.stats()
    for k in FKEYS:
        assert rel_err(s[k], so[k]) < 1e-11, (k, rel_err(s[k], so[k]))
    for a, b in (("engBond", "engBond"), ("engAngle", "engAngle"), ("engVdW", "engVdW"), ("engCoul", "engElec3")):
        assert abs(st[a] - sto[b]) <= 1e-12 * abs(sto[b]) + 1e-14, (a, st[a], sto[b])
    assert abs(np.sum(s["fx"])) < 1e-9 and abs(np.sum(s["fy"])) < 1e-9           # Newton's third law over all terms


@pytest.mark.parametrize("name", ["M1_bonded", "M1_bonded_fenn"])
def test_golden_bonded_trajectory(name):
    """3000-atom molecular liquid against the reference binary's trajectory (5 bond potentials, hcos angles)."""
    z = np.load(os.path.join(G, name + ".npz"))
    kw = dict(charges=(-0.2, 0.1), elec="fenn") if name.endswith("fenn") else {}
    case = inputs.molecular_case((10, 10, 10), **kw)
    for k in ("x", "vx"):
        assert np.array_equal(case[k], z["in_" + k])
    e = engine(case)
    done = 0
    for st in z["steps"].tolist()[1:]:
        e.step(st - done)
        done = st
        stt = e.stats()
        ref = dict(zip(BKEYS, z["e_%d" % st].tolist()))
        for a, b in (("engBond", "engBond"), ("engAngle", "engAngle"), ("engVdW", "engVdW"), ("engCoul", "engElec3"), ("engKin", "engKin"), ("engTot", "engTot")):
            assert abs(stt[a] - ref[b]) <= 1e-10 * abs(ref[b]) + 1e-13, (st, a, stt[a], ref[b])
        if ("x_%d" % st) in z:
            s = e.state()
            for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
                assert rel_err(s[k], z["%s_%d" % (k, st)]) < 1e-9, (st, k)


def test_bonded_input_files_and_energy_conservation(tmp_path):
    """from_dir (field.txt + bonds.txt + angles.txt) == from_case bit for bit; the NVE total energy of the molecular liquid stays put."""
    case = inputs.molecular_case((8, 8, 8), seed=21, vel_T=None)       # control.txt carries 'init_vel zero'
    d = str(tmp_path / "mol")
    inputs.write_input_files(case, d)
    a, b = api.Engine(api.Model.from_dir(d)), engine(case)
    a.step(300); b.step(300)
    sa, sb = a.state(), b.state()
    for k in ("x", "vx", "fx"):
        assert np.array_equal(sa[k], sb[k]), k
    e1 = b.stats()["engTot"]
    b.step(1700)
    e2 = b.stats()["engTot"]
    assert abs(e2 - e1) < 0.02 * abs(b.stats()["engKin"]), (e1, e2)      # bounded Verlet fluctuation (omega dt = 0.27 for the L-C stretch)


# ---------------------------------------------------------------------------------------------------
# Ewald sum ('elec pme'; "next" row f4)
# ---------------------------------------------------------------------------------------------------
def ewald_case(**kw):
    c = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, charges=(0.4, -0.4), elec="fenn", r_real=6.5, alpha=0.45, vel_T=80.0)
    c.update(elec_type=2, ewald_k=(6, 6, 6))
    c.update(kw)
    return c


def test_ewald_with_more_than_64k_of_lds():
    """kx + ky + kz = 57 harmonics: the force kernel needs 69 KiB of dynamic LDS, above the 64 KiB a launch gets without asking
    (hipFuncSetAttribute in upload_ewald; before, the launch was rejected silently and the reciprocal forces were missing)."""
    case = ewald_case(ewald_k=(19, 19, 19))
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    e = engine(case)
    s, st = e.state(), e.stats()
    for k in FKEYS:
        assert rel_err(s[k], so[k]) < 1e-11, (k, rel_err(s[k], so[k]))
    assert abs(st["engCoulRec"] - sto["engElec2"]) <= 1e-12 * abs(sto["engElec2"])
    assert sto["engElec2"] != 0.0


@pytest.mark.parametrize("kw", [{}, dict(ewald_k=(4, 7, 9)), dict(cell_list=3.4)])
def test_ewald_forces_match_oracle(kw):
    """reciprocal + real-space + constant parts of the Ewald sum against the oracle (forces 1e-11, energies 1e-12)."""
    case = ewald_case(**kw)
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    e = engine(case, cell_size=kw.get("cell_list", 0.0))
    s, st = e.state(), e.stats()
    for k in FKEYS:
        assert rel_err(s[k], so[k]) < 1e-11, (k, rel_err(s[k], so[k]))
    for a, b in (("engCoulRec", "engElec2"), ("engCoulConst", "engElec1"), ("engCoul", "engElec3"), ("engVdW", "engVdW")):
        assert abs(st[a] - sto[b]) <= 1e-12 * abs(sto[b]) + 1e-14, (a, st[a], sto[b])


def test_ewald_anisotropic_box_with_tie_on_the_cutoff():
    case = inputs.lj_case((6, 5, 4), a=5.4, seed=2, rc=5.2, cell_list=5.2, charges=(0.5, -0.5), elec="fenn", r_real=5.2, alpha=0.5, vel_T=200.0)
    case.update(elec_type=2, ewald_k=(7, 5, 6))
    o = oracle.Oracle(case)
    o.forces(0)
    e = engine(case)
    for k in FKEYS:
        assert rel_err(e.state()[k], o.state()[k]) < 1e-11
    assert abs(e.stats()["engCoulRec"] - o.stats()["engElec2"]) < 1e-12 * abs(o.stats()["engElec2"])
    e.step(25); o.step(25)
    for k in ("x", "vx", "fx"):
        assert rel_err(e.state()[k], o.state()[k]) < 1e-9, k


def test_golden_ewald_trajectory(tmp_path):
    """tests/golden/E1_ewald.npz (the reference's compiled ewald_rec in the loop), via the arrays and via 'elec pme' in control.txt."""
    z = np.load(os.path.join(G, "E1_ewald.npz"))
    case = ewald_case()
    assert np.array_equal(case["x"], z["in_x"])
    keys = EKEYS + ("engBond", "engAngle", "engElec1", "engElec2")
    e = engine(case)
    s0 = e.state()
    for k in FKEYS:
        assert rel_err(s0[k], z[k + "_0"]) < 1e-11
    done = 0
    for st in z["steps"].tolist()[1:]:
        e.step(st - done)
        done = st
        stt = e.stats()
        ref = dict(zip(keys, z["e_%d" % st].tolist()))
        for a, b in (("engCoulRec", "engElec2"), ("engCoulConst", "engElec1"), ("engCoul", "engElec3"), ("engVdW", "engVdW"), ("engKin", "engKin"), ("engTot", "engTot")):
            assert abs(stt[a] - ref[b]) <= 1e-10 * abs(ref[b]) + 1e-13, (st, a, stt[a], ref[b])
    s = e.state()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(s[k], z[k + "_40"]) < 1e-9, k
    case0 = ewald_case(vx=np.zeros(500), vy=np.zeros(500), vz=np.zeros(500))
    d = str(tmp_path / "pme")
    inputs.write_input_files(case0, d)
    a, b = api.Engine(api.Model.from_dir(d)), engine(case0)
    a.step(20); b.step(20)
    assert np.array_equal(a.state()["fx"], b.state()["fx"]) and a.stats()["engCoulRec"] == b.stats()["engCoulRec"]


def test_ewald_graph_equals_eager_and_conserves_energy():
    case = ewald_case()
    a, b = engine(case, use_graph=1), engine(case, use_graph=0)
    a.step(200); b.step(200)
    assert np.array_equal(a.state()["x"], b.state()["x"])
    e0 = a.stats()["engTot"]
    a.step(800)
    assert abs(a.stats()["engTot"] - e0) < 2e-2 * a.stats()["engKin"]        # truncation noise of rc 6.5 at alpha 0.45 (erfc(2.9) = 4e-5), not drift


def test_wall_crossing_counters_and_field():
    """hot gas: atoms cross the periodic walls; wall momenta, crossing counts and field energy vs the oracle."""
    case = mixed_case("lnjs+fenn+field", vel_T=3000.0)
    o = oracle.Oracle(case)
    o.forces(0)
    o.step(40)
    e = engine(case)
    e.step(40)
    st, sto = e.stats(), o.stats()
    assert sum(sto["cross"]) > 20
    assert [st["negCross"][0], st["posCross"][0], st["negCross"][1], st["posCross"][1], st["negCross"][2], st["posCross"][2]] == sto["cross"]
    sc = e.species_crossings()                    # per species (the columns of msd.dat): Xn, Xp, Yn, Yp, Zn, Zp
    assert sc.shape == (2, 6) and np.array_equal(sc, o.species_crossings()) and sc.sum(axis=0).tolist() == sto["cross"] and (sc.sum(axis=1) > 0).all()
    mom = [st["negMom"][0], st["posMom"][0], st["negMom"][1], st["posMom"][1], st["negMom"][2], st["posMom"][2]]
    ref = [sto[k] for k in ("momXn", "momXp", "momYn", "momYp", "momZn", "momZp")]
    assert rel_err(mom, ref) < 1e-11
    assert abs(st["engElecField"] - sto["engElecField"]) < 1e-11 * abs(sto["engElecField"])
    s, so = e.state(), o.state()
    for k in ("x", "vx", "fz"):
        assert rel_err(s[k], so[k]) < 1e-9
    for k in ("x", "y", "z"):
        assert s[k].min() >= 0.0 and (s[k] < np.array(case["box"])["xyz".index(k)]).all()


@pytest.mark.parametrize("variant", [1, 2])
def test_radiative_thermostat_matches_oracle(variant):
    """tstat_radi9 restated with the counter RNG: GPU vs CPU oracle, incl. photon table, U and radii."""
    case = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, T=298.0, tstat="radi", vel_T=150.0,
                          radii=[(2.73, 4.731, 0.2)], nEq=10, freqEq=5)
    o = oracle.Oracle(case)
    o.forces(0)
    e = engine(case, pair_variant=variant)
    ph = e.model.query("photons", seed=12345)
    assert np.array_equal(ph, o.photons())
    assert np.allclose(e.state()["radius"], o.state()["rad"], rtol=0, atol=0)
    for nst in (1, 9, 20):
        o.step(nst)
        e.step(nst)
        s, so, st, sto = e.state(), o.state(), e.stats(), o.stats()
        for a, b in (("vx", "vx"), ("vy", "vy"), ("vz", "vz"), ("x", "x"), ("U", "U"), ("radius", "rad"), ("fx", "fx")):
            assert rel_err(s[a], so[b]) < 1e-9, (nst, a, rel_err(s[a], so[b]))
        assert abs(st["engTemp"] - sto["engTemp"]) <= 1e-10 * abs(sto["engTemp"])
        assert abs(st["engKin"] - sto["engKin"]) <= 1e-10 * abs(sto["engKin"])
    assert so["U"].max() > 1e-4          # the radiate branch was exercised


def test_surk_radius_potential_with_thermostat():
    """'surk' radius-dependent potential fed by the thermostat's radii (case study 2 style), GPU vs oracle."""
    pos, box = inputs.fcc_positions((6, 6, 6), 5.8, 0.1, 5)
    N = len(pos)
    case = {"box": box.tolist(), "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
            "vdw": [(0, 0, 7, 6.0, [75.0, 8.0, 1.0, 1.0])], "radii": [(2.73, 4.731, 0.2)], "x": pos[:, 0].copy(), "y": pos[:, 1].copy(),
            "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 500.0, "tstat_type": 2,
            "cell_list": 2.7, "use_clist": 1, "elec_type": 0}
    o = oracle.Oracle(case)
    o.forces(0)
    e = engine(case)
    for nst in (0, 5, 25):
        o.step(nst)
        e.step(nst)
        s, so = e.state(), o.state()
        for a, b in (("fx", "fx"), ("fy", "fy"), ("vz", "vz"), ("U", "U"), ("radius", "rad")):
            assert rel_err(s[a], so[b]) < 1e-9, (nst, a, rel_err(s[a], so[b]))
        assert abs(e.stats()["engVdW"] - o.stats()["engVdW"]) <= 1e-10 * abs(o.stats()["engVdW"])


@pytest.mark.parametrize("cell", [2.7, 6.0])
def test_surk_tile_mode_equals_generic_kernel_and_oracle(cell):
    """the specialised one-species surk + radii tile mode (case study 2 / BASELINE config 5) against the generic kernel (debug bit 512),
    the per-atom kernel and the oracle: forces 1e-11, 20 thermostatted steps 1e-9; on the case study's 2.7 A cells (7^3 stencil) and on
    cut-off sized cells"""
    pos, box = inputs.fcc_positions((7, 7, 8), 2.9, 0.12, 9)
    N = len(pos)
    case = {"box": box.tolist(), "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
            "vdw": [(0, 0, 7, 6.0, [75.0, 8.0, 1.0, 1.0])], "radii": [(2.73, 4.731, 0.2)], "x": pos[:, 0].copy(), "y": pos[:, 1].copy(),
            "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 500.0, "tstat_type": 2,
            "cell_list": cell, "use_clist": 1, "elec_type": 0}
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    engines = [engine(case), engine(case, debug=512), engine(case, pair_variant=1)]
    for e in engines:
        s, st = e.state(), e.stats()
        for k in FKEYS:
            assert rel_err(s[k], so[k]) < 1e-11, (k, rel_err(s[k], so[k]))
        assert abs(st["engVdW"] - sto["engVdW"]) <= 1e-12 * abs(sto["engVdW"])
    o.step(20)
    for e in engines[:2]:
        e.step(20)
        s, so = e.state(), o.state()
        for a, b in (("x", "x"), ("vx", "vx"), ("fx", "fx"), ("U", "U"), ("radius", "rad")):
            assert rel_err(s[a], so[b]) < 1e-9, (a, rel_err(s[a], so[b]))
    # the two tile modes agree to round-off (same operation order per pair)
    for k in ("fx", "fy", "fz"):
        assert rel_err(engines[0].state()[k], engines[1].state()[k]) < 1e-13


def test_input_files_round_trip(tmp_path):
    """atoms.xyz / field.txt / control.txt / cuda.txt written in the reference grammar give the same run."""
    case = inputs.config("F3")
    case["nsteps"] = 7
    inputs.write_input_files(case, str(tmp_path))
    e1 = api.Engine(api.Model.from_dir(str(tmp_path)))
    e2 = engine(case)
    e1.step(7)
    e2.step(7)
    s1, s2 = e1.state(), e2.state()
    for k in ("x", "vx", "fx", "fy"):
        assert np.array_equal(s1[k], s2[k])


def test_dilute_gas_is_force_free():
    """case study 1 character: no pair inside the cut-off -> forces and energies exactly 0."""
    rng = np.random.Generator(np.random.PCG64(3))
    n = 12
    g = (np.stack(np.meshgrid(*[np.arange(n)] * 3, indexing="ij"), -1).reshape(-1, 3) + 0.5) * 30.0 + rng.uniform(-5, 5, (n ** 3, 3))
    N = len(g)
    case = {"box": [360.0] * 3, "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
            "vdw": [(0, 0, 1, 4.0, [0.01006, 3.3952])], "x": g[:, 0].copy(), "y": g[:, 1].copy(), "z": g[:, 2].copy(),
            "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 298.0, "cell_list": 85.0, "use_clist": 1, "elec_type": 3,
            "rReal": 8.0, "alpha": 0.4}
    e = engine(case)
    e.step(5)
    s, st = e.state(), e.stats()
    assert all(np.all(s[k] == 0.0) for k in FKEYS) and st["engVdW"] == 0.0 and st["engTot"] == 0.0
    assert np.array_equal(s["x"], case["x"])


def test_bitwise_reproducible_and_graph_equals_eager():
    """id-ordered cells make the result independent of atomic arrival order; hipGraph replay == eager launches."""
    case = inputs.config("F2")
    runs = []
    for kw in (dict(use_graph=1), dict(use_graph=0), dict(use_graph=1)):
        e = engine(case, **kw)
        e.step(21)
        runs.append(e.state())
    for k in ("x", "vx", "fx", "fz"):
        assert np.array_equal(runs[0][k], runs[1][k]) and np.array_equal(runs[0][k], runs[2][k])


@pytest.mark.parametrize("variant", [1, 2])
def test_second_half_kick_ownership(variant):
    """On plain NVE steps integrate2's work is done by the tile kernel's epilogue (small systems), by the next step's
    k_integrate1_bin (large systems; debug bit 256 forces that path here) or by k_integrate2 itself (debug bit 128): the three must
    agree bit for bit in x, v, f and in the energies, across graph replays (10 + 20 steps), odd step counts and repeated calls."""
    case = inputs.lj_case((6, 6, 6), a=5.3, seed=17, rc=7.0, cell_list=7.0, vel_T=150.0)
    ref = None
    for dbg in (128, 256, 0):
        e = engine(case, debug=dbg, pair_variant=variant)
        kin = []
        for n in (10, 20, 7, 1):
            e.step(n)
            kin.append(e.stats()["engKin"])
        s = e.state()
        if ref is None:
            ref = (kin, s)
            continue
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
            assert np.array_equal(s[k], ref[1][k]), (dbg, k)
        assert np.allclose(kin, ref[0], rtol=1e-13, atol=0), (dbg, kin, ref[0])       # summation order of the partials differs
    o = oracle.Oracle(case)
    o.forces(0)
    o.step(38)
    assert abs(kin[-1] - o.stats()["engKin"]) < 1e-11 * kin[-1]


@pytest.mark.parametrize("variant", [1, 2])
@pytest.mark.parametrize("mode", ["adaptive", "forced"])
def test_lazy_resort_is_exact(mode, variant):
    """lazy re-sort against the every-step schedule and the oracle on a hot liquid (3 000 K: atoms cross walls and change cells all the time).
    'adaptive': the default - the interval follows the largest step seen.  'forced': debug bit 8192 holds the interval at 32 steps although the atoms
    are far too fast for it, so atoms DO leave their cell's slack between sorts and the pair kernels fall back to the wider stencil - the result must
    not change.  60 steps in five calls; x / v / f 1e-9 against the oracle, wall counters and per-species crossings equal."""
    case = inputs.lj_case((7, 7, 7), a=5.4, seed=23, rc=6.5, cell_list=6.9, vel_T=3000.0 if mode == "adaptive" else 9000.0)
    case["dt"] = 0.001 if mode == "adaptive" else 0.002           # 'forced': up to 0.05 A per step against a slack of 0.08 A, held for 32 steps
    kw = dict(sort_every=32, debug=8192) if mode == "forced" else {}
    a = engine(case, pair_variant=variant, **kw)
    b = engine(case, pair_variant=variant, sort_every=1)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (1, 9, 20, 25, 5):
        a.step(n); b.step(n); o.step(n)
    sa, sb, so, sta, stb, sto = a.state(), b.state(), o.state(), a.stats(), b.stats(), o.stats()
    if mode == "forced":
        assert sta["sort_interval"] == 32 and sta["sort_violations"] > 0
    else:
        assert sta["sort_violations"] == 0
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], so[k]) < 1e-9, (k, rel_err(sa[k], so[k]))
    assert sta["negCross"] + sta["posCross"] == stb["negCross"] + stb["posCross"] == [sto["cross"][k] for k in (0, 2, 4, 1, 3, 5)]
    assert np.array_equal(a.species_crossings(), b.species_crossings())
    assert rel_err(sta["negMom"] + sta["posMom"], stb["negMom"] + stb["posMom"]) < 1e-10
    for k in ("engVdW", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-10 * abs(stb[k]), (k, sta[k], stb[k])
    a.forces()                                  # a force call between two sorts wraps and re-sorts without touching the counters
    s2 = a.state()
    for k in FKEYS:
        assert rel_err(s2[k], sb[k]) < 1e-10
    assert a.stats()["posCross"] == sta["posCross"]


@pytest.mark.parametrize("kind", ["lj", "lj_fennell", "buck", "elin", "part_unlisted", "dense_cells"])
def test_pair_lists_between_two_rebuilds(kind):
    """The steps between two rebuilds of the cell list walk the pair lists the rebuild recorded (k_pair_tile<BUILD> -> k_pair_list) instead of staging and
    filtering every cell again: same forces as the every-step schedule (summation order aside) and as the oracle.  'part_unlisted': debug bit 65536
    caps the lists at 14 iterations, so part of the cells keep no list and go through the clean-up launch of the staging kernel while the others
    walk their lists; 'dense_cells': cells of 3 rc hold ~108 atoms (> 64: no cell keeps a list, everything goes through the clean-up launch until the
    engine notices and stops recording)."""
    kw = {}
    if kind in ("lj", "part_unlisted"):
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=31, rc=7.5, cell_list=7.9, vel_T=120.0)
        if kind == "part_unlisted":
            kw = dict(debug=65536)
    elif kind == "lj_fennell":
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=32, rc=7.5, cell_list=7.9, vel_T=120.0, charges=(0.3, -0.3), elec="fenn", r_real=7.5, alpha=0.3)
    elif kind in ("buck", "elin"):           # one potential family in the LDS table / the generic switch-based body
        case = mixed_case(kind, n=8, seed=5)
    else:
        case = inputs.lj_case((15, 15, 15), a=5.6, seed=33, rc=5.5, cell_list=16.5, vel_T=120.0)        # 5 cells of 16.8 A per axis, 108 atoms each
    a = engine(case, pair_variant=2, **kw)
    b = engine(case, pair_variant=2, sort_every=1)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 30, 7):
        a.step(n); b.step(n); o.step(n)
    sa, sb, so, sta, stb, sto = a.state(), b.state(), o.state(), a.stats(), b.stats(), o.stats()
    assert sta["sort_interval"] > 1 and sta["sort_violations"] == 0, sta
    if kind == "dense_cells":
        assert sta["pair_lists"] == 0                     # nothing kept a list: the engine went back to staging every cell
    else:
        assert sta["pair_lists"] == 1
        if kind == "part_unlisted":
            assert 0 < sta["cells_without_list"] < sta["n_cells"], sta
        else:
            assert sta["cells_without_list"] == 0, sta
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], so[k]) < 1e-9, (k, rel_err(sa[k], so[k]))
    for k in ("engVdW", "engCoul", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-10 * max(abs(stb[k]), 1e-3), (k, sta[k], stb[k])
    assert abs(sta["engTot"] - sto["engTot"]) <= 1e-10 * abs(sto["engTot"])


@pytest.mark.parametrize("kind", ["lj", "lj_fennell", "part_unlisted", "hot"])
def test_next_step_fused_into_the_pair_kernel(kind):
    """On plain NVE steps of a lazy run that walks pair lists the pair kernel's epilogue also opens the next step (deferred second half-kick, first half-kick,
    drift, wall counters, displacement check: everything k_integrate1_bin<2> does), writing the new positions to a second set of coordinate arrays.  Same
    operations in the same order: positions, velocities and forces must be BIT-IDENTICAL to a run with the fusion switched off (debug bit 131072), whatever
    the pattern of calls (graph replay of whole cycles, eager remainders, single steps); wall counters equal, wall momenta to summation order.  'hot': a
    9 000 K gas held at a 32-step interval (debug bit 8192) - atoms leave the slack, the violation is flagged one step ahead by the epilogue and the
    clean-up launch (which carries the same epilogue) takes over."""
    kw = {}
    if kind in ("lj", "part_unlisted"):
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=41, rc=7.5, cell_list=7.9, vel_T=300.0)
        if kind == "part_unlisted":
            kw = dict(debug=65536)
    elif kind == "lj_fennell":
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=42, rc=7.5, cell_list=7.9, vel_T=300.0, charges=(0.3, -0.3), elec="fenn", r_real=7.5, alpha=0.3)
    else:
        case = inputs.lj_case((7, 7, 7), a=5.4, seed=23, rc=6.5, cell_list=6.9, vel_T=9000.0)
        case["dt"] = 0.002
        kw = dict(sort_every=32, debug=8192)
    a = engine(case, pair_variant=2, **kw)
    kb = dict(kw)
    kb["debug"] = kb.get("debug", 0) | 131072
    b = engine(case, pair_variant=2, **kb)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 1, 37, 2, 33, 64, 5):
        a.step(n); b.step(n); o.step(n)
        sa, sb = a.state(), b.state()
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
            assert np.array_equal(sa[k], sb[k]), (n, k, rel_err(sa[k], sb[k]))
    sta, stb, so, sto = a.stats(), b.stats(), o.state(), o.stats()
    assert sta["pair_lists"] == 1 and sta["sort_interval"] > 1
    if kind == "hot":
        assert sta["sort_violations"] > 0
    assert sta["negCross"] + sta["posCross"] == stb["negCross"] + stb["posCross"] == [sto["cross"][k] for k in (0, 2, 4, 1, 3, 5)]
    assert np.array_equal(a.species_crossings(), b.species_crossings())
    assert rel_err(sta["negMom"] + sta["posMom"], stb["negMom"] + stb["posMom"]) < 1e-10
    for k in ("engVdW", "engCoul", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-12 * max(abs(stb[k]), 1e-3), (k, sta[k], stb[k])
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], so[k]) < 1e-8, (k, rel_err(sa[k], so[k]))


@pytest.mark.parametrize("kind", ["lj", "lj_fennell", "buck_ewald", "surk"])
def test_pair_energies_only_where_somebody_can_see_them(kind):
    """The statistics of an aztot_step call are those of its last step (the reference prints energies every `stat` steps, cuStat.cu:308-330), so the
    list kernel of every other step of the call books no pair energies (k_pair_list<.., ENG = false>).  Nothing observable may change: positions,
    velocities and forces BIT-IDENTICAL to a run whose every step books them (debug bit 134217728), and the energies after every call equal to the last
    bit (the last step runs the same instantiation in both), whatever the pattern of calls; both agree with the oracle."""
    if kind == "lj":
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=61, rc=7.5, cell_list=7.9, vel_T=200.0)
    elif kind == "lj_fennell":
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=62, rc=7.5, cell_list=7.9, vel_T=200.0, charges=(0.3, -0.3), elec="fenn", r_real=7.5, alpha=0.3)
    elif kind == "buck_ewald":                # Buckingham table + real-space Ewald term (MODE 3) next to the reciprocal-space kernels
        case = family_with_coulomb("buck", "ewald", n=8, seed=7)
    else:                                     # one species, radius-dependent 'surk' potential with the radiative thermostat (MODE 4)
        pos, box = inputs.fcc_positions((7, 7, 7), 5.8, 0.1, 16)
        N = len(pos)
        case = {"box": box.tolist(), "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
                "vdw": [(0, 0, 7, 6.0, [75.0, 8.0, 1.0, 1.0])], "radii": [(2.73, 4.731, 0.2)], "x": pos[:, 0].copy(), "y": pos[:, 1].copy(),
                "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 300.0, "tstat_type": 2,
                "cell_list": 6.5, "use_clist": 1, "elec_type": 0}
    pv = {} if kind == "surk" else dict(pair_variant=2)
    a = engine(case, **pv)
    b = engine(case, debug=134217728, **pv)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 1, 37, 2, 33, 5):
        a.step(n); b.step(n); o.step(n)
        sa, sb, sta, stb = a.state(), b.state(), a.stats(), b.stats()
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
            assert np.array_equal(sa[k], sb[k]), (n, k, rel_err(sa[k], sb[k]))
        for k in ("engVdW", "engCoul", "engKin", "engTot"):
            assert sta[k] == stb[k], (n, k, sta[k], stb[k])
    assert sta["pair_lists"] == 1 and sta["sort_interval"] > 1
    so, sto = o.state(), o.stats()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], so[k]) < 1e-8, (k, rel_err(sa[k], so[k]))
    assert abs(sta["engTot"] - sto["engTot"]) <= 1e-9 * abs(sto["engTot"])


def test_sort_interval_runs_on_across_calls():
    """One GPU, pair lists: the interval between two rebuilds of the cell list does not end with an aztot_step call - 40 single-step calls rebuild the cells
    as rarely as one 40-step call does (counted through the kernel timers), the lists made at the last rebuild stay in force, and the trajectory is
    the same (summation order aside) as that of the single call and of the oracle.  aztot_set_state and aztot_forces end the interval: the next step
    rebuilds."""
    case = inputs.lj_case((8, 8, 8), a=5.6, seed=51, rc=7.5, cell_list=7.9, vel_T=60.0)
    a, b = engine(case, pair_variant=2), engine(case, pair_variant=2)
    o = oracle.Oracle(case)
    o.forces(1)
    a.step(10); b.step(10); o.step(10)            # the engines measure the atoms' speed and settle on an interval
    K = a.stats()["sort_interval"]
    assert K >= 8 and b.stats()["sort_interval"] == K and a.stats()["pair_lists"] == 1
    a.set_profile(1); b.set_profile(1)
    a.reset_kernel_times(); b.reset_kernel_times()
    for _ in range(40):
        a.step(1)
    b.step(40); o.step(40)
    ka, kb = a.kernel_times(), b.kernel_times()
    rebuilds_a = ka.get("integrate1_bin", {"calls": 0})["calls"]
    rebuilds_b = kb.get("integrate1_bin", {"calls": 0})["calls"]
    assert rebuilds_a <= 40 // K + 1 and rebuilds_b <= 40 // K + 1, (K, rebuilds_a, rebuilds_b)
    assert ka["pair_list"]["calls"] == 40 and "pair_tile" not in ka
    sa, sb, so = a.state(), b.state(), o.state()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], so[k]) < 1e-9, (k, rel_err(sa[k], so[k]))
    for k in ("engVdW", "engKin", "engTot"):
        assert abs(a.stats()[k] - b.stats()[k]) <= 1e-10 * abs(b.stats()[k])
    # a state upload ends the interval: the next step rebuilds the cells (and its lists) from what was uploaded
    a.reset_kernel_times()
    a.set_state(x=sa["x"], y=sa["y"], z=sa["z"])
    a.step(1)
    assert a.kernel_times()["integrate1_bin"]["calls"] == 1
    a.reset_kernel_times()
    a.forces()
    a.step(1)
    assert a.kernel_times()["integrate1_bin"]["calls"] == 1
    b.step(2); o.step(2)
    sa, so = a.state(), o.state()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], so[k]) < 1e-9, (k, rel_err(sa[k], so[k]))


def test_kick_and_radiative_thermostat_in_one_launch():
    """Radiative thermostat without equilibration scaling: k_integrate2 and k_post run as ONE launch (nothing global happens between the second half-kick and
    the thermostat), and when a plain step follows, the same launch opens it (k_boundary_radi: first half-kick, drift, wall counters, displacement check).
    Same operations in the same order: bit-identical to the two-launch form (debug bit 4194304) and to the form without the boundary kernel (33554432),
    per-atom internal energies and radii included, and equal to the oracle."""
    case = inputs.lj_case((8, 8, 8), a=5.6, seed=61, rc=7.5, cell_list=7.9, vel_T=60.0, T=60.0, tstat="radi", radii=[(2.73, 4.731, 0.2)])       # 5 cells per axis: the lazy re-sort engages
    # (debug bit 131072: without the pair kernel's fused epilogue, which since round 3 closes and opens such steps itself where no clean-up launch follows -
    #  test_radiative_thermostat_fused_into_the_pair_kernel; this test is about the boundary kernel that serves everywhere else)
    a = engine(case, pair_variant=2, debug=131072)
    b = engine(case, pair_variant=2, debug=4194304 | 131072)
    c = engine(case, pair_variant=2, debug=33554432 | 131072)      # one launch for kick + thermostat, but no k_boundary_radi (which also opens the next plain step)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (3, 20, 1, 16, 37):
        a.step(n); b.step(n); c.step(n); o.step(n)
    a.set_profile(1); a.reset_kernel_times(); a.step(4); b.step(4); c.step(4); o.step(4)
    kt = a.kernel_times()
    assert a.stats()["sort_interval"] >= 8, a.stats()
    assert "integrate2_post" in kt and "boundary" in kt and "post_tstat" not in kt and "integrate2" not in kt, sorted(kt)
    assert kt["boundary"]["calls"] == 3 and kt["integrate2_post"]["calls"] == 1, kt       # the last step before the host looks is closed on its own
    sa, sb, sc, so = a.state(), b.state(), c.state(), o.state()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius"):
        assert np.array_equal(sa[k], sc[k]), (k, rel_err(sa[k], sc[k]))
        assert np.array_equal(sa[k], sb[k]), (k, rel_err(sa[k], sb[k]))
        ko = "rad" if k == "radius" else k
        assert rel_err(sa[k], so[ko]) < 1e-8, (k, rel_err(sa[k], so[ko]))
    for k in ("engKin", "engTemp", "engTot"):
        assert a.stats()[k] == b.stats()[k], (k, a.stats()[k], b.stats()[k])


@pytest.mark.parametrize("generic", [False, True])
def test_pair_lists_with_thermostat_radii(generic):
    """Pair lists where the potential depends on per-atom radii the radiative thermostat rewrites every step ('surk', case study 2 style, on cut-off sized
    cells so that a tile holds the stencil): the list kernel gathers the radii afresh with the coordinates.  Specialised surk mode and the generic
    switch-based body (debug bit 512) against the every-step schedule and the oracle."""
    pos, box = inputs.fcc_positions((7, 7, 7), 5.8, 0.1, 15)
    N = len(pos)
    case = {"box": box.tolist(), "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
            "vdw": [(0, 0, 7, 6.0, [75.0, 8.0, 1.0, 1.0])], "radii": [(2.73, 4.731, 0.2)], "x": pos[:, 0].copy(), "y": pos[:, 1].copy(),
            "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 300.0, "tstat_type": 2,
            "cell_list": 6.5, "use_clist": 1, "elec_type": 0}
    dbg = 512 if generic else 0
    a = engine(case, debug=dbg)
    b = engine(case, debug=dbg, sort_every=1)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 25, 6):
        a.step(n); b.step(n); o.step(n)
    sta = a.stats()
    assert sta["pair_lists"] == 1 and sta["sort_interval"] > 1 and sta["cells_without_list"] == 0, sta
    sa, sb, so = a.state(), b.state(), o.state()
    for k, ko in (("x", "x"), ("vx", "vx"), ("fx", "fx"), ("fz", "fz"), ("U", "U"), ("radius", "rad")):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], so[ko]) < 1e-9, (k, rel_err(sa[k], so[ko]))
    assert abs(sta["engVdW"] - o.stats()["engVdW"]) <= 1e-10 * abs(o.stats()["engVdW"])


def test_tile_and_atom_kernels_agree_bitwise_on_energy_scale():
    case = inputs.config("F3")
    a, b = engine(case, pair_variant=1), engine(case, pair_variant=2)
    with pytest.raises(api.AztotError):          # (variant 3 - four waves sharing a tile of cell bins - was retired in round 4: refused, not silently replaced)
        engine(case, pair_variant=3)
    sa, sb = a.state(), b.state()
    for k in FKEYS:
        assert rel_err(sa[k], sb[k]) < 1e-13


@pytest.mark.parametrize("case_name", ["dense", "aniso", "thin_z"])
def test_dense_cells_and_odd_shapes(case_name):
    """~85 atoms per cell (several 64-atom batches per cell, the 320-slot LDS tile flushed many times per cell), an anisotropic
    grid with charges, and the minimum of three cells along an axis."""
    if case_name == "dense":        # ~85 atoms per cell
        case = inputs.lj_case((11, 11, 20), a=1.5, jitter=0.03, seed=3, rc=4.0, cell_list=4.0)
        case["vdw"] = [(0, 0, 1, 4.0, [0.002, 0.9])]
    elif case_name == "aniso":
        case = inputs.lj_case((5, 6, 11), a=5.3, seed=4, rc=7.0, cell_list=7.0, charges=(0.3, -0.3), elec="fenn", r_real=7.0, alpha=0.4)
    else:
        case = inputs.lj_case((6, 6, 4), a=5.3, seed=5, rc=7.0, cell_list=7.0)
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    for variant in (2, 1):
        e = engine(case, pair_variant=variant)
        s, st = e.state(), e.stats()
        for k in FKEYS:
            assert rel_err(s[k], so[k]) < 1e-11, (variant, k, rel_err(s[k], so[k]))
        assert abs(st["engVdW"] - sto["engVdW"]) <= 1e-12 * abs(sto["engVdW"]) + 1e-14
        assert st["pairs_dropped"] == sto["nDropped"]


@pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref/ref_driver did not travel to this machine")
@pytest.mark.parametrize("name", ["lj_nose", "buck", "bmhs", "fennel_field", "direct", "ewald", "bonded"])
def test_hip_path_against_the_reference_binary(name):
    """No oracle in between: the HIP hot path against the reference's OWN compiled serial code (oracle/_ref/ref_driver: box.cpp,
    vdw.cpp, elec.cpp, cell_list.cpp, integrators.cpp, temperature.cpp, bonds.cpp, angles.cpp built where they lie) on the same
    inputs - forces at step 0 to 1e-11, trajectory after 30 steps to 1e-9 (the north star's tolerance), energies to 1e-10."""
    if name == "lj_nose":
        case = inputs.lj_case((5, 5, 5), a=5.26, seed=21, rc=6.5, cell_list=6.5, T=140.0, vel_T=100.0)
        case.update(tstat_type=1, tau=0.08, nEq=12, freqEq=4)
    elif name == "buck":
        case = mixed_case("buck", seed=5)
    elif name == "bmhs":
        case = mixed_case("bmhs", seed=6)
    elif name == "fennel_field":
        case = mixed_case("lnjs+fenn+field", seed=7)
        case.update(Uy=0.0, Uz=0.0)       # the serial clear_force knows only dU/dx (integrators.cpp:17-39); y, z are the GPU path's
    elif name == "direct":
        case = mixed_case("lnjs+dir", seed=8)
    elif name == "ewald":
        case = ewald_case(ewald_k=(5, 6, 7))
    else:
        case = inputs.molecular_case((9, 9, 9), seed=31, charges=(-0.2, 0.1), elec="fenn")
    case = dict(case)
    case.update(nsteps=30, dump=[0, 30])
    ref = oracle.run_ref(case)
    e = engine(case)
    d0, d30 = ref["dumps"][0], ref["dumps"][30]
    s = e.state()
    for k in FKEYS:
        assert rel_err(s[k], d0[k]) < 1e-11, (name, k)
    assert per_atom_err(s, d0, FRC) < 1e-9, (name, per_atom_err(s, d0, FRC))
    e.step(30)
    s, st = e.state(), e.stats()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(s[k], d30[k]) < 1e-9, (name, k, rel_err(s[k], d30[k]))
    assert per_atom_err(s, d30, FRC) < 1e-9 and per_atom_err(s, d30, VEL) < 1e-9, (name, per_atom_err(s, d30, FRC), per_atom_err(s, d30, VEL))
    for a, b in (("engVdW", "engVdW"), ("engCoul", "engElec3"), ("engKin", "engKin"), ("engTot", "engTot"), ("engBond", "engBond"),
                 ("engAngle", "engAngle"), ("engCoulRec", "engElec2"), ("engCoulConst", "engElec1")):
        assert abs(st[a] - d30[b]) <= 1e-10 * abs(d30[b]) + 1e-12, (name, a, st[a], d30[b])


@pytest.mark.parametrize("seed", range(24))
def test_randomised_configurations(seed):
    """24 seeded random systems (tests/util.py random_case): every supported kernel variant against the oracle - forces, energies, and 10 steps."""
    case = random_case(seed)
    o = oracle.Oracle(case)
    o.forces(0)
    so, sto = o.state(), o.stats()
    scale = max(np.abs(so[k]).max() for k in FKEYS) + 1e-300
    for variant in (0, 2, 1):
        e = engine(case, pair_variant=variant)
        s, st = e.state(), e.stats()
        for k in FKEYS:
            assert np.abs(s[k] - so[k]).max() <= 2e-11 * scale, (seed, variant, k, np.abs(s[k] - so[k]).max() / scale)
        for a, b in (("engVdW", "engVdW"), ("engCoul", "engElec3"), ("engCoulRec", "engElec2"), ("engCoulConst", "engElec1")):
            assert abs(st[a] - sto[b]) <= 2e-12 * abs(sto[b]) + 1e-12, (seed, variant, a, st[a], sto[b])
        assert st["pairs_dropped"] == sto["nDropped"]
    e.step(10)
    o.step(10)
    s, s2 = e.state(), o.state()
    for k in ("x", "vx", "fx"):
        assert rel_err(s[k], s2[k]) < 1e-8, (seed, k, rel_err(s[k], s2[k]))


@pytest.mark.parametrize("seed", range(16))
def test_randomised_dynamics(seed):
    """the random systems again, now with a thermostat (Nose-Hoover, radiative) and/or an equilibration schedule and with bonds and
    angles between near neighbours: 25 steps in two calls against the oracle (state 1e-8, energies 1e-9, thermostat scalars 1e-8)."""
    case = add_random_dynamics(random_case(100 + seed, vel=0.3), seed)
    o = oracle.Oracle(case)
    o.forces(2 if case.get("bonds") is not None else 0)
    e = engine(case)
    for n in (7, 18):
        e.step(n)
        o.step(n)
    s, so, st, sto = e.state(), o.state(), e.stats(), o.stats()
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(s[k], so[k]) < 1e-8, (seed, k, rel_err(s[k], so[k]))
    if case.get("tstat_type") == 2:
        assert rel_err(s["U"], so["U"]) < 1e-8 and rel_err(s["radius"], so["rad"]) < 1e-8
    for a, b in (("engVdW", "engVdW"), ("engCoul", "engElec3"), ("engKin", "engKin"), ("engTot", "engTot"), ("engBond", "engBond"),
                 ("engAngle", "engAngle"), ("engTemp", "engTemp"), ("engCoulRec", "engElec2")):
        assert abs(st[a] - sto[b]) <= 1e-9 * abs(sto[b]) + 1e-10, (seed, a, st[a], sto[b])
    assert abs(st["nose_chit"] - sto["chit"]) <= 1e-8 * abs(sto["chit"]) + 1e-14


def _raw_case(pos, box, eps=0.01006, sigma=3.3952):
    pos = np.asarray(pos, dtype=float)
    N = len(pos)
    return {"box": list(box), "dt": 0.001, "nsteps": 0, "species": [(39.9, 0.0)], "vdw": [(0, 0, 1, 8.5, [eps, sigma])],
            "types": np.zeros(N, dtype=np.int32), "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(),
            "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "elec_type": 0, "use_clist": 1, "cell_list": 8.5}


@pytest.mark.parametrize("name", ["one_atom", "pair_across_the_wall", "everything_in_one_cell", "three_cells_per_axis", "two_cells_per_axis"])
def test_degenerate_inputs(name):
    """a single atom, a pair that only interacts through the periodic image, 40 atoms in one cell of an otherwise empty 343-cell box,
    and boxes of 3 and 2 cells per axis (the latter falls back to the per-atom kernel: a neighbour cell would be reached through
    two images) - every kernel variant against the oracle, forces and 5 steps."""
    rng = np.random.default_rng(7)
    case = {"one_atom": lambda: _raw_case([[1, 2, 3]], [40, 40, 40]),
            "pair_across_the_wall": lambda: _raw_case([[0.5, 20, 20], [39.0, 20, 20]], [40, 40, 40]),
            "everything_in_one_cell": lambda: _raw_case(rng.uniform(0.5, 8.0, (40, 3)), [60, 60, 60], eps=1e-4, sigma=1.0),
            "three_cells_per_axis": lambda: _raw_case(rng.uniform(0, 27, (200, 3)), [27, 27, 27], eps=0.001, sigma=2.0),
            "two_cells_per_axis": lambda: _raw_case(rng.uniform(0, 18, (100, 3)), [18, 18, 18], eps=0.001, sigma=2.0)}[name]()
    o = oracle.Oracle(case)
    o.forces(0)
    so = o.state()
    o.step(5)
    s5 = o.state()
    scale = max(np.abs(so[k]).max() for k in FKEYS)
    if name == "pair_across_the_wall":
        assert scale > 1e-3                      # r = 1.5 A through the wall: strongly repulsive
    for variant in (1, 2, 0):
        e = engine(case, pair_variant=variant)
        s = e.state()
        for k in FKEYS:
            assert np.abs(s[k] - so[k]).max() <= 1e-12 * scale, (variant, k)
        e.step(5)
        s = e.state()
        for k in ("x", "vx", "fx"):
            assert np.abs(s[k] - s5[k]).max() <= 1e-9 * (np.abs(s5[k]).max() + 1e-300), (variant, k)


def test_c2_40k_energy_conservation_and_newton3():
    """BASELINE config 2 (40 000 Ar, rc 8.5): size-independent properties at full size + oracle forces."""
    case = inputs.config("C2")
    e = engine(case)
    s = e.state()
    for k in FKEYS:
        assert abs(s[k].sum()) < 1e-9            # Newton 3: sum of forces vanishes
    o = oracle.Oracle(case)
    o.forces(1)
    so = o.state()
    for k in FKEYS:
        assert rel_err(s[k], so[k]) < 1e-11
    e0 = e.stats()["engVdW"]
    e.step(200)
    st = e.stats()
    assert abs(st["engTot"] - e0) < 2e-4 * abs(e0)      # NVE drift over 200 fs from a cold jittered lattice (unshifted cut-off)
    assert st["pairs_dropped"] == 0


# ---------------------------------------------------------------------------------------------------------------------------------
# round 3: explicit Verlet skin, lists for any cell population, restart, pressure
# ---------------------------------------------------------------------------------------------------------------------------------
def test_box_that_is_a_whole_number_of_cutoffs_still_gets_a_skin():
    """L = 6 x rc exactly: the reference's split_cells (cuCellList.cu:9-34) makes cells of exactly rc, which used to leave the lazy re-sort no slack at all
    (interval 1, no lists).  With options.skin (default: automatic) the cells are sized rc + skin, the lists reach rc + skin and the interval opens up;
    skin < 0 gives the old behaviour back.  Both against the oracle; the sizes the engine reports are the ones it was asked for."""
    rc = 8.5
    case = inputs.lj_case((9, 9, 9), a=6 * rc / 9, seed=77, rc=rc, cell_list=rc, vel_T=85.0)
    a = engine(case)                               # automatic skin
    b = engine(case, skin=-1.0)                    # cells exactly as control.cell_list gives them
    c = engine(case, skin=0.6)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 30, 25):
        a.step(n); b.step(n); c.step(n); o.step(n)
    sta, stb, stc, sto = a.stats(), b.stats(), c.stats(), o.stats()
    assert stb["n_cells"] == 6 ** 3 and stb["skin"] == 0.0 and stb["sort_interval"] == 1 and stb["pair_lists"] == 0 and stb["rebuilds"] == 65, stb
    assert sta["n_cells"] == 5 ** 3 and 0.25 < sta["skin"] <= 0.5 and sta["sort_interval"] > 4 and sta["pair_lists"] == 1 and sta["sort_violations"] == 0, sta
    assert sta["rebuilds"] < 30 and sta["cells_without_list"] == 0, sta
    assert 0.6 <= stc["skin"] <= 0.76 and stc["sort_interval"] >= sta["sort_interval"] and stc["sort_violations"] == 0, stc
    so = o.state()
    for e in (a, b, c):
        s, st = e.state(), e.stats()
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
            assert rel_err(s[k], so[k]) < 1e-9, (k, rel_err(s[k], so[k]))
        assert abs(st["engTot"] - sto["engTot"]) <= 1e-10 * abs(sto["engTot"])


@pytest.mark.parametrize("kind", ["crowded_cells", "dense_wide_stencil", "sparse_cells", "four_waves_per_cell"])
def test_pair_lists_for_any_cell_population(kind):
    """Lanes of k_pair_list are (atom, slice) with 64 / atoms slices per atom - not a power of two: cells of 17 - 21 atoms run three slices each, 22 - 32 two,
    up to 64 one, a single atom eight.  'crowded_cells': cells of 1.45 rc hold 20 - 45 atoms (NS 1 - 3, lists of ~100 iterations);
    'dense_wide_stencil': a dense liquid on cells of rc / 2.2 (7 x 7 x 7 stencil, ~900 candidates per cell: the LDS tile is several times the default);
    'sparse_cells': a thin gas (0 - 3 atoms per cell, 8 slices per atom).  Every cell keeps its list; x / v / f against the every-step schedule and the oracle."""
    if kind == "crowded_cells":
        case = inputs.lj_case((12, 12, 12), a=5.6, seed=41, rc=6.0, cell_list=8.7, vel_T=150.0)
    elif kind == "dense_wide_stencil":
        case = inputs.lj_case((12, 12, 12), a=2.3, jitter=0.05, seed=42, rc=6.0, cell_list=2.7, vel_T=300.0)
        case["vdw"] = [(0, 0, 1, 6.0, [0.002, 1.9])]
    elif kind == "four_waves_per_cell":
        # options.waves_per_cell = 4 on an ordinary liquid: four waves share a cell's tile and split its 10 - 20 atoms (3 - 5 each, 12 - 21 slices per atom)
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=44, rc=7.5, cell_list=7.9, vel_T=120.0)
    else:
        case = inputs.lj_case((6, 6, 6), a=14.0, jitter=2.0, seed=43, rc=7.0, cell_list=7.0, vel_T=300.0)
    a = engine(case, pair_variant=2, split=4 if kind == "four_waves_per_cell" else 0)
    b = engine(case, pair_variant=2, sort_every=1)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 30, 7):
        a.step(n); b.step(n); o.step(n)
    sa, sb, so, sta, stb, sto = a.state(), b.state(), o.state(), a.stats(), b.stats(), o.stats()
    assert sta["sort_interval"] > 1 and sta["sort_violations"] == 0 and sta["pair_lists"] == 1 and sta["cells_without_list"] == 0, sta
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], so[k]) < 1e-9, (k, rel_err(sa[k], so[k]))
    for k in ("engVdW", "engKin", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-10 * max(abs(stb[k]), 1e-3), (k, sta[k], stb[k])
    assert abs(sta["engTot"] - sto["engTot"]) <= 1e-10 * abs(sto["engTot"])


def test_lists_grow_when_cells_do_not_fit():
    """The capacities of the lists come from the mean density; a system that is twice as dense in one half of the box overflows them there.  Such cells keep
    no list for one interval (the clean-up launch serves them: still exact), the engine grows the lists at its next look, and from then on every cell has one."""
    case = inputs.lj_case((12, 6, 6), a=5.6, seed=45, rc=6.5, cell_list=6.5, vel_T=60.0)
    x = case["x"]
    L = case["box"][0]
    # squeeze the right half of the atoms into the right third of the box: ~1.9 x the mean density there
    right = x > 0.5 * L
    case["x"] = np.where(right, L * (2.0 / 3.0) + (x - 0.5 * L) * (2.0 / 3.0), x * (4.0 / 3.0))
    case["x"] = np.round(np.clip(case["x"], 0.0, L - 1e-6), 6)
    case["vdw"] = [(0, 0, 1, 6.5, [0.0005, 2.2])]        # soft, small atoms: the squeeze must not blow the system up
    a = engine(case, pair_variant=2)
    b = engine(case, pair_variant=2, sort_every=1)
    unlisted = []
    for n in (8, 8, 8, 30, 30):
        a.step(n); b.step(n)
        unlisted.append(a.stats()["cells_without_list"])
    sa, sb, sta = a.state(), b.state(), a.stats()
    assert sta["pair_lists"] == 1 and unlisted[-1] == 0, (unlisted, sta)
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
        assert rel_err(sa[k], sb[k]) < 1e-9, (k, rel_err(sa[k], sb[k]))


@pytest.mark.parametrize("kind", ["radi", "nose", "nve_lazy"])
def test_restart_continues_exactly(kind):
    """Checkpoint row of SURVEY section 5: state() + clock() of a running engine put into a NEW engine (set_state incl. U and radius, set_clock: step number =
    equilibration schedule + key of the thermostat's counter-based random numbers, Nose-Hoover pair, last kinetic energy) continue the run exactly.
    With the reference's every-step schedule the continuation is bit-identical; with the lazy schedule the restarted engine rebuilds its cells at another
    step than the original did, so the sums differ in order (1e-11)."""
    if kind == "radi":
        case = inputs.lj_case((6, 6, 6), a=5.4, seed=51, rc=6.5, cell_list=6.5, T=300.0, tstat="radi", vel_T=200.0, nEq=30, freqEq=5)
        kw = dict(sort_every=1)
    elif kind == "nose":
        case = inputs.lj_case((6, 6, 6), a=5.4, seed=52, rc=6.5, cell_list=6.5, T=150.0, vel_T=100.0, nEq=30, freqEq=5)
        case.update(tstat_type=1, tau=0.05)
        kw = dict(sort_every=1)
    else:
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=53, rc=7.5, cell_list=7.5, vel_T=120.0)
        kw = {}
    a = engine(case, **kw)
    a.step(22)
    s, clk, st22 = a.state(), a.clock(), a.stats()
    assert clk["step"] == 22
    a.step(31)
    b = engine(case, **kw)
    b.set_state(**{k: s[k] for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")})
    b.set_clock(**clk)
    b.step(31)
    sa, sb, sta, stb = a.state(), b.state(), a.stats(), b.stats()
    assert sta["step"] == stb["step"] == 53
    for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius"):
        if kind == "nve_lazy":
            assert rel_err(sa[k], sb[k]) < 1e-11 or np.abs(sa[k]).max() == 0, (k, rel_err(sa[k], sb[k]))
        else:
            assert np.array_equal(sa[k], sb[k], equal_nan=True), (k, rel_err(sa[k], sb[k]))      # (radius is undefined - NaN on both sides - without a `radii` section)
    for k in ("engKin", "engVdW", "engTemp", "nose_chit", "nose_conint"):
        assert abs(sta[k] - stb[k]) <= 1e-11 * max(abs(sta[k]), 1e-6), (k, sta[k], stb[k])


def test_pressure_from_wall_momentum():
    """aztot_stats.pressure: the serial path's wall-momentum pressure over the last `stat` window, P = 2 x 1.58e6 x dMom / (dt x area) per face
    (main.cpp:143-163; GPU path main.cu:135-155), averaged over the six faces - asserted against the same formula applied to the ORACLE's wall momenta
    (box.cpp:230-295) over three consecutive windows of a hot gas in an anisotropic box."""
    case = inputs.lj_case((6, 5, 7), a=6.0, seed=61, rc=6.5, cell_list=6.5, vel_T=2500.0)
    case["stat"] = 20
    e = engine(case)
    o = oracle.Oracle(case)
    o.forces(1)
    Lx, Ly, Lz = case["box"]
    rev_area = [1.0 / (Ly * Lz)] * 2 + [1.0 / (Lx * Lz)] * 2 + [1.0 / (Lx * Ly)] * 2
    keys = ("momXn", "momXp", "momYn", "momYp", "momZn", "momZp")
    prev = [0.0] * 6
    seen = []
    for w in range(3):
        e.step(20); o.step(20)
        st, sto = e.stats(), o.stats()
        now = [sto[k] for k in keys]
        want = sum(2.0 * 1.58e6 * ra * (m1 - m0) / (20 * case["dt"]) for ra, m1, m0 in zip(rev_area, now, prev)) / 6.0
        prev = now
        assert want > 0.0
        assert abs(st["pressure"] - want) <= 1e-9 * abs(want), (w, st["pressure"], want)
        seen.append(st["pressure"])
    assert len(set(seen)) == 3          # a new window every time
    e.step(7)
    assert e.stats()["pressure"] == seen[-1]          # inside a window the last value stands (the reference prints it every `stat` steps only)


@pytest.mark.parametrize("kind", ["lj", "radi", "fennell"])
def test_window_run_again_without_the_cleanup_launch_is_exact(kind):
    """Small systems on one GPU run their plain steps WITHOUT the clean-up launch behind k_pair_list (Engine::choose_optimism); a look that finds a skin violation
    (or a cell that kept no list) goes back to the snapshot the last clean look left and runs the window again with the launch in place.  Debug bit 8192 holds the
    interval at 32 steps on atoms far too fast for it, so windows ARE run again - and the result must be what an engine that launches the clean-up kernel behind
    every step (debug bit 4) arrives at: the same trajectory (positions, velocities, forces, the radiative thermostat's per-atom energy and radius), the same
    energies, wall counters and per-species crossings, to summation order (the repaired steps are the same kernels in both engines); and both equal the
    every-step schedule to 1e-9.  'radi': the snapshot has to carry the thermostat's state and the step number its random numbers are keyed by."""
    if kind == "radi":
        case = inputs.lj_case((7, 7, 7), a=5.4, seed=61, rc=6.5, cell_list=6.9, T=3000.0, tstat="radi", vel_T=6000.0, radii=[(2.73, 4.731, 0.2)])
    elif kind == "fennell":
        case = inputs.lj_case((7, 7, 7), a=5.4, seed=62, rc=6.5, cell_list=6.9, charges=(0.2, -0.2), elec="fenn", r_real=6.5, vel_T=9000.0)
    else:
        case = inputs.lj_case((7, 7, 7), a=5.4, seed=23, rc=6.5, cell_list=6.9, vel_T=9000.0)
    case["dt"] = 0.002
    a = engine(case, sort_every=32, debug=8192)
    b = engine(case, sort_every=32, debug=8192 | 4)
    c = engine(case, sort_every=1)
    for n in (3, 40, 9, 70, 33):               # (an engine's first two looks are spent with the launch in place: the later calls are the ones that run without it)
        a.step(n); b.step(n); c.step(n)
    sa, sb, sc, sta, stb, stc = a.state(), b.state(), c.state(), a.stats(), b.stats(), c.stats()
    assert sta["sort_violations"] > 0 and stb["sort_violations"] > 0
    keys = ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz") + (("U", "radius") if kind == "radi" else ())
    for k in keys:
        assert rel_err(sa[k], sb[k]) < 1e-11, (k, rel_err(sa[k], sb[k]))
        assert rel_err(sa[k], sc[k]) < 1e-9, (k, rel_err(sa[k], sc[k]))
    assert sta["negCross"] == stb["negCross"] == stc["negCross"] and sta["posCross"] == stb["posCross"] == stc["posCross"]
    assert np.array_equal(a.species_crossings(), b.species_crossings()) and np.array_equal(a.species_crossings(), c.species_crossings())
    for k in ("engVdW", "engKin", "engTot", "engCoul", "engTemp"):
        if abs(stc[k]) > 0:
            assert abs(sta[k] - stb[k]) <= 1e-11 * abs(stb[k]), (k, sta[k], stb[k])
            assert abs(sta[k] - stc[k]) <= 1e-9 * abs(stc[k]), (k, sta[k], stc[k])
    assert sta["step"] == stb["step"] == stc["step"] == 155


@pytest.mark.parametrize("kind", ["gas", "liquid", "two_species", "equil"])
def test_radiative_thermostat_fused_into_the_pair_kernel(kind):
    """Runs with the radiative thermostat whose pair kernel reads no radii (case study 1): on plain steps of a lazy run without a clean-up launch the epilogue of
    k_pair_list closes the step as k_integrate2_post does (second half-kick, then the thermostat on the fully kicked velocity, random draws keyed by the number of
    the step being closed) and opens the next one - one launch per step instead of two.  Same operations in the same order: positions, velocities, forces and
    the thermostat's per-atom energy and radius must be BIT-IDENTICAL to a run with the fusion switched off (debug bit 131072), whatever the pattern of calls;
    energies equal to summation order, and both equal to the oracle.  'equil': an equilibration schedule - the steps it acts on (and their neighbours) take
    the unfused path."""
    if kind == "gas":        # the dilute gas of case study 1 in small: cells of 20 atoms' worth of empty space
        rng = np.random.Generator(np.random.PCG64(77))
        N, L = 3000, 480.0
        g = 15
        site = rng.permutation(g ** 3)[:N]
        pos = np.stack([site // (g * g), (site // g) % g, site % g], axis=1) * (L / g) + L / (2 * g) + rng.uniform(-10.0, 10.0, size=(N, 3))
        pos = np.round(np.mod(pos, L), 6)
        case = inputs.lj_case((2, 2, 2), a=5.4, seed=1, rc=4.0, cell_list=80.0, T=298.0, tstat="radi", radii=[(2.73, 4.731, 0.2)])
        case.update(box=[L, L, L], types=np.zeros(N, dtype=np.int32), x=pos[:, 0].copy(), y=pos[:, 1].copy(), z=pos[:, 2].copy(), vx=np.zeros(N), vy=np.zeros(N), vz=np.zeros(N))
    elif kind == "two_species":
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=43, rc=7.5, cell_list=7.9, T=250.0, tstat="radi", vel_T=200.0, charges=(0.3, -0.3), elec="fenn", r_real=7.5, alpha=0.3,
                              radii=[(2.73, 4.731, 0.2), (2.73, 4.731, 0.2)])
    else:
        case = inputs.lj_case((8, 8, 8), a=5.6, seed=44, rc=7.5, cell_list=7.9, T=250.0, tstat="radi", vel_T=200.0, radii=[(2.73, 4.731, 0.2)],
                              nEq=60 if kind == "equil" else 0, freqEq=7)
    a = engine(case, pair_variant=2)
    b = engine(case, pair_variant=2, debug=131072)
    o = oracle.Oracle(case)
    o.forces(1)
    for n in (10, 1, 37, 2, 33, 64, 5):
        a.step(n); b.step(n); o.step(n)
        sa, sb = a.state(), b.state()
        for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius"):
            assert np.array_equal(sa[k], sb[k]), (n, k, rel_err(sa[k], sb[k]))
    sta, stb, so, sto = a.stats(), b.stats(), o.state(), o.stats()
    assert sta["pair_lists"] == 1 and sta["sort_interval"] > 1
    assert sta["negCross"] + sta["posCross"] == stb["negCross"] + stb["posCross"]
    for k in ("engVdW", "engCoul", "engKin", "engTemp", "engTot"):
        assert abs(sta[k] - stb[k]) <= 1e-12 * max(abs(stb[k]), 1e-3), (k, sta[k], stb[k])
    for k in ("x", "y", "z", "vx", "vy", "vz", "U"):
        assert rel_err(sa[k], so[k]) < 1e-8, (k, rel_err(sa[k], so[k]))
    assert abs(sta["engTemp"] - sto["engTemp"]) <= 1e-9 * max(abs(sto["engTemp"]), 1e-3)
