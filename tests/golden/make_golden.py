"""Generate the committed golden fixtures from the reference's own compiled serial path.

Run HERE (the container with /root/reference):   python tests/golden/make_golden.py
It executes oracle/_ref/ref_driver (the reference's box.cpp / vdw.cpp / elec.cpp / cell_list.cpp /
integrators.cpp / temperature.cpp / bonds.cpp / angles.cpp compiled where they lie + our harness) on small seeded cases and stores inputs and
outputs as data under tests/golden/.  No reference source text is stored.
"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from aztotmd_amd import inputs          # noqa: E402
from oracle import oracle               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
EKEYS = ("engVdW", "engElec3", "engKin", "engTot", "engElecField", "Temp", "momXn", "momXp", "momYn", "momYp", "momZn", "momZp")


def two_atom_case(r, vdw, species=((39.9, 0.0),), types=(0, 0), elec=0, rReal=0.0, alpha=0.0):
    return {"box": [60.0, 60.0, 60.0], "dt": 0.001, "nsteps": 0, "species": list(species), "vdw": vdw,
            "types": np.array(types, dtype=np.int32), "x": np.array([10.0, 10.0 + r]), "y": np.array([30.0, 30.0]),
            "z": np.array([30.0, 30.0]), "vx": np.zeros(2), "vy": np.zeros(2), "vz": np.zeros(2),
            "elec_type": elec, "rReal": rReal, "alpha": alpha, "use_clist": 0, "init_forces": 1, "dump": [0]}


def pair_tables():
    rs = [2.0, 2.5, 3.0, 3.5, 3.810998, 4.0, 5.0, 6.0, 7.25, 8.0, 8.5, 8.500001]
    pots = {
        "lnjs": (1, 8.5, [0.01006, 3.3952]),
        "buck": (2, 8.5, [1822.0, 0.3, 63.0]),
        "p746": (3, 8.5, [120.0, 4.0, 30.0]),
        "bmhs": (4, 8.5, [0.25, 3.1, 2.4, 60.0, 80.0]),
    }
    out = {}
    for name, (t, rc, p) in pots.items():
        rows = []
        for r in rs:
            d = oracle.run_ref(two_atom_case(r, [(0, 0, t, rc, p)]))["dumps"][0]
            rows.append({"r": r, "engVdW": d["engVdW"], "fx0": float(d["fx"][0]), "fx1": float(d["fx"][1])})
        out[name] = {"type": t, "rc": rc, "params": p, "rows": rows}
    lj = [(0, 0, 1, 8.5, [0.01006, 3.3952]), (0, 1, 1, 8.5, [0.01006, 3.3952]), (1, 1, 1, 8.5, [0.01006, 3.3952])]
    sp = ((39.9, 0.2), (39.9, -0.2))
    for name, et, rr, al in (("fenn", 3, 8.5, 0.4), ("dir", 1, 8.5, 0.0)):
        rows = []
        for r in rs:
            d = oracle.run_ref(two_atom_case(r, lj, species=sp, types=(0, 1), elec=et, rReal=rr, alpha=al))["dumps"][0]
            rows.append({"r": r, "engVdW": d["engVdW"], "engElec3": d["engElec3"], "fx0": float(d["fx"][0])})
        out[name] = {"elec_type": et, "rReal": rr, "alpha": al, "charges": [0.2, -0.2], "rows": rows}
    with open(os.path.join(HERE, "pairs_kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def save_run(name, case, dump, keep_full):
    case = dict(case)
    case["nsteps"] = max(dump)
    case["dump"] = dump
    ref = oracle.run_ref(case)
    data = {"box": np.array(case["box"]), "dt": case["dt"], "types": np.asarray(case["types"], dtype=np.int32),
            "species": np.array(case["species"]), "nHead": ref["nHead"], "cells": np.array(ref["cells"]),
            "steps": np.array(dump)}
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        data["in_" + k] = np.asarray(case[k])
    for st in dump:
        d = ref["dumps"][st]
        data["e_%d" % st] = np.array([d[k] for k in EKEYS])
        for k in (("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz") if st in keep_full else ()):
            data["%s_%d" % (k, st)] = d[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "N=%d" % len(case["types"]), "cells", ref["cells"], "E0 %.12e" % ref["dumps"][dump[0]]["engVdW"])


def survey_f2_case():
    """The 4 000-atom probe system of SURVEY.md Appendix F (python random.seed(12345))."""
    random.seed(12345)
    a = 5.26
    basis = [(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)]
    pos = []
    for i in range(10):
        for j in range(10):
            for k in range(10):
                for b in basis:
                    pos.append([float("%f" % ((c + bb) * a + 0.25 + random.uniform(-.15, .15))) for c, bb in zip((i, j, k), b)])
    pos = np.array(pos)
    N = len(pos)
    return {"box": [52.6, 52.6, 52.6], "dt": 0.001, "species": [(39.9, 0.0)], "vdw": [(0, 0, 1, 8.5, [0.01006, 3.3952])],
            "types": np.zeros(N, dtype=np.int32), "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(),
            "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "elec_type": 0, "use_clist": 1, "center_box": 1,
            "init_forces": 1, "T": 85.0}


def bond_tables():
    """Known answers of the five bond potentials + the harmonic-cosine angle from the reference binary
    (bond_iter bonds.cpp:731-787, angle_iter angles.cpp:179-227): one dimer / one trimer, no pair potential."""
    pots = {"harm": (1, [30.0, 1.0]), "mors": (2, [4.0, 2.0, 1.0, 0.5]), "pdn": (3, [4.0, 2.0, 1.0, 0.5, 0.002]),
            "buck": (4, [2.0e4, 0.1, 1.513]), "e612": (5, [2.0e4, 0.1, 1.1467, 0.2, 0.05])}
    out = {}
    for name, (t, p) in pots.items():
        rows = []
        for r in (0.7, 0.85, 0.95, 1.0, 1.05, 1.2, 1.6, 2.5):
            c = two_atom_case(r, [], species=((12.0, 0.0), (1.0, 0.0)), types=(0, 1))
            c.update(nsteps=1, dump=[1], dt=1e-9, bond_types=[(0, 1, t, p)], bonds=np.array([[1, 0, 1]]))   # listed ligand-first: turned
            d = oracle.run_ref(c)["dumps"][1]
            rows.append({"r": r, "engBond": d["engBond"], "fx0": float(d["fx"][0]), "fx1": float(d["fx"][1])})
        out[name] = {"type": t, "params": p, "rows": rows}
    rows = []
    for th in (60.0, 90.0, 104.5, 120.0, 150.0, 175.0):
        t = np.deg2rad(th)
        c = {"box": [60.0, 60.0, 60.0], "dt": 1e-9, "nsteps": 1, "species": [(12.0, 0.0), (1.0, 0.0)], "vdw": [],
             "types": np.array([0, 1, 1], dtype=np.int32), "x": np.array([59.9, 0.95, 59.9 + 1.1 * np.cos(t)]),
             "y": np.array([30.0, 30.0, 30.0 + 1.1 * np.sin(t)]), "z": np.array([30.0, 30.0, 30.0]),
             "vx": np.zeros(3), "vy": np.zeros(3), "vz": np.zeros(3), "use_clist": 0, "init_forces": 1, "dump": [1],
             "angle_types": [(0, 1, [3.0, -0.33])], "angles": np.array([[0, 1, 2, 1]])}
        d = oracle.run_ref(c)["dumps"][1]
        rows.append({"theta": th, "x": c["x"].tolist(), "y": c["y"].tolist(), "engAngle": d["engAngle"],
                     "fx": d["fx"].tolist(), "fy": d["fy"].tolist(), "fz": d["fz"].tolist()})
    out["hcos"] = {"params": [3.0, -0.33], "rows": rows}
    with open(os.path.join(HERE, "bonds_kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def save_bonded(name, case, dump, keep_full):
    """Trajectory fixture of a molecular system; the case is regenerated by inputs.molecular_case in the tests,
    the fixture keeps the inputs too so that a drift of the generator is caught."""
    case = dict(case)
    case["nsteps"] = max(dump)
    case["dump"] = dump
    ref = oracle.run_ref(case)
    keys = EKEYS + ("engBond", "engAngle")
    data = {"box": np.array(case["box"]), "dt": case["dt"], "types": np.asarray(case["types"], dtype=np.int32),
            "species": np.array(case["species"]), "nHead": ref["nHead"], "cells": np.array(ref["cells"]), "steps": np.array(dump),
            "bonds": np.asarray(case["bonds"], dtype=np.int32), "angles": np.asarray(case["angles"], dtype=np.int32)}
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        data["in_" + k] = np.asarray(case[k])
    for st in dump:
        d = ref["dumps"][st]
        data["e_%d" % st] = np.array([d[k] for k in keys])
        for k in (("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz") if st in keep_full else ()):
            data["%s_%d" % (k, st)] = d[k]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **data)
    print(name, "N=%d" % len(case["types"]), "cells", ref["cells"], "nHead", ref["nHead"], "Ebond %.12e" % ref["dumps"][dump[-1]]["engBond"])


def bonded_fixtures():
    bond_tables()
    save_bonded("M1_bonded", inputs.molecular_case((10, 10, 10)), [0, 1, 10, 40], keep_full=[1, 40])
    save_bonded("M1_bonded_fenn", inputs.molecular_case((10, 10, 10), charges=(-0.2, 0.1), elec="fenn"), [0, 1, 40], keep_full=[40])


def ewald_case():
    """500 ions (+-0.4 e, LJ) with the full Ewald sum: 'elec pme 6.5 0.45 6 6 6'."""
    c = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, charges=(0.4, -0.4), elec="fenn", r_real=6.5, alpha=0.45, vel_T=80.0)
    c.update(elec_type=2, ewald_k=(6, 6, 6))
    return c


def ewald_fixtures():
    """Trajectory with the reference's ewald_rec / ewald_const / coul_iter (elec.cpp) in the loop."""
    case = ewald_case()
    dump = [0, 1, 10, 40]
    case.update(nsteps=40, dump=dump)
    ref = oracle.run_ref(case)
    keys = EKEYS + ("engBond", "engAngle", "engElec1", "engElec2")
    data = {"box": np.array(case["box"]), "dt": case["dt"], "types": np.asarray(case["types"], dtype=np.int32),
            "species": np.array(case["species"]), "nHead": ref["nHead"], "cells": np.array(ref["cells"]), "steps": np.array(dump)}
    for k in ("x", "y", "z", "vx", "vy", "vz"):
        data["in_" + k] = np.asarray(case[k])
    for st in dump:
        d = ref["dumps"][st]
        data["e_%d" % st] = np.array([d[k] for k in keys])
        for k in (("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz") if st in (0, 40) else ()):
            data["%s_%d" % (k, st)] = d[k]
    np.savez_compressed(os.path.join(HERE, "E1_ewald.npz"), **data)
    d = ref["dumps"][40]
    print("E1_ewald", "N=%d" % len(case["types"]), "Econst %.9f Erec %.9f Ereal %.9f" % (d["engElec1"], d["engElec2"], d["engElec3"]))


if __name__ == "__main__":
    oracle.build()
    if sys.argv[1:] == ["ewald"]:
        ewald_fixtures()
        sys.exit(0)
    if sys.argv[1:] == ["bonded"]:      # only the fixtures of the bonds + angles row (leaves the others untouched)
        bonded_fixtures()
        sys.exit(0)
    pair_tables()
    save_run("F1_lj", inputs.config("F1"), [0, 1, 10, 50], keep_full=[0, 1, 10, 50])
    save_run("F2_lj", inputs.config("F2"), [0, 1, 10, 50], keep_full=[0, 50])
    save_run("F3_fennel", inputs.config("F3"), [0, 1, 10, 50], keep_full=[0, 50])
    c = inputs.config("F1"); c["nEq"] = 20; c["freqEq"] = 5; c["T"] = 85.0      # equilibration T-scaling
    vel = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, vel_T=60.0)
    for k in ("vx", "vy", "vz"):
        c[k] = vel[k]
    save_run("F1_tscale", c, [0, 5, 20, 30], keep_full=[5, 30])
    save_run("F2_survey", survey_f2_case(), [0, 50], keep_full=[])
    # Nose-Hoover (tstat_nose, temperature.cpp:339-360) with equilibration rescaling on top
    c = inputs.lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5, T=120.0, vel_T=80.0)
    c.update(tstat_type=1, tau=0.05, nEq=10, freqEq=5)
    save_run("F1_nose", c, [0, 1, 10, 40], keep_full=[1, 40])
    bonded_fixtures()
    ewald_fixtures()
