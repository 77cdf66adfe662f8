"""Shared helpers for the parity tests."""
import numpy as np

from aztotmd_amd import inputs


def rel_err(a, b):
    """max |a-b| / max |b| (array-level relative error: the north star's 'forces within 1e-9 relative')."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


def per_atom_err(a, b, keys=("fx", "fy", "fz"), floor=1e-3):
    """The north star's "forces within 1e-9 relative" read atom by atom: max_i |dF_i| / max(|F_i|, floor * F_rms) over the vectors (a[k], b[k]).
    rel_err scales by the LARGEST component of the whole array, which says nothing about an atom whose force is a hundred times smaller; here every
    atom is measured against its own force (atoms whose force nearly vanishes - a lattice site - against a thousandth of the rms force)."""
    d = np.sqrt(sum((np.asarray(a[k], dtype=float) - np.asarray(b[k], dtype=float)) ** 2 for k in keys))
    mag = np.sqrt(sum(np.asarray(b[k], dtype=float) ** 2 for k in keys))
    rms = float(np.sqrt((mag ** 2).mean()))
    if rms == 0.0:
        return float(d.max())
    return float((d / np.maximum(mag, floor * rms)).max())


VEL = ("vx", "vy", "vz")
FRC = ("fx", "fy", "fz")


def mixed_case(pot, n=5, a=5.6, seed=99, rc=7.0, vel_T=90.0):
    """Two-species liquid with the given potential family (same generator as the live-reference CPU test)."""
    base = inputs.lj_case((n, n, n), a=a, seed=seed, rc=rc, vel_T=vel_T, cell_list=rc)
    N = len(base["types"])
    sp = [(39.9, 0.0), (20.2, 0.0)]
    names = ["A", "B"]
    base["types"] = (np.arange(N) % 2).astype(np.int32)
    pairs = ((0, 0), (0, 1), (1, 1))
    if pot == "buck":
        vdw = [(0, 0, 2, rc, [1822.0, 0.3, 63.0]), (0, 1, 2, rc - 0.5, [1400.0, 0.29, 40.0]), (1, 1, 2, rc - 1.0, [900.0, 0.28, 20.0])]
    elif pot == "bmhs":
        vdw = [(a_, b_, 4, rc, [0.25, 3.1, 2.4 + 0.1 * a_, 60.0, 80.0]) for a_, b_ in pairs]
    elif pot == "p746":
        vdw = [(a_, b_, 3, rc, [3000.0, 1.0, 20.0 + 5 * b_]) for a_, b_ in pairs]
    elif pot == "elin":
        vdw = [(a_, b_, 5, rc, [900.0, 0.4, 0.002 + 0.001 * b_]) for a_, b_ in pairs]
    elif pot == "einv":
        vdw = [(a_, b_, 6, rc, [900.0, 0.4, 0.5 + 0.1 * b_]) for a_, b_ in pairs]
    else:
        vdw = [(a_, b_, 1, rc, [0.01006, 3.3952]) for a_, b_ in pairs]
        sp = [(39.9, 0.3), (20.2, -0.3)]
        if pot == "lnjs+dir":
            base.update(elec_type=1, rReal=rc)
        elif pot == "lnjs+fenn+field":
            base.update(elec_type=3, rReal=rc, alpha=0.35, Ux=0.02, Uy=-0.01, Uz=0.005)
    base.update(species=sp, vdw=vdw, names=names)
    return base


def family_with_coulomb(family, elec, **kw):
    """One potential family (buck / bmhs / p746 / lnjs) for every species pair + charges with the given electrostatics
    ('dir', 'fenn', 'ewald'): the combinations the specialised tile kernels (MODE 2 / 3 x family) cover."""
    base = mixed_case(family if family != "lnjs" else "lnjs+dir", **kw)
    base["species"] = [(39.9, 0.3), (20.2, -0.3)]
    rc = 7.0
    base.update(elec_type={"dir": 1, "ewald": 2, "fenn": 3}[elec], rReal=rc, alpha=0.0 if elec == "dir" else 0.4)
    if elec == "ewald":
        base["ewald_k"] = (5, 5, 5)
    return base


def random_case(seed, x_cells=0, vel=1.0):
    """A random small system: box shape, density, cut-off, cell edge (below and above the cut-off), 1-3 species, potential family mix,
    electrostatics, external field - whatever the input surface allows for the pair path."""
    rng = np.random.default_rng(1000 + seed)
    box = rng.uniform(16.0, 34.0, size=3)
    rc = float(rng.uniform(4.0, min(7.5, 0.49 * box.min())))
    N = int(rng.integers(150, 700))
    if x_cells:                                                  # slab tests: a box long enough along x for that many cut-off lengths
        box[0] = rc * x_cells * float(rng.uniform(1.02, 1.3))
        N = int(N * box[0] / 25.0)
    ns = int(rng.integers(1, 4))
    # points with a minimum separation (rejection on a jittered grid) so that no pair sits deep inside the repulsive wall
    g = np.ceil(N ** (1 / 3)).astype(int) + 1
    sites = rng.permutation(g ** 3)[:N]
    pos = (np.stack([sites // (g * g), (sites // g) % g, sites % g], axis=1) + 0.5 + rng.uniform(-0.2, 0.2, size=(N, 3))) * (box / g)
    pos = np.mod(pos, box)
    types = rng.integers(0, ns, size=N).astype(np.int32)
    fam = ["lnjs", "buck", "bmhs", "p746", "mixed", "elin"][int(rng.integers(0, 6))]
    d0 = float((box.prod() / N) ** (1 / 3))                     # typical neighbour distance: keeps the potentials in a sane range
    vdw = []
    for a_ in range(ns):
        for b_ in range(a_, ns):
            rcp = rc * float(rng.uniform(0.7, 1.0))
            kind = fam if fam != "mixed" else ["lnjs", "buck", "bmhs"][int(rng.integers(0, 3))]
            if kind == "lnjs":
                vdw.append((a_, b_, 1, rcp, [float(rng.uniform(0.005, 0.02)), 0.8 * d0]))
            elif kind == "buck":
                vdw.append((a_, b_, 2, rcp, [float(rng.uniform(500, 2000)), 0.09 * d0, float(rng.uniform(5, 40))]))
            elif kind == "bmhs":
                vdw.append((a_, b_, 4, rcp, [0.25, float(rng.uniform(2.5, 3.5)), 0.8 * d0, float(rng.uniform(5, 40)), float(rng.uniform(5, 40))]))
            elif kind == "p746":
                vdw.append((a_, b_, 3, rcp, [float(rng.uniform(500, 3000)) * (d0 / 3.0) ** 7, 1.0, float(rng.uniform(5, 30))]))
            else:
                vdw.append((a_, b_, 5, rcp, [float(rng.uniform(300, 900)), 0.12 * d0, float(rng.uniform(0.001, 0.004))]))
            if rng.random() < 0.15 and len(vdw) > 1:
                vdw.pop()                                        # some species pairs have no potential at all
    elec = ["none", "dir", "fenn", "ewald"][int(rng.integers(0, 4))]
    charges = rng.uniform(-0.4, 0.4, size=ns) if elec != "none" else np.zeros(ns)
    if elec != "none" and ns > 1 and rng.random() < 0.3:
        charges[0] = 0.0                                        # a neutral species among charged ones
    case = {"box": [float(v) for v in box], "dt": 0.0005, "nsteps": 0, "species": [(float(rng.uniform(10, 60)), float(q)) for q in charges],
            "vdw": vdw, "types": types, "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(),
            "vx": rng.normal(0, vel, N), "vy": rng.normal(0, vel, N), "vz": rng.normal(0, vel, N),
            "elec_type": {"none": 0, "dir": 1, "ewald": 2, "fenn": 3}[elec], "rReal": rc if elec != "none" else 0.0,
            "alpha": float(rng.uniform(0.3, 0.55)) if elec in ("fenn", "ewald") else 0.0, "use_clist": 1,
            "cell_list": rc * float(rng.choice([0.45, 0.7, 1.0, 1.3]) if not x_cells else rng.choice([1.0, 1.15])), "Ux": float(rng.choice([0.0, 0.01])), "Uy": 0.0, "Uz": float(rng.choice([0.0, -0.02]))}
    if elec == "ewald":
        case["ewald_k"] = tuple(int(v) for v in rng.integers(3, 8, size=3))
    if elec != "none" and not np.any(charges != 0.0):
        case["elec_type"] = 0
    return case


def add_random_dynamics(case, seed):
    """Decorate a random_case with a thermostat / equilibration schedule and with bonds + angles between near neighbours."""
    rng = np.random.default_rng(5000 + seed)
    case = dict(case)
    mode = ["none", "nose", "radi", "equil"][int(rng.integers(0, 4))]
    case["T"] = float(rng.uniform(80.0, 400.0))
    if mode == "nose":
        case.update(tstat_type=1, tau=float(rng.uniform(0.02, 0.2)))
    elif mode == "radi":
        case.update(tstat_type=2, radii=[(2.73, 4.731, 0.2)] * len(case["species"]))
    if mode in ("equil", "nose") or rng.random() < 0.3:
        case.update(nEq=int(rng.integers(4, 15)), freqEq=int(rng.integers(1, 5)))
    if rng.random() < 0.6:
        box = np.array(case["box"])
        pos = np.stack([case["x"], case["y"], case["z"]], axis=1)
        types = np.asarray(case["types"])
        N = len(types)
        d = pos[:, None, :] - pos[None, :, :]
        d -= box * np.round(d / box)
        r = np.sqrt((d ** 2).sum(-1)) + np.eye(N) * 1e9
        nn = np.argsort(r, axis=1)[:, :2]
        rmax = 0.45 * min(case.get("rReal") or 1e9, max(v[3] for v in case["vdw"]) if case["vdw"] else 1e9, 0.49 * box.min())
        btype, bond_types, bonds, angle_types, atype, angles, seen = {}, [], [], [], {}, [], set()
        for i in rng.permutation(N)[: N // 3]:
            js = [int(j) for j in nn[i] if r[i, j] < rmax]
            for j in js:
                key = (min(i, j), max(i, j))
                if key in seen:
                    continue
                seen.add(key)
                sp = (int(types[i]), int(types[j]))
                if sp not in btype and sp[::-1] not in btype:
                    kind = int(rng.integers(1, 3))
                    r0 = float(r[i, j])
                    bond_types.append((sp[0], sp[1], kind, [8.0, r0] if kind == 1 else [1.5, 1.2, r0, 0.3]))
                    btype[sp] = len(bond_types)
                t = btype.get(sp) or btype[sp[::-1]]
                bonds.append((int(i), j, t))
            if len(js) == 2 and r[js[0], js[1]] < 2 * rmax:
                c = int(types[i])
                if c not in atype:
                    angle_types.append((c, 1, [float(rng.uniform(0.5, 2.0)), float(rng.uniform(-0.6, 0.2))]))
                    atype[c] = len(angle_types)
                angles.append((int(i), js[0], js[1], atype[c]))
        if bonds:
            case.update(bond_types=bond_types, bonds=np.array(bonds, dtype=np.int32))
            if angles:
                case.update(angle_types=angle_types, angles=np.array(angles, dtype=np.int32))
    return case


def materialise_case_study(k, directory, nstep=None):
    """Write the reference's shipped example input k (1 or 2) - stored as data in tests/golden/case_study_k.npz by
    tests/golden/make_case_studies.py - back into `directory` as atoms.xyz / field.txt / control.txt / cuda.txt."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "case_study_%d.npz" % k))
    os.makedirs(directory, exist_ok=True)
    names, idx, pos = [str(s) for s in z["names"]], z["name_idx"], z["pos"]
    eol = "\r\n" if bool(z["crlf"]) else "\n"           # the shipped files have DOS line ends: kept, the parser must cope
    with open(os.path.join(directory, "atoms.xyz"), "w", newline="") as f:
        f.write("%d%s%s%s" % (int(z["n"]), eol, bytes(z["box_line"]).decode(), eol))
        f.writelines("%s\t%f\t%f\t%f%s" % (names[idx[i]], pos[i, 0], pos[i, 1], pos[i, 2], eol) for i in range(len(pos)))
    for fn in ("field.txt", "control.txt", "cuda.txt"):
        data = bytes(z[fn.replace(".", "_")])
        if fn == "control.txt" and nstep is not None:
            import re
            data = re.sub(rb"nstep\s+\d+", b"nstep %d" % nstep, data, count=1)
        with open(os.path.join(directory, fn), "wb") as f:
            f.write(data)
    return directory


def case_from_parsed(o, seed=12345):
    """oracle/parse.py's view of an input directory as the `case` dict the oracle (and aztot_model_create) take."""
    N = o["n_atoms"]
    sp = o["species"]
    return {"box": list(o["box"]), "dt": o["dt"], "nsteps": 0, "species": [(s["mass_amu"], s["charge"]) for s in sp],
            "names": [s["name"] for s in sp], "vdw": [tuple(v) for v in o["vdw_raw"]], "types": np.asarray(o["types"], dtype=np.int32),
            "x": np.asarray(o["x"], dtype=float), "y": np.asarray(o["y"], dtype=float), "z": np.asarray(o["z"], dtype=float),
            "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "elec_type": o["elec_type"], "rReal": o["r_real"] if o["elec_type"] else 0.0,
            "alpha": o["alpha"] if o["elec_type"] else 0.0, "T": o["temperature"], "tstat_type": o["tstat_type"], "tau": o["tau"],
            "nEq": o["nequil"], "freqEq": o["eqfreq"] or 1, "use_clist": o["use_cell_list"], "cell_list": o["cell_list"],
            "radii": [(s["radA"], s["radB"], s["mxEng"]) for s in sp], "frozen": [s["frozen"] for s in sp], "seed": seed,
            "Ux": o["elecfield"][0], "Uy": o["elecfield"][1], "Uz": o["elecfield"][2], "center_box": 0, "init_forces": 1}
