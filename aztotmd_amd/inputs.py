"""Synthetic input generator for the benchmark / parity configurations of SURVEY.md section 8(d).

Produces the reference's own input surface: `atoms.xyz`, `field.txt`, `control.txt`, `cuda.txt`
(grammar: sys_init.cpp:174-989, cuInit.cu:684-754 of the reference) and, for array-based entry
points, a plain dict ("case") with the same information.

Species / potentials / electrostatics are given in the reference's *input* units
(Angstrom, ps, eV, e, amu).
"""
import os

import numpy as np

VDW_TYPES = {"lnjs": 1, "buck": 2, "p746": 3, "bmhs": 4, "elin": 5, "einv": 6, "surk": 7}
VDW_NAMES = {v: k for k, v in VDW_TYPES.items()}
VDW_NPARAM = {1: 2, 2: 3, 3: 3, 4: 5, 5: 3, 6: 3, 7: 4}          # vdw.cpp:195
ELEC_TYPES = {"none": 0, "dir": 1, "pme": 2, "fenn": 3}
ELEC_NAMES = {v: k for k, v in ELEC_TYPES.items()}
TSTAT_NAMES = {0: "none", 1: "nose", 2: "radi"}

AR_MASS = 39.9
AR_EPS = 0.01006
AR_SIGMA = 3.3952


def fcc_positions(ncell, a, jitter, seed, offset=0.25):
    """FCC lattice of ncell=(nx,ny,nz) conventional cells, edge a, + U(-jitter, jitter) per coordinate."""
    nx, ny, nz = ncell
    basis = np.array([(0, 0, 0), (.5, .5, 0), (.5, 0, .5), (0, .5, .5)], dtype=np.float64)
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    cells = np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1).astype(np.float64)
    pos = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a + offset
    rng = np.random.Generator(np.random.PCG64(seed))
    pos += rng.uniform(-jitter, jitter, size=pos.shape)
    box = np.array([nx * a, ny * a, nz * a], dtype=np.float64)
    pos = np.mod(pos, box)
    pos[pos >= box] = 0.0
    return pos, box


def lj_case(ncell, a=5.735, jitter=0.15, seed=20240501, rc=8.5, dt=0.001, charges=None, elec="none",
            r_real=8.5, alpha=0.4, T=85.0, tstat="none", nsteps=0, cell_list=None, vel_T=None, quantize=True,
            radii=None, nEq=0, freqEq=1):
    """LJ-argon box (configs C2/C3/C4 and the small parity fixtures F1-F3 of SURVEY 8d).

    charges: None (one neutral species) or (qA, qB) -> two species alternating by atom index.
    vel_T: if given, Maxwell-Boltzmann velocities at that temperature (zero net momentum);
           default: zero velocities (`init_vel zero`).
    quantize: round coordinates to 6 decimals so that what is written to atoms.xyz with %f is
              exactly what the array entry points see.
    """
    pos, box = fcc_positions(ncell, a, jitter, seed)
    if quantize:
        pos = np.round(pos, 6)
        box = np.round(box, 6)
        pos[pos >= box] = 0.0
    N = pos.shape[0]
    if charges is None:
        species = [(AR_MASS, 0.0)]
        names = ["Ar"]
        types = np.zeros(N, dtype=np.int32)
        vdw = [(0, 0, VDW_TYPES["lnjs"], rc, [AR_EPS, AR_SIGMA])]
    else:
        species = [(AR_MASS, charges[0]), (AR_MASS, charges[1])]
        names = ["A", "B"]
        types = (np.arange(N) % 2).astype(np.int32)
        vdw = [(0, 0, 1, rc, [AR_EPS, AR_SIGMA]), (0, 1, 1, rc, [AR_EPS, AR_SIGMA]), (1, 1, 1, rc, [AR_EPS, AR_SIGMA])]
    vel = np.zeros_like(pos)
    if vel_T:
        kB = 1.3806488E-23 / 1.60217733E-19
        m = AR_MASS * (1.6605402E-27 / (1.60217733E-19 * 1e-24 / 1e-20))
        rng = np.random.Generator(np.random.PCG64(seed + 7))
        vel = rng.normal(0.0, np.sqrt(kB * vel_T / m), size=pos.shape)
        vel -= vel.mean(axis=0)
    case = {
        "box": box.tolist(), "dt": dt, "nsteps": nsteps,
        "species": species, "names": names, "vdw": vdw, "types": types,
        "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(),
        "vx": vel[:, 0].copy(), "vy": vel[:, 1].copy(), "vz": vel[:, 2].copy(),
        "elec_type": ELEC_TYPES[elec], "rReal": r_real if elec != "none" else 0.0, "alpha": alpha if elec != "none" else 0.0,
        "T": T, "tstat_type": {"none": 0, "nose": 1, "radi": 2}[tstat], "nEq": nEq, "freqEq": freqEq,
        "use_clist": 1, "cell_list": cell_list if cell_list is not None else rc,
        "center_box": 0, "init_forces": 1, "radii": radii, "seed": 12345,
    }
    return case


BOND_TYPES = {"harm": 1, "mors": 2, "pdn": 3, "buck": 4, "e612": 5}       # bonds.cpp:158-252
BOND_NAMES = {v: k for k, v in BOND_TYPES.items()}
BOND_NPARAM = {1: 2, 2: 4, 3: 5, 4: 3, 5: 5}


def molecular_case(ngrid, a=3.6, seed=77, rc=7.5, dt=0.0005, charges=None, elec="none", r_real=7.5, alpha=0.35,
                   vel_T=120.0, bond_mix=True, cell_list=None, quantize=True):
    """Bent triatomic molecules L-C-L on a simple cubic grid of ngrid=(nx,ny,nz) sites with spacing a, random
    orientation: the synthetic bonded input SURVEY Appendix G asks for ("next" row f2: constant bonds + hcos angles).

    Species 0 = C (central), 1 = L (ligand).  Only C-C has a pair potential (LJ): the reference does not exclude
    bonded pairs from the non-bonded loop (no exclusion list in pair_1 / pair_inter), so a C-L or L-L entry would
    act inside the molecule too.  bond_mix cycles the five bond potentials of bonds.cpp:731-787 over the molecules,
    tuned to the same minimum (r0 = 1.0) so that every branch of bond_iter is exercised; every second bond is listed
    ligand-first to exercise the turn of read_bondlist (bonds.cpp:62-67).
    """
    nx, ny, nz = ngrid
    rng = np.random.Generator(np.random.PCG64(seed))
    ii, jj, kk = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    cen = (np.stack([ii.ravel(), jj.ravel(), kk.ravel()], axis=1) + 0.5) * a
    cen = cen + rng.uniform(-0.1, 0.1, size=cen.shape)
    M = cen.shape[0]
    box = np.array([nx * a, ny * a, nz * a], dtype=np.float64)
    r0, cos0 = 1.0, -0.33
    # random orthonormal pair (u, w) per molecule; ligands at angle acos(cos0) +- a little
    u = rng.normal(size=(M, 3)); u /= np.linalg.norm(u, axis=1)[:, None]
    w = rng.normal(size=(M, 3)); w -= (w * u).sum(1)[:, None] * u; w /= np.linalg.norm(w, axis=1)[:, None]
    th = np.arccos(cos0) + rng.uniform(-0.15, 0.15, size=M)
    l1 = cen + (r0 + rng.uniform(-0.05, 0.05, size=(M, 1))) * u
    l2 = cen + (r0 + rng.uniform(-0.05, 0.05, size=(M, 1))) * (np.cos(th)[:, None] * u + np.sin(th)[:, None] * w)
    pos = np.empty((3 * M, 3)); pos[0::3] = cen; pos[1::3] = l1; pos[2::3] = l2
    pos = np.mod(pos, box)
    if quantize:
        pos = np.round(pos, 6); box = np.round(box, 6)
    pos[pos >= box] = 0.0
    types = np.tile(np.array([0, 1, 1], dtype=np.int32), M)
    qC, qL = charges if charges else (0.0, 0.0)
    species = [(15.999, qC), (1.008, qL)]
    k_h = 30.0
    bond_types = [(0, 1, 1, [k_h, r0])]
    if bond_mix:
        D, aa = 4.0, 2.0                               # Morse: k = 2 D a^2 = 32
        bond_types += [(0, 1, 2, [D, aa, r0, 0.5]),
                       (0, 1, 3, [D, aa, r0, 0.5, 0.002]),
                       (1, 0, 4, [2.0e4, 0.1, 1.513]),   # listed L-C on purpose; buck minimum at r = 1.0 needs ro < 1/7
                       (0, 1, 5, [2.0e4, 0.1, 1.1467, 0.2, 0.05])]
    angle_types = [(0, 1, [3.0, cos0]), (0, 1, [1.5, -0.5])]
    nbt = len(bond_types)
    bonds, angles = [], []
    for m in range(M):
        c, a1, a2 = 3 * m, 3 * m + 1, 3 * m + 2
        bonds.append((c, a1, 1 + (m % nbt)))
        bonds.append((a2, c, 1 + ((m + 1) % nbt)))
        angles.append((c, a1, a2, 1 + (m % 2)))
    vel = np.zeros_like(pos)
    if vel_T:
        kB = 1.3806488E-23 / 1.60217733E-19
        msc = 1.6605402E-27 / (1.60217733E-19 * 1e-24 / 1e-20)
        mass = np.where(types == 0, species[0][0], species[1][0]) * msc
        vel = rng.normal(size=pos.shape) * np.sqrt(kB * vel_T / mass)[:, None]
        vel -= (vel * mass[:, None]).sum(0) / mass.sum()
    return {
        "box": box.tolist(), "dt": dt, "nsteps": 0,
        "species": species, "names": ["C", "L"], "vdw": [(0, 0, VDW_TYPES["lnjs"], rc, [0.0067, 3.15])], "types": types,
        "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(),
        "vx": vel[:, 0].copy(), "vy": vel[:, 1].copy(), "vz": vel[:, 2].copy(),
        "elec_type": ELEC_TYPES[elec], "rReal": r_real if elec != "none" else 0.0, "alpha": alpha if elec != "none" else 0.0,
        "T": 120.0, "tstat_type": 0, "nEq": 0, "freqEq": 1,
        "use_clist": 1, "cell_list": cell_list if cell_list is not None else rc,
        "center_box": 0, "init_forces": 1, "radii": None, "seed": 12345,
        "bond_types": bond_types, "angle_types": angle_types,
        "bonds": np.array(bonds, dtype=np.int32), "angles": np.array(angles, dtype=np.int32),
    }


def write_input_files(case, directory, stat=200):
    """Write atoms.xyz / field.txt / control.txt / cuda.txt for `case` (reference grammar)."""
    os.makedirs(directory, exist_ok=True)
    names = case.get("names") or ["S%d" % i for i in range(len(case["species"]))]
    N = len(case["types"])
    with open(os.path.join(directory, "atoms.xyz"), "w") as f:
        f.write("%d\n" % N)
        f.write("1 %f %f %f\n" % tuple(case["box"]))
        t = np.asarray(case["types"])
        x, y, z = case["x"], case["y"], case["z"]
        lines = ["%s\t%f\t%f\t%f\n" % (names[t[i]], x[i], y[i], z[i]) for i in range(N)]
        f.writelines(lines)
    with open(os.path.join(directory, "field.txt"), "w") as f:
        f.write("spec %d\n" % len(case["species"]))
        for nm, (m, q) in zip(names, case["species"]):
            f.write("%s\t%s\t%r\t%r\t0.0\n" % (nm, nm, m, q))
        f.write("red-ox 0\n")
        f.write("vdw %d\n" % len(case["vdw"]))
        for a, b, t, rc, p in case["vdw"]:
            ps = "\t".join(repr(float(v)) for v in list(p)[:VDW_NPARAM[t]])
            f.write("%s\t%s\t%s\t%r\t%s\n" % (names[a], names[b], VDW_NAMES[t], float(rc), ps))
        if case.get("bond_types"):
            f.write("bonds %d\n" % len(case["bond_types"]))
            for i, (a, b, t, p) in enumerate(case["bond_types"]):
                ps = "\t".join(repr(float(v)) for v in list(p)[:BOND_NPARAM[t]])
                f.write("%d\t%s\t%s\t%s\t%s\tcon\tcon\n" % (i + 1, names[a], names[b], BOND_NAMES[t], ps))
        if case.get("angle_types"):
            f.write("angles %d\n" % len(case["angle_types"]))
            for i, (c, t, p) in enumerate(case["angle_types"]):
                f.write("%d\t%s\thcos\t%r\t%r\n" % (i + 1, names[c], float(p[0]), float(p[1])))
        if case.get("bonds") is not None and len(case["bonds"]):
            f.write("bond_list 1\n")
        if case.get("angles") is not None and len(case["angles"]):
            f.write("angle_list 1\n")
        if case.get("radii"):
            f.write("radii %d\n" % len(case["species"]))
            for nm, r in zip(names, case["radii"]):
                f.write("%s\t%r\t%r\t%r\n" % (nm, r[0], r[1], r[2]))
    with open(os.path.join(directory, "control.txt"), "w") as f:
        f.write("timestep %r ps\n" % case["dt"])
        f.write("nstep %d\n" % max(int(case.get("nsteps", 0)), 0))
        f.write("nequil %d\n" % case.get("nEq", 0))
        f.write("eqfreq %d\n" % case.get("freqEq", 1))
        ts = TSTAT_NAMES[case.get("tstat_type", 0)]
        extra = "\t0.2" if ts == "radi" else ("\t%r" % float(case.get("tau", 0.0)) if ts == "nose" else "")
        f.write("temperature %r\t%s%s\n" % (float(case.get("T", 0.0)), ts, extra))
        f.write("init_vel\tzero\n")
        f.write("permittivity 1.0\n")
        f.write("cell_list\t%r\n" % float(case.get("cell_list", 8.5)))
        et = ELEC_NAMES[case.get("elec_type", 0)]
        if et == "none":
            f.write("elec\tnone\n")
        elif et == "dir":
            f.write("elec\tdir\t%r\n" % case["rReal"])
        elif et == "pme":
            f.write("elec\tpme\t%r\t%r\t%d\t%d\t%d\n" % ((case["rReal"], case["alpha"]) + tuple(int(v) for v in case["ewald_k"])))
        else:
            f.write("elec\t%s\t%r\t%r\n" % (et, case["rReal"], case["alpha"]))
        f.write("rdf\t8.0\t0.02\t1000000\t1000000\tnucl\n")
        f.write("stat\t%d\n" % stat)
    if case.get("bonds") is not None and len(case["bonds"]):
        with open(os.path.join(directory, "bonds.txt"), "w") as f:       # read_bondlist, bonds.cpp:25-110
            f.write("%d\n" % len(case["bonds"]))
            f.writelines("%d %d %d\n" % tuple(int(v) for v in row) for row in case["bonds"])
    if case.get("angles") is not None and len(case["angles"]):
        with open(os.path.join(directory, "angles.txt"), "w") as f:      # read_anglelist, angles.cpp:22-60
            f.write("%d\n" % len(case["angles"]))
            f.writelines("%d %d %d %d\n" % tuple(int(v) for v in row) for row in case["angles"])
    with open(os.path.join(directory, "cuda.txt"), "w") as f:
        f.write("nstep stat 50\nnthread a 16\nnthread b 32\n")
    return directory


# named configurations of SURVEY.md 8(d)
def config(name):
    if name == "F1":      # 500-atom FCC(5^3)
        return lj_case((5, 5, 5), a=5.26, seed=11, rc=6.5, cell_list=6.5)
    if name == "F2":      # 4 000-atom probe system, rc 8.5
        return lj_case((10, 10, 10), a=5.26, seed=12345)
    if name == "F3":      # F2 with +-0.2 charges + Fennell
        return lj_case((10, 10, 10), a=5.26, seed=12345, charges=(0.2, -0.2), elec="fenn")
    if name == "C2":      # 40 000 Ar LJ
        return lj_case((20, 20, 25), seed=20240501)
    if name == "C3":      # 1 000 188 atoms, LJ + Fennell
        return lj_case((63, 63, 63), seed=20240502, charges=(0.2, -0.2), elec="fenn")
    if name == "C4":      # 1 000 188 atoms, pure LJ
        return lj_case((63, 63, 63), seed=20240502)
    if name == "C4T":     # C4 thermalised: Maxwell velocities at argon's 85 K (`init_vel gaus` state; the reference's cost does not depend on temperature, ours does)
        return lj_case((63, 63, 63), seed=20240502, vel_T=85.0)
    if name == "C3T":     # C3 thermalised at 85 K
        return lj_case((63, 63, 63), seed=20240502, charges=(0.2, -0.2), elec="fenn", vel_T=85.0)
    if name == "C2T":     # C2 thermalised at 85 K (small twin of C4T for parity tests)
        return lj_case((20, 20, 25), seed=20240501, vel_T=85.0)
    if name == "C4X":     # C4's lattice in a box that is an exact multiple of the cut-off (42 x 8.5 A): no accidental overhang of the cells over rc
        return lj_case((63, 63, 63), a=42 * 8.5 / 63, seed=20240502)
    if name == "C4L":     # the C4 liquid on a lattice of 64^3 cells (1 048 576 atoms): with cells of 9.176 A (--cell-size) the box is 40 cell layers = 8 ranks x 5 layers = 8 x 8
                          # lattice cells, so ONE emulated rank of 8 that exchanges its halo with itself (bench.py --emulate-ranks 8) sees a physical seam
        return lj_case((64, 64, 64), seed=20240508)
    if name == "C4LT":    # C4L thermalised at 85 K: what a slab rank of 8 costs when its atoms move (its sort interval is then a third of the cold lattice's)
        return lj_case((64, 64, 64), seed=20240508, vel_T=85.0)
    if name == "C1":      # synthetic twin of 'case study 1': 40 000 Ar gas atoms in a 1141.5 A box, LJ rc 4 A, cell_list 85 A, radiative thermostat
        rng = np.random.Generator(np.random.PCG64(20240506))
        N, L = 40000, 1141.5
        g = 35                                              # 35^3 = 42 875 lattice sites, 32.6 A apart, +-12 A jitter: a gas without overlaps
        site = rng.permutation(g ** 3)[:N]
        pos = np.stack([site // (g * g), (site // g) % g, site % g], axis=1) * (L / g) + L / (2 * g) + rng.uniform(-12.0, 12.0, size=(N, 3))
        pos = np.round(np.mod(pos, L), 6)
        pos[pos >= L] = 0.0
        return {"box": [L, L, L], "dt": 0.001, "nsteps": 0, "species": [(AR_MASS, 0.0)], "names": ["Ar"],
                "vdw": [(0, 0, VDW_TYPES["lnjs"], 4.0, [AR_EPS, AR_SIGMA])], "types": np.zeros(N, dtype=np.int32),
                "x": pos[:, 0].copy(), "y": pos[:, 1].copy(), "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N),
                "elec_type": 0, "rReal": 0.0, "alpha": 0.0, "T": 298.0, "tstat_type": 2, "nEq": 0, "freqEq": 1, "use_clist": 1,
                "cell_list": 85.0, "center_box": 0, "init_forces": 1, "radii": None, "seed": 12345}
    if name == "B3":      # 1 000 188 ions, Born-Mayer-Huggins + Fennell: the generic (any-potential) pair path at full size
        c = lj_case((63, 63, 63), seed=20240505, charges=(0.2, -0.2), elec="fenn")
        c["vdw"] = [(0, 0, 4, 8.5, [0.25, 3.1, 3.3, 60.0, 80.0]), (0, 1, 4, 8.5, [0.25, 3.1, 3.2, 50.0, 60.0]), (1, 1, 4, 8.5, [0.25, 3.1, 3.4, 70.0, 90.0])]
        return c
    if name == "E2":      # 40 000 ions, LJ + full Ewald sum ('elec pme 8.5 0.35 12 12 14'; "next" row f4)
        c = lj_case((20, 20, 25), seed=20240504, charges=(0.2, -0.2), elec="fenn", r_real=8.5, alpha=0.35)
        c.update(elec_type=2, ewald_k=(12, 12, 14))
        return c
    if name in ("S4", "S40"):   # periodic analogues of 'case study 2' (BASELINE config 5): surk rc 6.0 on 2.7 A cells, radii, radiative thermostat at 500 K;
                                # 4 000 atoms at the case study's density (0.46 / A^3: ~410 neighbours inside rc) and a 40 000-atom box of the same
        c = lj_case((10, 10, 10) if name == "S4" else (20, 20, 25), a=2.06, jitter=0.05, seed=20240507, rc=6.0, cell_list=2.7, T=500.0, tstat="radi",
                    radii=[(2.73, 4.731, 0.2)])
        c["vdw"] = [(0, 0, VDW_TYPES["surk"], 6.0, [75.0, 8.0, 1.0, 1.0])]
        return c
    if name == "M4":      # 1 029 000 atoms: 343 000 bent triatomics (bonds + angles), LJ + Fennell  ("next" row f2)
        return molecular_case((70, 70, 70), seed=20240503, charges=(-0.2, 0.1), elec="fenn", quantize=False)
    raise KeyError(name)
