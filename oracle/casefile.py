"""TEST INFRASTRUCTURE (oracle/): binary case-file I/O shared by oracle/ref_driver.cpp and the tests.

A *case* is one fully specified hot-path problem: atoms, box, species, pair potentials,
electrostatics mode, time step and which steps to dump.  Values are in the reference's *input*
units (Angstrom, ps, eV, e, amu) exactly as they would appear in field.txt / control.txt.

Layout (little endian):
  "AZTC" i32 version
  i32 N, nSpec, nVdw ; f64 box[3] ; f64 dt ; i32 nsteps
  i32 elec_type ; f64 rReal, alpha
  i32 use_clist, center_box, init_forces
  i32 nEq, freqEq ; f64 T ; i32 tstat_type
  f64 Ux ; f64 tau (Nose-Hoover relaxation time)
  i32 ndump ; i32 dump_steps[ndump]
  nSpec x (f64 mass_amu, f64 charge)
  nVdw  x (i32 a, i32 b, i32 type, f64 rc, f64 p[5])
  i32 type[N] ; f64 x[N], y[N], z[N], vx[N], vy[N], vz[N]
  (version >= 3) bonded section - field.txt 'bonds'/'angles' + bonds.txt / angles.txt:
  i32 nBondTypes x (i32 spec1, spec2, type(1 harm 2 mors 3 pdn 4 buck 5 e612), f64 p[5])
  i32 nAngleTypes x (i32 central, type(1 hcos), f64 k, cos0)
  i32 nBonds x (i32 at1, at2, type 1-based) ; i32 nAngles x (i32 central, lig1, lig2, type 1-based)
  (version >= 4) i32 kx, ky, kz  - 'elec pme rReal alpha kx ky kz' (read_elec, elec.cpp:33-38)

Output of ref_driver: i32 N, nHead, cn(packed), ndump ; per dump: i32 step, f64 e[16], 9 x f64[N].
"""
import struct
import numpy as np

VDW_TYPES = {"lnjs": 1, "buck": 2, "p746": 3, "bmhs": 4, "elin": 5, "einv": 6, "surk": 7}
ELEC_TYPES = {"none": 0, "dir": 1, "pme": 2, "fenn": 3}
ENERGY_FIELDS = ("engVdW", "engElec3", "engKin", "engTot", "engElecField", "Temp",
                 "momXn", "momXp", "momYn", "momYp", "momZn", "momZp", "engBond", "engAngle", "engElec1", "engElec2")
BOND_TYPES = {"harm": 1, "mors": 2, "pdn": 3, "buck": 4, "e612": 5}


def write_case(path, case):
    N = len(case["types"])
    species = case["species"]          # list of (mass_amu, charge)
    vdw = case["vdw"]                  # list of (a, b, type_id, rc, [p0..p4])
    dump = list(case.get("dump", [0]))
    with open(path, "wb") as f:
        f.write(b"AZTC")
        f.write(struct.pack("<i", 4))
        f.write(struct.pack("<iii", N, len(species), len(vdw)))
        f.write(struct.pack("<ddd", *case["box"]))
        f.write(struct.pack("<d", case["dt"]))
        f.write(struct.pack("<i", case.get("nsteps", 0)))
        f.write(struct.pack("<idd", case.get("elec_type", 0), case.get("rReal", 0.0), case.get("alpha", 0.0)))
        f.write(struct.pack("<iii", case.get("use_clist", 1), case.get("center_box", 0), case.get("init_forces", 1)))
        f.write(struct.pack("<iidi", case.get("nEq", 0), case.get("freqEq", 1), case.get("T", 0.0), case.get("tstat_type", 0)))
        f.write(struct.pack("<dd", case.get("Ux", 0.0), case.get("tau", 0.0)))
        f.write(struct.pack("<i", len(dump)))
        f.write(struct.pack("<%di" % len(dump), *dump))
        for m, q in species:
            f.write(struct.pack("<dd", m, q))
        for a, b, t, rc, p in vdw:
            p = list(p) + [0.0] * (5 - len(p))
            f.write(struct.pack("<iiid5d", a, b, t, rc, *p))
        f.write(np.asarray(case["types"], dtype="<i4").tobytes())
        for k in ("x", "y", "z", "vx", "vy", "vz"):
            f.write(np.ascontiguousarray(case[k], dtype="<f8").tobytes())
        bt, at = case.get("bond_types") or [], case.get("angle_types") or []
        f.write(struct.pack("<i", len(bt)))
        for s1, s2, t, p in bt:
            p = list(p) + [0.0] * (5 - len(p))
            f.write(struct.pack("<iii5d", s1, s2, t, *p))
        f.write(struct.pack("<i", len(at)))
        for c, t, p in at:
            f.write(struct.pack("<iidd", c, t, p[0], p[1]))
        bonds = np.asarray(case.get("bonds") if case.get("bonds") is not None else [], dtype="<i4").reshape(-1, 3)
        angles = np.asarray(case.get("angles") if case.get("angles") is not None else [], dtype="<i4").reshape(-1, 4)
        f.write(struct.pack("<i", len(bonds)))
        f.write(np.ascontiguousarray(bonds).tobytes())
        f.write(struct.pack("<i", len(angles)))
        f.write(np.ascontiguousarray(angles).tobytes())
        f.write(struct.pack("<iii", *[int(v) for v in case.get("ewald_k", (0, 0, 0))]))


def read_ref_output(path):
    with open(path, "rb") as f:
        N, nHead, cn, ndump = struct.unpack("<4i", f.read(16))
        out = {"N": N, "nHead": nHead, "cells": (cn // 1000000, (cn // 1000) % 1000, cn % 1000), "dumps": {}}
        for _ in range(ndump):
            hdr = f.read(4)
            if len(hdr) < 4:
                break
            (step,) = struct.unpack("<i", hdr)
            e = struct.unpack("<16d", f.read(128))
            d = dict(zip(ENERGY_FIELDS, e))
            for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz"):
                d[k] = np.frombuffer(f.read(8 * N), dtype="<f8").copy()
            out["dumps"][step] = d
    return out
