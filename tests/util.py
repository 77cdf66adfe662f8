"""Shared helpers for the parity tests."""
import numpy as np

from aztotmd_amd import inputs


def rel_err(a, b):
    """max |a-b| / max |b| (array-level relative error: the north star's 'forces within 1e-9 relative')."""
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300))


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
