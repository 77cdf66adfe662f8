"""TEST INFRASTRUCTURE (oracle/): ctypes front-end of liboracle.so + runner of oracle/_ref/ref_driver.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

from . import casefile

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAT_FIELDS = ("engVdW", "engElec3", "engKin", "engTot", "engElecField", "engTemp", "Temp",
               "momXn", "momXp", "momYn", "momYp", "momZn", "momZp", "nDropped", "iStep", "tKin", "chit", "conint",
               "engBond", "engAngle", "engElec1", "engElec2")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when the reference sources are present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "aztot_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src"):
        drv = os.path.join(_HERE, "_ref", "ref_driver")
        dsrc = os.path.join(_HERE, "ref_driver.cpp")
        if force or not os.path.exists(drv) or os.path.getmtime(drv) < os.path.getmtime(dsrc):
            subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = C.CDLL(so)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_int, dp, ip] + [dp] * 6
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_set_species.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double]
        L.orc_set_vdw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, dp]
        L.orc_set_vdw.restype = C.c_int
        L.orc_set_elec.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_double]
        L.orc_set_control.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_double, C.c_uint64]
        L.orc_prepare.argtypes = [C.c_void_p]
        L.orc_set_nose.argtypes = [C.c_void_p, C.c_double]
        L.orc_set_ewald.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_center_box.argtypes = [C.c_void_p]
        L.orc_forces.argtypes = [C.c_void_p, C.c_int]
        L.orc_step.argtypes = [C.c_void_p, C.c_int]
        L.orc_stage_integrate1.argtypes = [C.c_void_p]
        L.orc_stage_integrate2.argtypes = [C.c_void_p, C.c_int]
        L.orc_stage_tstat.argtypes = [C.c_void_p, C.c_longlong]
        L.orc_get_state.argtypes = [C.c_void_p] + [dp] * 11
        L.orc_set_vel.argtypes = [C.c_void_p] + [dp] * 3
        L.orc_set_forces.argtypes = [C.c_void_p] + [dp] * 3
        L.orc_set_thermo.argtypes = [C.c_void_p] + [dp] * 2
        L.orc_get_stats.argtypes = [C.c_void_p, dp]
        L.orc_set_bond_types.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, dp]
        L.orc_set_angle_types.argtypes = [C.c_void_p, C.c_int, ip, ip, dp]
        L.orc_set_bond_list.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.orc_set_angle_list.argtypes = [C.c_void_p, C.c_int, ip, ip, ip, ip]
        L.orc_bond_pair.argtypes = [C.c_int, dp, C.c_double, dp]
        L.orc_bond_pair.restype = C.c_double
        L.orc_get_cross.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.orc_get_species_cross.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
        L.orc_cells.argtypes = [C.c_void_p, ip]
        L.orc_cells.restype = C.c_int
        L.orc_neig_table.argtypes = [C.c_void_p]
        L.orc_neig_table.restype = ip
        L.orc_photons.argtypes = [C.c_void_p]
        L.orc_photons.restype = dp
        L.orc_vdw_pair.argtypes = [C.c_int, C.c_double, dp, C.c_double, C.c_double, C.c_double, dp]
        L.orc_vdw_pair.restype = C.c_double
        L.orc_coul_pair.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, dp]
        L.orc_coul_pair.restype = C.c_double
        L.orc_elec_consts.argtypes = [C.c_void_p, dp]
        L.orc_photon_engs.argtypes = [C.c_int, dp, C.c_double, C.c_uint64]
        L.orc_unit_vectors.argtypes = [dp, dp, dp]
        L.orc_rng.argtypes = [C.c_uint64] * 4
        L.orc_rng.restype = C.c_uint32
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Oracle:
    """One oracle system.  `case` is the dict format of oracle/casefile.py, plus optional keys
    'radii' (list of (radA, radB, mxEng) per species), 'frozen', 'seed', 'Uy', 'Uz'."""

    def __init__(self, case):
        L = lib()
        self.L = L
        self.case = case
        N = len(case["types"])
        self.N = N
        types = np.ascontiguousarray(case["types"], dtype=np.int32)
        arr = [_f8(case[k]) for k in ("x", "y", "z", "vx", "vy", "vz")]
        box = _f8(case["box"])
        self.h = C.c_void_p(L.orc_create(N, len(case["species"]), _dp(box), types.ctypes.data_as(C.POINTER(C.c_int)),
                                          *[_dp(a) for a in arr]))
        radii = case.get("radii") or [(0.0, 0.0, 0.0)] * len(case["species"])
        frozen = case.get("frozen") or [0] * len(case["species"])
        for i, (m, q) in enumerate(case["species"]):
            L.orc_set_species(self.h, i, m, q, int(frozen[i]), *radii[i])
        for a, b, t, rc, p in case["vdw"]:
            pp = _f8(list(p) + [0.0] * (5 - len(p)))
            if not L.orc_set_vdw(self.h, a, b, t, rc, _dp(pp)):
                raise ValueError("bad vdw type %r" % (t,))
        L.orc_set_elec(self.h, case.get("elec_type", 0), case.get("rReal", 0.0), case.get("alpha", 0.0))
        L.orc_set_control(self.h, case["dt"], case.get("T", 0.0), case.get("tstat_type", 0), case.get("nEq", 0),
                          case.get("freqEq", 1), case.get("use_clist", 1), case.get("Ux", 0.0), case.get("Uy", 0.0),
                          case.get("Uz", 0.0), case.get("seed", 12345))
        L.orc_set_nose(self.h, case.get("tau", 0.0))
        L.orc_set_ewald(self.h, *[int(v) for v in case.get("ewald_k", (0, 0, 0))])
        L.orc_prepare(self.h)
        self._set_bonded(case)
        if case.get("center_box", 0):
            L.orc_center_box(self.h)

    def _set_bonded(self, case):
        L = self.L
        i4 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        ipt = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        bt, at = case.get("bond_types") or [], case.get("angle_types") or []
        if bt:
            s1, s2, tp = i4([b[0] for b in bt]), i4([b[1] for b in bt]), i4([b[2] for b in bt])
            pp = _f8([list(b[3]) + [0.0] * (5 - len(b[3])) for b in bt])
            if L.orc_set_bond_types(self.h, len(bt), ipt(s1), ipt(s2), ipt(tp), _dp(pp)):
                raise ValueError("bad bond type table")
        if at:
            ce, tp = i4([a[0] for a in at]), i4([a[1] for a in at])
            pp = _f8([list(a[2]) for a in at])
            if L.orc_set_angle_types(self.h, len(at), ipt(ce), ipt(tp), _dp(pp)):
                raise ValueError("bad angle type table")
        bonds = case.get("bonds")
        if bonds is not None and len(bonds):
            b = i4(bonds).reshape(-1, 3)
            cols = [i4(b[:, k]) for k in range(3)]
            rc = L.orc_set_bond_list(self.h, len(b), *[ipt(c) for c in cols])
            if rc:
                raise ValueError("bond list rejected (%d): species do not match the bond type" % rc)
        angles = case.get("angles")
        if angles is not None and len(angles):
            a = i4(angles).reshape(-1, 4)
            cols = [i4(a[:, k]) for k in range(4)]
            rc = L.orc_set_angle_list(self.h, len(a), *[ipt(c) for c in cols])
            if rc:
                raise ValueError("angle list rejected (%d)" % rc)

    def close(self):
        if self.h:
            self.L.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forces(self, mode=1):
        self.L.orc_forces(self.h, mode)

    def step(self, n=1):
        self.L.orc_step(self.h, n)

    def state(self):
        out = {k: np.empty(self.N) for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "rad")}
        self.L.orc_get_state(self.h, *[_dp(out[k]) for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "rad")])
        return out

    def stats(self):
        s = np.empty(22)
        self.L.orc_get_stats(self.h, _dp(s))
        d = dict(zip(STAT_FIELDS, s.tolist()))
        cr = (C.c_longlong * 6)()
        self.L.orc_get_cross(self.h, cr)
        d["cross"] = list(cr)
        return d

    def species_crossings(self):
        ns = len(self.case["species"])
        cr = (C.c_longlong * (6 * ns))()
        self.L.orc_get_species_cross(self.h, cr)
        return np.array(list(cr), dtype=np.int64).reshape(ns, 6)

    def set_vel(self, vx, vy, vz):
        self.L.orc_set_vel(self.h, _dp(_f8(vx)), _dp(_f8(vy)), _dp(_f8(vz)))

    def set_forces(self, fx, fy, fz):
        self.L.orc_set_forces(self.h, _dp(_f8(fx)), _dp(_f8(fy)), _dp(_f8(fz)))

    def set_thermo(self, U, rad):
        self.L.orc_set_thermo(self.h, _dp(_f8(U)), _dp(_f8(rad)))

    def cells(self):
        d = (C.c_int * 3)()
        n = self.L.orc_cells(self.h, d)
        return n, tuple(d)

    def consts(self):
        o = np.empty(8)
        self.L.orc_elec_consts(self.h, _dp(o))
        return dict(zip(("scale", "scale2", "daipi2", "rMax", "tKin", "kB", "m_scale", "Fcoul_scale"), o.tolist()))

    def photons(self):
        p = self.L.orc_photons(self.h)
        return np.ctypeslib.as_array(p, shape=(self.N,)).copy() if p else None


def vdw_pair(type_id, rc, params, r2, radi=0.0, radj=0.0):
    p = _f8(list(params) + [0.0] * (5 - len(params)))
    e = C.c_double(0.0)
    f = lib().orc_vdw_pair(type_id, rc, _dp(p), r2, radi, radj, C.byref(e))
    return f, e.value


def bond_pair(type_id, params, r2):
    """(-(1/r) dU/dr, U) of one bond potential (bond_iter, bonds.cpp:731-787)."""
    p = _f8(list(params) + [0.0] * (5 - len(params)))
    e = C.c_double(0.0)
    f = lib().orc_bond_pair(type_id, _dp(p), r2, C.byref(e))
    return f, e.value


def coul_pair(type_id, rReal, alpha, qi, qj, r2):
    e = C.c_double(0.0)
    f = lib().orc_coul_pair(type_id, rReal, alpha, qi, qj, r2, C.byref(e))
    return f, e.value


def ref_available():
    return os.path.exists(os.path.join(_HERE, "_ref", "ref_driver"))


def run_ref(case, timeout=600):
    """Run the compiled reference serial path (oracle/_ref/ref_driver) on `case`."""
    drv = os.path.join(_HERE, "_ref", "ref_driver")
    with tempfile.TemporaryDirectory() as td:
        cpath, opath = os.path.join(td, "case.bin"), os.path.join(td, "out.bin")
        casefile.write_case(cpath, case)
        r = subprocess.run([drv, cpath, opath], capture_output=True, text=True, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("ref_driver failed (%d): %s" % (r.returncode, r.stderr[-2000:]))
        out = casefile.read_ref_output(opath)
        try:
            out["summary"] = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            out["summary"] = {}
        return out
