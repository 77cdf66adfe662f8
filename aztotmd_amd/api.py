"""ctypes front-end of the C ABI in include/aztot.h (mirrors the reference's host seam:
init_md -> init_cudaMD -> [step loop] -> md_to_host -> free_device_md; main.cu:239-463)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

STAT_FIELDS = ("step", "time", "engTot", "engKin", "engVdW", "engCoul", "engElecField", "engTemp", "engPot", "temperature",
               "posMom", "negMom", "posCross", "negCross", "pressure", "pairs_dropped", "n_cells", "nose_chit", "nose_conint",
               "engBond", "engAngle", "engCoulRec", "engCoulConst", "sort_interval", "sort_violations", "pair_lists", "cells_without_list", "rebuilds", "skin")


class AztotError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("aztot error %d: %s" % (code, msg))
        self.code = code


class _Species(C.Structure):
    _fields_ = [("name", C.c_char * 8), ("mass_amu", C.c_double), ("charge", C.c_double), ("frozen", C.c_int32),
                ("radA", C.c_double), ("radB", C.c_double), ("mxEng", C.c_double)]


class _Vdw(C.Structure):
    _fields_ = [("spec_a", C.c_int32), ("spec_b", C.c_int32), ("type", C.c_int32), ("rcut", C.c_double), ("p", C.c_double * 5)]


class _Control(C.Structure):
    _fields_ = [("timestep", C.c_double), ("nstep", C.c_int32), ("nequil", C.c_int32), ("eqfreq", C.c_int32),
                ("temperature", C.c_double), ("tstat_type", C.c_int32), ("tstat_tau", C.c_double), ("elec_type", C.c_int32),
                ("r_real", C.c_double), ("alpha", C.c_double), ("init_vel", C.c_int32), ("init_vel_par", C.c_double * 3),
                ("elecfield", C.c_double * 3), ("use_cell_list", C.c_int32), ("cell_list", C.c_double), ("stat", C.c_int32),
                ("ewald_k", C.c_int32 * 3)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class _System(C.Structure):
    _fields_ = [("n_atoms", C.c_int32), ("n_species", C.c_int32), ("n_vdw", C.c_int32), ("box", C.c_double * 3),
                ("types", _ip), ("x", _dp), ("y", _dp), ("z", _dp), ("vx", _dp), ("vy", _dp), ("vz", _dp),
                ("species", C.POINTER(_Species)), ("vdw", C.POINTER(_Vdw)), ("control", _Control)]


class _BondType(C.Structure):
    _fields_ = [("spec_a", C.c_int32), ("spec_b", C.c_int32), ("type", C.c_int32), ("p", C.c_double * 5)]


class _AngleType(C.Structure):
    _fields_ = [("central", C.c_int32), ("type", C.c_int32), ("k", C.c_double), ("cos0", C.c_double)]


class _Bonded(C.Structure):
    _fields_ = [("n_bond_types", C.c_int32), ("n_angle_types", C.c_int32), ("n_bonds", C.c_int32), ("n_angles", C.c_int32),
                ("bond_types", C.POINTER(_BondType)), ("angle_types", C.POINTER(_AngleType)),
                ("bond_a", _ip), ("bond_b", _ip), ("bond_type", _ip),
                ("angle_c", _ip), ("angle_l1", _ip), ("angle_l2", _ip), ("angle_type", _ip)]


class _Options(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("initial_forces", C.c_int32), ("center_box", C.c_int32), ("seed", C.c_uint64),
                ("pair_variant", C.c_int32), ("cell_size", C.c_double), ("use_graph", C.c_int32), ("profile", C.c_int32),
                ("sort_every", C.c_int32), ("skin", C.c_double), ("waves_per_cell", C.c_int32), ("energies_every_step", C.c_int32),
                ("loopback_ranks", C.c_int32), ("reserved", C.c_int32 * 5)]


class _Stats(C.Structure):
    _fields_ = [("step", C.c_int64), ("time", C.c_double), ("engTot", C.c_double), ("engKin", C.c_double), ("engVdW", C.c_double),
                ("engCoul", C.c_double), ("engElecField", C.c_double), ("engTemp", C.c_double), ("engPot", C.c_double),
                ("temperature", C.c_double), ("posMom", C.c_double * 3), ("negMom", C.c_double * 3), ("posCross", C.c_int64 * 3),
                ("negCross", C.c_int64 * 3), ("pressure", C.c_double), ("pairs_dropped", C.c_int64), ("n_cells", C.c_int64),
                ("nose_chit", C.c_double), ("nose_conint", C.c_double), ("engBond", C.c_double), ("engAngle", C.c_double),
                ("engCoulRec", C.c_double), ("engCoulConst", C.c_double), ("sort_interval", C.c_int64), ("sort_violations", C.c_int64),
                ("pair_lists", C.c_int64), ("cells_without_list", C.c_int64), ("rebuilds", C.c_int64), ("skin", C.c_double)]


class _Clock(C.Structure):
    _fields_ = [("step", C.c_int64), ("nose_chit", C.c_double), ("nose_conint", C.c_double), ("eng_kin", C.c_double)]


class _State(C.Structure):
    _fields_ = [("n_atoms", C.c_int32)] + [(k, _dp) for k in ("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")] + [("types", _ip)]


SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, _dp, C.c_int)

EXPORTS = ("aztot_device_count", "aztot_device_synchronize", "aztot_init_md", "aztot_model_create", "aztot_model_set_bonded", "aztot_model_query", "aztot_model_species_name", "aztot_free_md", "aztot_default_options",
           "aztot_init_device", "aztot_free_device", "aztot_step", "aztot_sync", "aztot_forces", "aztot_get_stats", "aztot_species_crossings", "aztot_md_to_host",
           "aztot_set_state", "aztot_get_clock", "aztot_set_clock", "aztot_cell_table", "aztot_kernel_times", "aztot_reset_kernel_times", "aztot_set_profile", "aztot_comm_id_bytes", "aztot_comm_make_id", "aztot_comm_selftest", "aztot_comm_ranks",
           "aztot_init_device_slab", "aztot_last_error", "aztot_version")


def library_path():
    # AZTOT_LIB lets an experiment load another build of the SAME library (kernel A/B comparisons); never a fallback
    return os.environ.get("AZTOT_LIB") or os.path.join(_HERE, "libaztot.so")


def build_library(force=False):
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    src_dir = os.path.join(_HERE, "csrc")
    so = library_path()
    newest = max(os.path.getmtime(os.path.join(src_dir, f)) for f in os.listdir(src_dir) if f.endswith((".h", ".hip", ".cpp")) or f == "Makefile")
    hdr = os.path.join(os.path.dirname(_HERE), "include", "aztot.h")
    newest = max(newest, os.path.getmtime(hdr))
    def stale():
        return force or not os.path.exists(so) or os.path.getmtime(so) < newest

    if stale():
        # N ranks started against a stale library would run N makes over the same object files: one builds, the others wait and find it done
        import fcntl
        with open(os.path.join(src_dir, ".build.lock"), "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if stale():
                    subprocess.check_call(["make", "-C", src_dir, "all"], stdout=subprocess.DEVNULL)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    return so


def lib():
    """Load libaztot.so.  There is deliberately no fallback: a missing library is an error."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not os.path.exists(so):
            raise ImportError("aztotmd_amd: %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950). There is no CPU fallback for the hot path." % so)
        L = C.CDLL(so)
        L.aztot_last_error.restype = C.c_char_p
        L.aztot_version.restype = C.c_char_p
        L.aztot_init_md.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.aztot_model_create.argtypes = [C.POINTER(_System), C.POINTER(C.c_void_p)]
        L.aztot_model_query.argtypes = [C.c_void_p, C.c_char_p, _dp, C.c_int]
        L.aztot_model_set_bonded.argtypes = [C.c_void_p, C.POINTER(_Bonded)]
        L.aztot_free_md.argtypes = [C.c_void_p]
        L.aztot_free_md.restype = None
        L.aztot_default_options.argtypes = [C.POINTER(_Options)]
        L.aztot_default_options.restype = None
        L.aztot_init_device.argtypes = [C.c_void_p, C.POINTER(_Options), C.POINTER(C.c_void_p)]
        L.aztot_init_device_slab.argtypes = [C.c_void_p, C.POINTER(_Options), C.c_int, C.c_int, C.c_void_p, SENDRECV_FN, ALLREDUCE_FN,
                                             C.c_void_p, C.POINTER(C.c_void_p)]
        L.aztot_free_device.argtypes = [C.c_void_p]
        L.aztot_free_device.restype = None
        L.aztot_step.argtypes = [C.c_void_p, C.c_int]
        L.aztot_sync.argtypes = [C.c_void_p]
        L.aztot_forces.argtypes = [C.c_void_p]
        L.aztot_get_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
        L.aztot_species_crossings.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
        L.aztot_md_to_host.argtypes = [C.c_void_p, C.POINTER(_State)]
        L.aztot_set_state.argtypes = [C.c_void_p, C.POINTER(_State)]
        L.aztot_get_clock.argtypes = [C.c_void_p, C.POINTER(_Clock)]
        L.aztot_set_clock.argtypes = [C.c_void_p, C.POINTER(_Clock)]
        L.aztot_cell_table.argtypes = [C.c_void_p, _ip, _ip, C.c_int, _ip, C.c_int]
        L.aztot_kernel_times.argtypes = [C.c_void_p, C.c_char_p, C.c_int, _dp, C.POINTER(C.c_int64), C.c_int]
        L.aztot_reset_kernel_times.argtypes = [C.c_void_p]
        L.aztot_set_profile.argtypes = [C.c_void_p, C.c_int]
        L.aztot_comm_make_id.argtypes = [C.c_void_p]
        L.aztot_comm_selftest.argtypes = [C.c_int]
        L.aztot_comm_ranks.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def _check(rc):
    if rc < 0:
        raise AztotError(rc, lib().aztot_last_error().decode("utf-8", "replace"))
    return rc


def _f8(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Model:
    """Host model (reference: Atoms/Field/Sim/Elec/TStat/Box after init_md, sys_init.cpp:1036)."""

    def __init__(self, handle, keep=None):
        self.h = handle
        self._keep = keep

    @classmethod
    def from_dir(cls, directory):
        h = C.c_void_p()
        _check(lib().aztot_init_md(os.fsencode(directory), C.byref(h)))
        return cls(h)

    @classmethod
    def from_case(cls, case):
        """`case`: dict as produced by aztotmd_amd.inputs.lj_case (input units)."""
        N = len(case["types"])
        types = np.ascontiguousarray(case["types"], dtype=np.int32)
        arrs = {k: _f8(case[k]) for k in ("x", "y", "z", "vx", "vy", "vz")}
        nsp = len(case["species"])
        sp = (_Species * nsp)()
        names = case.get("names") or ["S%d" % i for i in range(nsp)]
        radii = case.get("radii") or [(0.0, 0.0, 0.0)] * nsp
        frozen = case.get("frozen") or [0] * nsp
        for i, (m, q) in enumerate(case["species"]):
            sp[i].name = names[i].encode()[:7]
            sp[i].mass_amu, sp[i].charge, sp[i].frozen = m, q, int(frozen[i])
            sp[i].radA, sp[i].radB, sp[i].mxEng = radii[i]
        vd = (_Vdw * max(len(case["vdw"]), 1))()
        for i, (a, b, t, rc, p) in enumerate(case["vdw"]):
            vd[i].spec_a, vd[i].spec_b, vd[i].type, vd[i].rcut = a, b, t, rc
            for k, v in enumerate(list(p)[:5]):
                vd[i].p[k] = v
        s = _System()
        s.n_atoms, s.n_species, s.n_vdw = N, nsp, len(case["vdw"])
        s.box = (C.c_double * 3)(*case["box"])
        s.types = types.ctypes.data_as(_ip)
        for k in ("x", "y", "z", "vx", "vy", "vz"):
            setattr(s, k, arrs[k].ctypes.data_as(_dp))
        s.species, s.vdw = sp, vd
        c = s.control
        c.timestep, c.nstep, c.nequil, c.eqfreq = case["dt"], int(case.get("nsteps", 0)), int(case.get("nEq", 0)), int(case.get("freqEq", 1))
        c.temperature, c.tstat_type = float(case.get("T", 0.0)), int(case.get("tstat_type", 0))
        c.tstat_tau = float(case.get("tau", 0.0))
        c.elec_type, c.r_real, c.alpha = int(case.get("elec_type", 0)), float(case.get("rReal", 0.0)), float(case.get("alpha", 0.0))
        c.ewald_k = (C.c_int32 * 3)(*[int(v) for v in case.get("ewald_k", (0, 0, 0))])
        c.init_vel = 0
        c.elecfield = (C.c_double * 3)(case.get("Ux", 0.0), case.get("Uy", 0.0), case.get("Uz", 0.0))
        c.use_cell_list = int(case.get("use_clist", 1))
        c.cell_list = float(case.get("cell_list", 0.0) or 0.0)
        c.stat = int(case.get("stat", 200))
        h = C.c_void_p()
        _check(lib().aztot_model_create(C.byref(s), C.byref(h)))
        m = cls(h)
        if case.get("bond_types") or case.get("angle_types"):
            m.set_bonded(case.get("bond_types") or [], case.get("angle_types") or [], case.get("bonds"), case.get("angles"))
        return m

    def set_bonded(self, bond_types, angle_types, bonds=None, angles=None):
        """bond_types: [(specA, specB, type_id, [p..])], angle_types: [(central, type_id, [k, cos0])],
        bonds: (n,3) int array (at1, at2, type id 1-based) = bonds.txt, angles: (n,4) (central, lig1, lig2, type id) = angles.txt."""
        bt = (_BondType * max(len(bond_types), 1))()
        for i, (a, b, t, p) in enumerate(bond_types):
            bt[i].spec_a, bt[i].spec_b, bt[i].type = a, b, t
            for k, v in enumerate(list(p)[:5]):
                bt[i].p[k] = v
        at = (_AngleType * max(len(angle_types), 1))()
        for i, (c, t, p) in enumerate(angle_types):
            at[i].central, at[i].type, at[i].k, at[i].cos0 = c, t, p[0], p[1]
        b = np.ascontiguousarray(bonds if bonds is not None else np.zeros((0, 3)), dtype=np.int32).reshape(-1, 3)
        a = np.ascontiguousarray(angles if angles is not None else np.zeros((0, 4)), dtype=np.int32).reshape(-1, 4)
        bc = [np.ascontiguousarray(b[:, k]) for k in range(3)]
        ac = [np.ascontiguousarray(a[:, k]) for k in range(4)]
        d = _Bonded()
        d.n_bond_types, d.n_angle_types, d.n_bonds, d.n_angles = len(bond_types), len(angle_types), len(b), len(a)
        d.bond_types, d.angle_types = bt, at
        d.bond_a, d.bond_b, d.bond_type = [c.ctypes.data_as(_ip) for c in bc]
        d.angle_c, d.angle_l1, d.angle_l2, d.angle_type = [c.ctypes.data_as(_ip) for c in ac]
        _check(lib().aztot_model_set_bonded(self.h, C.byref(d)))

    def query(self, key, seed=None):
        n = _check(lib().aztot_model_query(self.h, key.encode(), None, 0))
        out = np.zeros(max(n, 1))
        if seed is not None:
            out[0] = float(seed)
        _check(lib().aztot_model_query(self.h, key.encode(), out.ctypes.data_as(_dp), n))
        return out[:n]

    def close(self):
        if self.h:
            lib().aztot_free_md(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """Device state + step driver (reference: cudaMD + the loop body of main.cu:281-410)."""

    def __init__(self, model, device=0, initial_forces=1, center_box=0, seed=12345, pair_variant=0, cell_size=0.0, use_graph=1,
                 profile=0, slab=None, debug=0, sort_every=0, split=0, skin=0.0, energies_every_step=0):
        """slab: None or dict(rank=, nranks=, rccl_id=bytes) or dict(rank=, nranks=, sendrecv=callable, allreduce=callable).
        debug: measurement / test switches (DebugBit in csrc/engine.h); they are not part of the ABI - the library reads them from the
        environment variable AZTOT_DEBUG when the device handle is created, so it is set around that call here."""
        L = lib()
        o = _Options()
        L.aztot_default_options(C.byref(o))
        o.device, o.initial_forces, o.center_box, o.seed = device, initial_forces, center_box, seed
        o.pair_variant, o.cell_size, o.use_graph, o.profile = pair_variant, cell_size, use_graph, profile
        o.sort_every = sort_every           # 0: adaptive lazy re-sort (default), 1: rebuild the cells every step, n: at most every n-th step
        o.skin = skin                       # Verlet skin in A (0: automatic, < 0: none)
        o.waves_per_cell = split            # staging kernel: waves per cell (0: the engine decides; 1, 2, 4, 8: forced - measurements)
        o.energies_every_step = energies_every_step
        self.model = model
        self.N = int(model.query("n_atoms")[0])
        self.h = C.c_void_p()
        self._cb = None
        old_dbg = os.environ.get("AZTOT_DEBUG")
        if debug:
            os.environ["AZTOT_DEBUG"] = str(int(debug) | int(old_dbg or "0", 0))
        try:
            self._create(L, model, o, slab)
        finally:
            if debug:
                if old_dbg is None:
                    os.environ.pop("AZTOT_DEBUG", None)
                else:
                    os.environ["AZTOT_DEBUG"] = old_dbg

    def _create(self, L, model, o, slab):
        if slab is None:
            _check(L.aztot_init_device(model.h, C.byref(o), C.byref(self.h)))
        else:
            idb = slab.get("rccl_id")
            if slab.get("loopback"):
                # measurement aid: one rank of an N-rank decomposition exchanging with itself (see LoopbackExchanger)
                o.loopback_ranks = 1
                _check(L.aztot_init_device_slab(model.h, C.byref(o), slab["rank"], slab["nranks"], None, SENDRECV_FN(), ALLREDUCE_FN(), None,
                                                C.byref(self.h)))
            elif idb is not None:
                buf = C.create_string_buffer(bytes(idb), len(idb))
                self._cb = (buf,)
                _check(L.aztot_init_device_slab(model.h, C.byref(o), slab["rank"], slab["nranks"], C.cast(buf, C.c_void_p),
                                                SENDRECV_FN(), ALLREDUCE_FN(), None, C.byref(self.h)))
            else:
                sr_py, ar_py = slab["sendrecv"], slab["allreduce"]

                def _sr(ctx, speer, sbuf, sbytes, rpeer, rbuf, rcap, rbytes):
                    try:
                        data = C.string_at(sbuf, sbytes)
                        got = sr_py(speer, data, rpeer, rcap)
                        C.memmove(rbuf, got, len(got))
                        rbytes[0] = len(got)
                        return 0
                    except Exception:   # noqa: BLE001 - reported through the C return code
                        import traceback
                        traceback.print_exc()
                        return 1

                def _ar(ctx, buf, n):
                    try:
                        a = np.ctypeslib.as_array(buf, shape=(n,))
                        a[:] = ar_py(a.copy())
                        return 0
                    except Exception:   # noqa: BLE001
                        import traceback
                        traceback.print_exc()
                        return 1

                self._cb = (SENDRECV_FN(_sr), ALLREDUCE_FN(_ar))
                _check(L.aztot_init_device_slab(model.h, C.byref(o), slab["rank"], slab["nranks"], None, self._cb[0], self._cb[1], None,
                                                C.byref(self.h)))

    def step(self, n=1):
        _check(lib().aztot_step(self.h, int(n)))

    def sync(self):
        """everything earlier calls queued or deferred has happened when this returns (aztot_sync): the end of a timed region"""
        _check(lib().aztot_sync(self.h))

    def forces(self):
        _check(lib().aztot_forces(self.h))

    def stats(self):
        s = _Stats()
        _check(lib().aztot_get_stats(self.h, C.byref(s)))
        d = {}
        for k in STAT_FIELDS:
            v = getattr(s, k)
            d[k] = list(v) if hasattr(v, "__len__") else v
        return d

    def species_crossings(self):
        """(n_species, 6) int array: crossings of the walls Xn, Xp, Yn, Yp, Zn, Zp per species (the columns of msd.dat)."""
        ns = int(self.model.query("n_species")[0])
        out = np.zeros(6 * ns, dtype=np.int64)
        _check(lib().aztot_species_crossings(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), out.size))
        return out.reshape(ns, 6)

    def state(self, keys=("x", "y", "z", "vx", "vy", "vz", "fx", "fy", "fz", "U", "radius")):
        """Per-atom arrays in ORIGINAL atom order.  On a slab rank only the owned atoms are filled (others NaN)."""
        st = _State()
        st.n_atoms = self.N
        out = {}
        for k in keys:
            out[k] = np.full(self.N, np.nan)
            setattr(st, k, out[k].ctypes.data_as(_dp))
        out["types"] = np.full(self.N, -1, dtype=np.int32)
        st.types = out["types"].ctypes.data_as(_ip)
        _check(lib().aztot_md_to_host(self.h, C.byref(st)))
        out["n_owned"] = st.n_atoms
        return out

    def set_state(self, **arrays):
        st = _State()
        st.n_atoms = self.N
        keep = []
        for k, v in arrays.items():
            a = _f8(v)
            keep.append(a)
            setattr(st, k, a.ctypes.data_as(_dp))
        _check(lib().aztot_set_state(self.h, C.byref(st)))

    def clock(self):
        """step number + thermostat scalars: with state() everything a restart needs (aztot_clock)"""
        c = _Clock()
        _check(lib().aztot_get_clock(self.h, C.byref(c)))
        return {"step": c.step, "nose_chit": c.nose_chit, "nose_conint": c.nose_conint, "eng_kin": c.eng_kin}

    def set_clock(self, step, nose_chit=0.0, nose_conint=0.0, eng_kin=0.0):
        c = _Clock(int(step), float(nose_chit), float(nose_conint), float(eng_kin))
        _check(lib().aztot_set_clock(self.h, C.byref(c)))

    def cell_table(self):
        """(dims, cell_start[n_cells + 1], atom_id[resident atoms]) of the sorted cell list the device holds."""
        dims = np.zeros(3, dtype=np.int32)
        n = _check(lib().aztot_cell_table(self.h, dims.ctypes.data_as(_ip), None, 0, None, 0))
        start = np.zeros(n + 1, dtype=np.int32)
        _check(lib().aztot_cell_table(self.h, dims.ctypes.data_as(_ip), start.ctypes.data_as(_ip), n + 1, None, 0))
        ids = np.zeros(max(int(start[-1]), 1), dtype=np.int32)
        _check(lib().aztot_cell_table(self.h, dims.ctypes.data_as(_ip), start.ctypes.data_as(_ip), n + 1, ids.ctypes.data_as(_ip), ids.size))
        return tuple(int(v) for v in dims), start, ids[:int(start[-1])]

    def comm_ranks(self):
        """ranks of the RCCL communicator carrying the halo (0: no RCCL in use)."""
        return _check(lib().aztot_comm_ranks(self.h))

    def kernel_times(self):
        names = C.create_string_buffer(4096)
        ms = (C.c_double * 64)()
        calls = (C.c_int64 * 64)()
        n = _check(lib().aztot_kernel_times(self.h, names, 4096, ms, calls, 64))
        parts = names.raw.split(b"\0")
        return {parts[i].decode(): {"ms": ms[i], "calls": calls[i]} for i in range(n)}

    def reset_kernel_times(self):
        _check(lib().aztot_reset_kernel_times(self.h))

    def set_profile(self, on):
        _check(lib().aztot_set_profile(self.h, int(bool(on))))

    def close(self):
        if getattr(self, "h", None):
            lib().aztot_free_device(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    """HIP devices visible to this process (through libaztot, i.e. the system ROCm runtime: importing torch into a process that later
    brings up RCCL would put torch's bundled, un-initialised HSA copy in front of it)."""
    return int(lib().aztot_device_count())


def device_synchronize(device=0):
    """hipDeviceSynchronize through libaztot (the bracket of a timed region; no torch needed in the process)"""
    _check(lib().aztot_device_synchronize(int(device)))


def rccl_unique_id():
    L = lib()
    n = L.aztot_comm_id_bytes()
    buf = C.create_string_buffer(n)
    _check(L.aztot_comm_make_id(buf))
    return buf.raw


def rccl_selftest(device=0):
    """One-rank RCCL communicator: ring exchange with itself + all-reduces (raises AztotError if RCCL is unusable here)."""
    _check(lib().aztot_comm_selftest(device))
