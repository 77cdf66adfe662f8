"""TEST INFRASTRUCTURE (oracle/): independent Python restatement of the reference's input parser.

Follows the grammar *as implemented* by the reference:
  * keyword scanning find_int / find_double / find_str / find_number (utils.cpp:87-195): rewind, try the scanf
    template at the current position, on mismatch swallow one whitespace-delimited token (characters the partial
    match already consumed stay consumed), repeat until EOF - so directives are order-free, the first match wins
    and everything else is a comment;
  * read_field (sys_init.cpp:174-485), read_vdw (vdw.cpp:234-308), read_atoms_box (sys_init.cpp:487-565),
    read_sim (sys_init.cpp:590-989), read_tstat (temperature.cpp:91-259), read_elec (elec.cpp:14-67),
    read_cuda (cuInit.cu:684-754) and the derived parameters of init_md (sys_init.cpp:1036-1119).
Used by the CPU tests to check the product's C++ parser (aztotmd_amd/csrc/sys_init.cpp) - two independent
implementations of the same text format.
"""
import math
import os
import re

VDW_TYPES = {"lnjs": 1, "buck": 2, "p746": 3, "bmhs": 4, "elin": 5, "einv": 6, "surk": 7}
VDW_NPARAM = {1: 2, 2: 3, 3: 3, 4: 5, 5: 3, 6: 3, 7: 4}
BOND_TYPES = {"harm": 1, "mors": 2, "pdn": 3, "buck": 4, "e612": 5}
BOND_NPARAM = {1: 2, 2: 4, 3: 5, 4: 3, 5: 5}

# const.h:17-49
_E_SI, _Q_SI, _KB_SI, _E0_SI, _AMU_SI = 1.60217733E-19, 1.60217657E-19, 1.3806488E-23, 8.854187817E-12, 1.6605402E-27
PI = 3.14159265359
M_SCALE = _AMU_SI / (_E_SI * 1e-24 / 1e-10 / 1e-10)
KB = _KB_SI / _E_SI
FCOUL = (0.25 / PI / _E0_SI * _Q_SI * _Q_SI / 1e-10 / 1e-10) / (_E_SI / 1e-10)

_INT = re.compile(r"[+-]?\d+")
_FLT = re.compile(r"[+-]?(?:\d+\.?\d*(?:[eE][+-]?\d+)?|\.\d+(?:[eE][+-]?\d+)?|inf|nan)", re.I)


class Scanner:
    """A FILE* with the handful of fscanf behaviours the reference relies on."""

    def __init__(self, text):
        self.s = text
        self.p = 0

    def eof(self):
        return self.p >= len(self.s)

    def skip_ws(self):
        while self.p < len(self.s) and self.s[self.p].isspace():
            self.p += 1

    def scan(self, templ):
        """fscanf(f, templ, &value) for templates made of literals, white space and ONE conversion (%d %lf %s).
        Returns the converted value or None; consumed input stays consumed, exactly as with fscanf."""
        i = 0
        while i < len(templ):
            c = templ[i]
            if c.isspace():
                self.skip_ws()
                i += 1
            elif c == "%":
                j = i + 1
                while templ[j].isdigit():
                    j += 1
                conv = templ[j:j + 2] if templ[j] == "l" else templ[j]
                self.skip_ws()
                if conv == "d":
                    m = _INT.match(self.s, self.p)
                    if not m:
                        return None
                    self.p = m.end()
                    val = int(m.group())
                elif conv in ("lf", "f"):
                    m = _FLT.match(self.s, self.p)
                    if not m:
                        return None
                    self.p = m.end()
                    val = float(m.group())
                else:  # %s
                    m = re.compile(r"\S+").match(self.s, self.p)
                    if not m:
                        return None
                    self.p = m.end()
                    val = m.group()
                # trailing template white space swallows input white space
                rest = templ[j + len(conv):]
                if rest and rest.isspace():
                    self.skip_ws()
                return val
            else:
                if self.p < len(self.s) and self.s[self.p] == c:
                    self.p += 1
                    i += 1
                else:
                    return None
        return None

    def token(self):
        self.skip_ws()
        m = re.compile(r"\S+").match(self.s, self.p)
        if not m:
            self.p = len(self.s)
            return None
        self.p = m.end()
        return m.group()

    def find(self, templ):
        """find_int / find_double / find_str: utils.cpp:87-195"""
        self.p = 0
        while not self.eof():
            v = self.scan(templ)
            if v is not None:
                return v
            if self.token() is None:
                break
        return None

    def next(self, kind):
        return self.scan({"d": "%d", "f": "%lf", "s": "%s"}[kind])


def _read(path):
    with open(path, "r", errors="replace") as f:
        return f.read()


def prepare_vdw(t, rc, p):
    """read_vdw: vdw.cpp:261-299 (r_scale = E_scale = 1)"""
    p = list(p) + [0.0] * (5 - len(p))
    d = {"type": t, "r2cut": rc * rc, "rcut": rc, "p0": p[0], "p1": p[1], "p2": p[2], "p3": p[3], "p4": p[4], "use_radii": 0}
    if t == 1:
        d["p0"] *= 4
        d["p1"] = d["p1"] * d["p1"]
        d["p2"] = 6 * d["p0"]
        d["p3"] = d["p4"] = 0.0
    elif t in (2, 3, 5, 6):
        d["p3"] = d["p4"] = 0.0
    elif t == 7:
        d["p4"] = 0.0
        d["use_radii"] = 1
    return d


def parse_dir(directory, with_atoms=True):
    out = {"warnings": []}
    # ---- field.txt
    f = Scanner(_read(os.path.join(directory, "field.txt")))
    n = f.find(" spec %d")
    if not n:
        raise ValueError("ERROR[004] no 'spec' section")
    species = []
    for _ in range(n):
        name, nucl = f.next("s"), f.next("s")
        mass, charge, energy = f.next("f"), f.next("f"), f.next("f")
        species.append({"name": name, "nucleus": nucl, "mass_amu": mass, "mass": mass * M_SCALE, "charge": charge,
                        "charged": 0 if abs(charge) < 1e-10 else 1, "frozen": 0, "radA": 0.0, "radB": 0.0, "mxEng": 0.0, "number": 0})
    names = [s["name"] for s in species]
    charged_spec = int(any(s["charge"] != 0.0 for s in species))
    nf = f.find(" frozensp %d")
    if nf:
        for _ in range(nf):
            nm = f.next("s")
            if nm in names:
                species[names.index(nm)]["frozen"] = 1
    ns = len(species)
    vdw = [[None] * ns for _ in range(ns)]
    max_rvdw = 0.0
    vdw_raw = []          # (specA, specB, type id, rc, parameters as written): the array entry points take input units
    nv = f.find(" vdw %d")
    for _ in range(nv or 0):
        a, b, c = f.next("s"), f.next("s"), f.next("s")
        rc, p0, p1 = f.next("f"), f.next("f"), f.next("f")
        t = VDW_TYPES[c]
        p = [p0, p1] + [f.next("f") for _ in range(VDW_NPARAM[t] - 2)]
        pp = prepare_vdw(t, rc, p)
        max_rvdw = max(max_rvdw, rc)
        ia, ib = names.index(a), names.index(b)
        vdw_raw.append((ia, ib, t, rc, list(p)))
        vdw[ia][ib] = pp
        if t != 7:
            vdw[ib][ia] = pp
    # bonds / angles sections (sys_init.cpp:289-314,411-427 ; read_bond bonds.cpp:125-364 ; read_angle angles.cpp:78-128)
    bond_types, angle_types = [], []
    for _ in range(f.find(" bonds %d") or 0):
        f.next("d")
        a, b, key = f.next("s"), f.next("s"), f.next("s")
        t = BOND_TYPES[key]
        p = [f.next("f") for _ in range(BOND_NPARAM[t])]
        tail = [f.next("s"), f.next("s")]
        if tail != ["con", "con"]:
            raise ValueError("out of scope: variable bonds")
        bond_types.append({"type": t, "spec1": names.index(a), "spec2": names.index(b), "p": p + [0.0] * (5 - len(p))})
    for _ in range(f.find(" angles %d ") or 0):
        f.next("d")
        a, key = f.next("s"), f.next("s")
        if key != "hcos":
            raise ValueError("ERROR[012]")
        angle_types.append({"type": 1, "central": names.index(a), "k": f.next("f"), "cos0": f.next("f")})
    out.update(bond_types=bond_types, angle_types=angle_types, bond_list=f.find(" bond_list %d") is not None,
               angle_list=f.find(" angle_list %d") is not None)
    if f.find(" radii %d") is not None:
        for _ in range(ns):
            nm = f.next("s")
            sp = species[names.index(nm)]
            sp["radA"], sp["radB"], sp["mxEng"] = f.next("f"), f.next("f"), f.next("f")
    out.update(species=species, vdw=vdw, vdw_raw=vdw_raw, n_vdw=nv or 0, max_rvdw=max_rvdw)

    # ---- atoms.xyz
    a = Scanner(_read(os.path.join(directory, "atoms.xyz")))
    N = a.next("d")
    btype = a.next("d")
    if btype != 1:
        raise ValueError("ERROR[008] unknown box type")
    box = [a.next("f"), a.next("f"), a.next("f")]
    out.update(n_atoms=N, box=box)
    if with_atoms:
        types, xs, ys, zs = [], [], [], []
        for _ in range(N):
            nm = a.next("s")
            types.append(names.index(nm))
            xs.append(a.next("f")); ys.append(a.next("f")); zs.append(a.next("f"))
        for t in types:
            species[t]["number"] += 1
        out.update(types=types, x=xs, y=ys, z=zs)
        # bonds.txt / angles.txt (read_bondlist bonds.cpp:25-110 with its turn ; read_anglelist angles.cpp:22-60)
        bonds, angles = [], []
        if out["bond_list"] and os.path.exists(os.path.join(directory, "bonds.txt")):
            b = Scanner(_read(os.path.join(directory, "bonds.txt")))
            for _ in range(b.next("d")):
                a1, a2, k = b.next("d"), b.next("d"), b.next("d")
                bt = bond_types[k - 1]
                if bt["spec1"] != types[a1]:
                    a1, a2 = a2, a1
                if (bt["spec1"], bt["spec2"]) != (types[a1], types[a2]):
                    raise ValueError("ERROR [121-123] species do not match the bond type")
                bonds.append((a1, a2, k))
        if out["angle_list"] and angle_types and os.path.exists(os.path.join(directory, "angles.txt")):
            b = Scanner(_read(os.path.join(directory, "angles.txt")))
            for _ in range(b.next("d")):
                angles.append((b.next("d"), b.next("d"), b.next("d"), b.next("d")))
        out.update(bonds=bonds, angles=angles)

    # ---- control.txt
    c = Scanner(_read(os.path.join(directory, "control.txt")))
    dt = c.find(" timestep %lf ")
    if dt is None:
        raise ValueError("ERROR[411]")
    tsim = c.find(" timesim %lf ")
    if tsim is None:
        nstep = c.find(" nstep %d")
        if nstep is None:
            raise ValueError("ERROR[412]")
    else:
        nstep = int(tsim / dt)
    teq = c.find(" timeequil %lf ")
    nequil = (c.find(" nequil %d ") or 0) if teq is None else int(teq / dt)
    eqfreq = (c.find(" eqfreq %d ") or 0) if nequil else 0
    T = c.find(" temperature %lf ")
    if T is None:
        raise ValueError("ERROR[404]")
    ts = c.next("s")
    tstat = {"none": 0, "nose": 1, "radi": 2}[ts]
    tau = 0.0
    if ts == "nose":
        tau = c.next("f")
    elif ts == "radi":
        if c.next("d") is None:
            raise ValueError("ERROR[a002]")
    es = c.find(" elec %s")
    if es is None:
        raise ValueError("ERROR[401]")
    elec = {"none": 0, "dir": 1, "pme": 2, "fenn": 3}[es]
    r_real = alpha = 0.0
    ewald_k = [0, 0, 0]
    if es == "dir":
        r_real = c.next("f")
    elif es == "pme":
        r_real, alpha = c.next("f"), c.next("f")
        ewald_k = [c.next("d"), c.next("d"), c.next("d")]
    elif es == "fenn":
        r_real, alpha = c.next("f"), c.next("f")
    if not charged_spec and elec:
        elec = 0
    iv = c.find(" init_vel %s")
    if iv is None:
        raise ValueError("ERROR[406]")
    init_vel = {"zero": 0, "gaus": 1, "const": 2, "keng": 3}[iv]
    ivp = [0.0, 0.0, 0.0]
    if iv == "const":
        ivp = [c.next("f"), c.next("f"), c.next("f")]
    elif iv == "keng":
        ivp[0] = c.next("f")
    E = [0.0, 0.0, 0.0]
    ex = c.find(" elecfield %lf ")
    if ex is not None:
        E = [ex, c.next("f") or 0.0, c.next("f") or 0.0]
    cl = c.find(" cell_list %lf ")
    stat = c.find(" stat %d ")
    out.update(dt=dt, nstep=nstep, nequil=nequil, eqfreq=eqfreq, temperature=T, tstat_type=tstat, tau=tau, elec_type=elec,
               r_real=r_real, alpha=alpha, init_vel=init_vel, init_vel_par=ivp, elecfield=E, use_cell_list=int(cl is not None),
               cell_list=cl or 0.0, stat=stat if stat is not None else 1000, ewald_k=ewald_k)

    # ---- cuda.txt (optional)
    cp = os.path.join(directory, "cuda.txt")
    nthread = [16, 32, 10]
    if os.path.exists(cp):
        cu = Scanner(_read(cp))
        v = cu.find(" nthread a %d"); nthread[0] = v if v is not None else 16
        v = cu.find(" nthread b %d"); nthread[1] = v if v is not None else 32
        v = cu.find(" nstep stat %d"); nthread[2] = v if v is not None else 10
    out["nthread"] = nthread

    # ---- derived (init_md: sys_init.cpp:1053-1112 ; prepare_elec elec.cpp:399-405)
    for s in species:
        s["rMass_hdt"] = 0.5 * dt / s["mass"]
    rmax = r_real if elec else (max_rvdw if nv else 0.0)
    degfree = 3 * N - (1 if tstat else 0)
    out.update(rmax=rmax, r2max=rmax * rmax, degfree=degfree, tkin=0.5 * T * KB * degfree)
    if elec == 3:
        aRc = alpha * r_real
        daipi2 = 2 * alpha / math.sqrt(PI)
        out.update(daipi2=daipi2, scale=math.erfc(aRc) / r_real,
                   scale2=math.erfc(aRc) / (r_real * r_real) + daipi2 * math.exp(-aRc * aRc) / r_real)
    elif elec == 2:
        # prepare_elec elec.cpp:377-397 + the k-vector loop of ewald_rec elec.cpp:229-330 / cuInit.cu:1017-1046 + ewald_const :144-164
        twopi = 2.0 * PI
        ra, rb, rc = 1.0 / box[0], 1.0 / box[1], 1.0 / box[2]
        rvol = 1.0 / (box[0] * box[1] * box[2])
        scale = 2 * twopi * rvol * FCOUL / 1.0
        # ip1..3 through prepare_box's cell-matrix algebra (box.cpp:92-151): 1/la, 1/lb, 1/lc up to rounding, which matters for
        # k-vectors that sit exactly on the cut-off sphere
        la, lb, lc = box
        axb3, bxc1, cxa2 = la * lb, lb * lc, la * lc
        rdet, rv = 1.0 / (la * bxc1), 1.0 / (la * lb * lc)
        iax, iby, icz = rdet * bxc1, rdet * cxa2, rdet * axb3
        ip = [rv / math.sqrt((iby * icz) * (iby * icz)), rv / math.sqrt((iax * icz) * (iax * icz)), rv / math.sqrt((iax * iby) * (iax * iby))]
        rkcut = ewald_k[0] * ip[0]
        if rkcut > ewald_k[1] * ip[1]:
            rkcut = ewald_k[1] * ip[1]
        if rkcut > ewald_k[2] * ip[2]:
            rkcut = ewald_k[2] * ip[2]
        rkcut *= twopi * 1.05
        mr4a2 = -0.25 / alpha / alpha
        kvecs = []
        mmin, nmin = 0, 1
        for l in range(ewald_k[0]):
            for m in range(mmin, ewald_k[1]):
                for n in range(nmin, ewald_k[2]):
                    rk = (l * twopi * ra, m * twopi * rb, n * twopi * rc)
                    rk2 = rk[0] * rk[0] + rk[1] * rk[1] + rk[2] * rk[2]
                    if rk2 < rkcut * rkcut:
                        kvecs.append((l, m, n) + rk + (math.exp(rk2 * mr4a2) / rk2,))
                nmin = 1 - ewald_k[2]
            mmin = 1 - ewald_k[1]
        sq = eng = 0.0
        if with_atoms:
            for t in out["types"]:
                q = species[t]["charge"]
                sq += q
                eng += q * q
        eng *= (-1.0) * alpha / math.sqrt(PI)
        out.update(daipi2=2 * alpha / math.sqrt(PI), scale=scale, scale2=2 * scale, kvecs=kvecs,
                   eng_elec1=FCOUL * (eng + (-0.5 * PI * (sq * sq / alpha / alpha) * rvol)) / 1.0)
    else:
        out.update(daipi2=0.0, scale=0.0, scale2=0.0)
    return out
