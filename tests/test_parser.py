"""CPU tests of the input surface: the product's C++ parser (libaztot.so, no GPU needed) against the independent
Python restatement in oracle/parse.py, on (a) generated inputs that exercise the grammar quirks of the reference's
scanner, (b) files written by aztotmd_amd.inputs, (c) the reference's shipped case studies when they are present."""
import os

import numpy as np
import pytest

from aztotmd_amd import api, inputs
from oracle import parse

QUIRKY_FIELD = """// comment words are simply not keywords
spec 2
Ar  Ar   39.9   0.0   0.0
Cl- Cl   35.45  -1.0  0.3
red-ox 0
frozensp 1 Cl-
vdw 3
Ar  Ar  lnjs 4.0    0.01006 3.3952
Ar  Cl- buck 6.0    300.0     0.7    12.5
Cl- Cl- bmhs 7.5    0.25 3.1 2.4 60.0 80.0
radii 1
Ar  2.73 4.731 0.2
Cl- 3.0  2.0   6.0

vdw 1
Ar Ar lnjs 9.9 9.9 9.9
"""
QUIRKY_CONTROL = """timestep 0.002 ps
nstep 1234
nequil  40
eqfreq 10
temperature 298.0\tradi\t0.2
// nose 0.2
init_vel\tzero\t0.0332
permittivity  1.0
cell_list\t85.0
max_neigh\t185
elec\tfenn\t8.0\t0.4\t6\t6\t6
rdf\t14.0   0.02\t50\t500000\tnucl
eJump\t0\t1.7\tmetr
Ux\t\t0.0
stat\t\t200
"""
QUIRKY_CUDA = "nstep stat 50\nnstep msdstat 50\nnthread a 16\nnthread b 48\nnstep traj\t10\n"


def write_quirky(d):
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, "field.txt"), "w").write(QUIRKY_FIELD)
    open(os.path.join(d, "control.txt"), "w").write(QUIRKY_CONTROL)
    open(os.path.join(d, "cuda.txt"), "w").write(QUIRKY_CUDA)
    with open(os.path.join(d, "atoms.xyz"), "w") as f:
        f.write("6\n1 40.000000 41.000000 42.000000\n")
        for i in range(6):
            f.write("%s\t%f\t%f\t%f\n" % ("Ar" if i % 2 else "Cl-", 1.5 * i + 0.25, 2.0 * i, 39.0 - i))


def compare(directory):
    m = api.Model.from_dir(directory)
    o = parse.parse_dir(directory)
    q = m.query
    assert int(q("n_atoms")[0]) == o["n_atoms"] and list(q("box")) == o["box"]
    for key, okey in (("dt", "dt"), ("nstep", "nstep"), ("nequil", "nequil"), ("eqfreq", "eqfreq"), ("temperature", "temperature"),
                      ("tstat_type", "tstat_type"), ("elec_type", "elec_type"), ("r_real", "r_real"), ("alpha", "alpha"),
                      ("cell_list", "cell_list"), ("use_cell_list", "use_cell_list"), ("stat", "stat"), ("rmax", "rmax"),
                      ("r2max", "r2max"), ("degfree", "degfree"), ("tkin", "tkin"), ("scale", "scale"), ("scale2", "scale2"),
                      ("daipi2", "daipi2")):
        assert q(key)[0] == pytest.approx(o[okey], rel=1e-15, abs=0), key
    assert list(q("elecfield")) == o["elecfield"]
    assert [int(v) for v in q("nthread")] == o["nthread"]
    assert q("kB")[0] == pytest.approx(parse.KB, rel=1e-15) and q("m_scale")[0] == pytest.approx(parse.M_SCALE, rel=1e-15)
    assert q("fcoul")[0] == pytest.approx(parse.FCOUL, rel=1e-15)
    sp = q("species").reshape(-1, 10)
    for row, s in zip(sp, o["species"]):
        ref = [s["mass_amu"], s["mass"], s["charge"], s["charged"], s["frozen"], s["rMass_hdt"], s["radA"], s["radB"], s["mxEng"], s["number"]]
        assert row.tolist() == pytest.approx(ref, rel=1e-15, abs=0)
    vd = q("vdw").reshape(len(o["species"]), len(o["species"]), 8)
    for a in range(len(o["species"])):
        for b in range(len(o["species"])):
            p = o["vdw"][a][b]
            if p is None:
                assert vd[a, b, 0] == 0
            else:
                ref = [p["type"], p["r2cut"], p["p0"], p["p1"], p["p2"], p["p3"], p["p4"], p["use_radii"]]
                assert vd[a, b].tolist() == pytest.approx(ref, rel=1e-15, abs=0), (a, b)
    assert [int(t) for t in q("types")] == o["types"]
    nb = [int(v) for v in q("n_bonded")]
    assert nb == [len(o["bond_types"]), len(o["angle_types"]), len(o["bonds"]), len(o["angles"])]
    for row, b in zip(q("bond_types").reshape(-1, 8), o["bond_types"]):
        assert row.tolist() == [b["type"], b["spec1"], b["spec2"]] + b["p"]
    for row, a in zip(q("angle_types").reshape(-1, 4), o["angle_types"]):
        assert row.tolist() == [a["type"], a["central"], a["k"], a["cos0"]]
    assert q("bonds").reshape(-1, 3).astype(int).tolist() == [list(b) for b in o["bonds"]]
    assert q("angles").reshape(-1, 4).astype(int).tolist() == [list(a) for a in o["angles"]]
    for k in ("x", "y", "z"):
        assert np.array_equal(q(k), np.array(o[k]))
    return m, o


def test_quirky_grammar(tmp_path):
    d = str(tmp_path / "quirky")
    write_quirky(d)
    m, o = compare(d)
    # first-match-wins: the second 'vdw' block is a comment; 'radi 0.2' parses the 0; the 'fenn' line's trailing 6 6 6 is ignored
    assert o["vdw"][0][0]["r2cut"] == 16.0 and o["tstat_type"] == 2 and o["elec_type"] == 3 and o["rmax"] == 8.0
    assert o["species"][1]["frozen"] == 1 and o["elecfield"] == [0.0, 0.0, 0.0] and o["nthread"] == [16, 48, 50]
    assert o["vdw"][0][1]["type"] == 2 and o["vdw"][1][0]["p2"] == 12.5 and o["vdw"][1][1]["p4"] == 80.0


def test_generated_inputs_round_trip(tmp_path):
    case = inputs.config("F3")
    case["nsteps"] = 77
    d = str(tmp_path / "f3")
    inputs.write_input_files(case, d)
    m, o = compare(d)
    assert o["nstep"] == 77 and o["elec_type"] == 3 and o["n_atoms"] == 4000
    m2 = api.Model.from_case(case)
    for k in ("rmax", "tkin", "scale", "scale2", "vdw", "species", "x", "types"):
        assert np.array_equal(m.query(k), m2.query(k)), k


def test_neutral_species_demote_electrostatics(tmp_path):
    case = inputs.config("F1")
    case.update(elec_type=3, rReal=8.0, alpha=0.4)
    d = str(tmp_path / "neutral")
    inputs.write_input_files(case, d)
    m, o = compare(d)
    assert o["elec_type"] == 0 and o["rmax"] == 6.5          # elec.cpp:52-56 ; sys_init.cpp:1060-1071


def test_bonded_inputs_round_trip(tmp_path):
    """field.txt 'bonds' / 'angles' / 'bond_list' / 'angle_list' + bonds.txt + angles.txt."""
    case = inputs.molecular_case((5, 5, 5), charges=(-0.2, 0.1), elec="fenn")
    d = str(tmp_path / "mol")
    inputs.write_input_files(case, d)
    m, o = compare(d)
    assert len(o["bond_types"]) == 5 and len(o["angle_types"]) == 2 and len(o["bonds"]) == 250 and len(o["angles"]) == 125
    assert o["bond_types"][3]["spec1"] == 1                      # the buck line is written L-C ...
    assert all(o["types"][a1] == o["bond_types"][k - 1]["spec1"] for a1, a2, k in o["bonds"])    # ... and every bond is turned to it
    m2 = api.Model.from_case(case)
    for k in ("bond_types", "angle_types", "bonds", "angles", "n_bonded", "rmax", "degfree"):
        assert np.array_equal(m.query(k), m2.query(k)), k
    assert m.query("degfree")[0] == 3 * 375                      # sim->nBonds is never set: bonds do not reduce degFree (sys_init.cpp:600,1099)


@pytest.mark.parametrize("field_edit,bonds_txt,code", [
    (("\tcon\tcon\n", "\tcon\tbr 2.0 C L\n"), None, "out of scope"),          # breakable bond: use_bnd = 2
    (("\tcon\tcon\n", "\tmut 0.5 2\tcon\n"), None, "out of scope"),
    (("\tharm\t", "\tspring\t"), None, "ERROR[126]"),
    (None, "1\n1 2 1\n", "ERROR [123]"),                                       # ligand-ligand pair for a C-L bond type
    (None, "1\n0 1 9\n", "ERROR[121]"),
    (None, "3\n0 1 1\n", "ERROR[121]"),                                        # truncated list
])
def test_bonded_error_reporting(tmp_path, field_edit, bonds_txt, code):
    case = inputs.molecular_case((3, 3, 3))
    d = str(tmp_path / "molbad")
    inputs.write_input_files(case, d)
    if field_edit:
        text = open(os.path.join(d, "field.txt")).read()
        assert field_edit[0] in text
        open(os.path.join(d, "field.txt"), "w").write(text.replace(field_edit[0], field_edit[1], 1))
    if bonds_txt:
        open(os.path.join(d, "bonds.txt"), "w").write(bonds_txt)
    with pytest.raises(api.AztotError) as ei:
        api.Model.from_dir(d)
    assert code in str(ei.value)


def test_set_bonded_is_all_or_nothing():
    case = inputs.molecular_case((3, 3, 3), rc=5.0)
    m = api.Model.from_case(case)
    before = m.query("bonds").copy()
    with pytest.raises(api.AztotError):
        m.set_bonded(case["bond_types"], case["angle_types"], np.array([[1, 2, 1]]), case["angles"])
    assert np.array_equal(m.query("bonds"), before) and int(m.query("n_bonded")[2]) == 54


@pytest.mark.parametrize("text,code", [("spec 1\nAr Ar 39.9 0 0\nvdw 1\nAr Ar lnjs 4 0.01 3.4\nlinkage 2\n", "out of scope"),
                                       ("vdw 1\nAr Ar lnjs 4 0.01 3.4\n", "ERROR[004]"),
                                       ("spec 1\nAr Ar 39.9 0 0\nvdw 1\nAr Xe lnjs 4 0.01 3.4\n", "ERROR[005]"),
                                       ("spec 1\nAr Ar 39.9 0 0\nvdw 1\nAr Ar morse 4 0.01 3.4\n", "ERROR[006]")])
def test_error_reporting(tmp_path, text, code):
    d = str(tmp_path / "bad")
    write_quirky(d)
    open(os.path.join(d, "field.txt"), "w").write(text)
    with pytest.raises(api.AztotError) as ei:
        api.Model.from_dir(d)
    assert code in str(ei.value)


def test_missing_files_and_scope(tmp_path):
    with pytest.raises(api.AztotError) as ei:
        api.Model.from_dir(str(tmp_path / "nowhere"))
    assert "ERROR[001]" in str(ei.value)
    d = str(tmp_path / "pme")
    write_quirky(d)
    open(os.path.join(d, "control.txt"), "w").write(QUIRKY_CONTROL.replace("elec\tfenn\t8.0\t0.4\t6\t6\t6", "elec\tpme\t8.0\t0.4\t0\t6\t6"))
    m = api.Model.from_dir(d)
    with pytest.raises(api.AztotError) as ei:     # 'elec pme' needs at least one k-vector per axis
        m.query("rmax")
    assert "ERROR[404]" in str(ei.value)


def test_ewald_directive(tmp_path):
    """'elec pme rReal alpha kx ky kz' is the reference's plain Ewald sum (read_elec elec.cpp:33-38, prepare_elec :377-397):
    constants, the k-vector list in ewald_rec's order and the constant energy term against the Python restatement."""
    d = str(tmp_path / "ewald")
    write_quirky(d)
    open(os.path.join(d, "control.txt"), "w").write(QUIRKY_CONTROL.replace("elec\tfenn\t8.0\t0.4\t6\t6\t6", "elec\tpme\t8.0\t0.4\t5\t6\t7"))
    m, o = compare(d)
    assert o["elec_type"] == 2 and o["ewald_k"] == [5, 6, 7] and o["rmax"] == 8.0
    ew = m.query("ewald")
    assert [int(v) for v in ew[:3]] == [5, 6, 7] and int(ew[6]) == len(o["kvecs"]) > 100
    assert ew[5] == pytest.approx(o["eng_elec1"], rel=1e-14)
    kv = m.query("kvecs").reshape(-1, 7)
    assert np.array_equal(kv[:, :3].astype(int), np.array([k[:3] for k in o["kvecs"]]))
    assert np.allclose(kv[:, 3:], np.array([k[3:] for k in o["kvecs"]]), rtol=1e-15, atol=0)
    assert tuple(kv[0, :3]) == (0, 0, 1) and kv[:, 0].min() == 0 and kv[:, 1].min() < 0 and kv[:, 2].min() < 0      # half space: l >= 0


def test_ewald_kvector_on_the_cutoff_sphere(tmp_path):
    """Box 6 x 5 x 4 lattice cells with k = (7, 5, 6): (l, m, n) = (6, 1, 1) has |k|^2 == rkcut^2 in exact arithmetic, so the
    reference's rounding (ip1..3 through prepare_box's matrix algebra, box.cpp:92-151) decides - both parsers must follow it
    (the live-reference oracle test pins the oracle on the same system)."""
    case = inputs.lj_case((6, 5, 4), a=5.4, seed=2, rc=5.2, cell_list=5.2, charges=(0.5, -0.5), elec="fenn", r_real=5.2, alpha=0.5)
    case.update(elec_type=2, ewald_k=(7, 5, 6))
    d = str(tmp_path / "tie")
    inputs.write_input_files(case, d)
    m, o = compare(d)
    kv = m.query("kvecs").reshape(-1, 7)[:, :3].astype(int).tolist()
    assert kv == [list(k[:3]) for k in o["kvecs"]] and len(kv) == int(m.query("ewald")[6])
    from oracle import oracle
    orc = oracle.Oracle(case)
    orc.forces(0)
    e2 = orc.stats()["engElec2"]
    assert abs(e2 - 1.2152744508039863) < 1e-12       # value of the reference binary (1.2048... if the tie falls the other way)


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present on this machine")
@pytest.mark.parametrize("sub", ["case study 1", "case study 2", "src"])
def test_reference_case_studies(sub):
    m, o = compare(os.path.join(REF, sub))
    if sub == "case study 1":
        assert o["n_atoms"] == 40000 and o["elec_type"] == 0 and o["rmax"] == 4.0 and o["cell_list"] == 85.0 and o["tstat_type"] == 2
    else:
        assert o["n_atoms"] == 4000 and o["vdw"][0][0]["type"] == 7 and o["vdw"][0][0]["use_radii"] == 1 and o["rmax"] == 6.0


@pytest.mark.parametrize("k", [1, 2])
def test_case_study_fixtures(k, tmp_path):
    """the reference's shipped example inputs, kept as data in tests/golden/case_study_k.npz: the four files written back from the fixture
    parse identically in the product's C++ parser and in the Python restatement - on any machine, also where the reference tree is absent -
    and (where it is present) are byte-identical to the originals, DOS line ends included."""
    import filecmp
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import materialise_case_study
    d = materialise_case_study(k, str(tmp_path / ("cs%d" % k)))
    m, o = compare(d)
    if k == 1:
        assert o["n_atoms"] == 40000 and o["elec_type"] == 0 and o["rmax"] == 4.0 and o["cell_list"] == 85.0 and o["tstat_type"] == 2
        assert o["nstep"] == 100000 and o["stat"] == 200
    else:
        assert o["n_atoms"] == 4000 and o["vdw"][0][0]["type"] == 7 and o["vdw"][0][0]["use_radii"] == 1 and o["rmax"] == 6.0
        assert o["cell_list"] == 2.7 and o["nequil"] == 10000 and o["eqfreq"] == 2500 and o["species"][0]["radA"] == 2.73
    ref = os.path.join(REF, "case study %d" % k)
    if os.path.isdir(ref):
        for f in ("atoms.xyz", "field.txt", "control.txt", "cuda.txt"):
            assert filecmp.cmp(os.path.join(d, f), os.path.join(ref, f), shallow=False), f
    d2 = materialise_case_study(k, str(tmp_path / "cut"), nstep=20)          # the test harness may shorten the run, nothing else
    assert parse.parse_dir(d2, with_atoms=False)["nstep"] == 20


def test_thermostat_tables_match_oracle():
    from oracle import oracle
    case = inputs.lj_case((3, 3, 3), a=5.26, seed=11, rc=6.5, T=298.0, tstat="radi", radii=[(2.73, 4.731, 0.2)])
    m = api.Model.from_case(case)
    o = oracle.Oracle(case)
    assert np.array_equal(m.query("photons", seed=12345), o.photons())
    uv = m.query("uvects").reshape(3, -1)
    L = oracle.lib()
    import ctypes as C
    ux, uy, uz = np.empty(3072), np.empty(3072), np.empty(3072)
    L.orc_unit_vectors(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in (ux, uy, uz)])
    assert np.array_equal(uv[0], ux) and np.array_equal(uv[1], uy) and np.array_equal(uv[2], uz)
    assert abs((uv ** 2).sum(axis=0) - 1.0).max() < 1e-15 and abs(uv.sum(axis=1)).max() < 1e-12
    ph = o.photons()
    kT = 8.617328270398135e-05 * 298.0
    assert 4.0 * kT < ph.mean() < 6.0 * kT          # Gamma(5, kT) has mean 5 kT (temperature.cpp:28-89)
