import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from aztotmd_amd import api, inputs
from oracle import oracle
from util import rel_err
pos, box = inputs.fcc_positions((6, 6, 6), 5.8, 0.1, 5)
N = len(pos)
case = {"box": box.tolist(), "dt": 0.001, "species": [(39.9, 0.0)], "names": ["Ar"], "types": np.zeros(N, dtype=np.int32),
        "vdw": [(0, 0, 7, 6.0, [75.0, 8.0, 1.0, 1.0])], "radii": [(2.73, 4.731, 0.2)], "x": pos[:, 0].copy(), "y": pos[:, 1].copy(),
        "z": pos[:, 2].copy(), "vx": np.zeros(N), "vy": np.zeros(N), "vz": np.zeros(N), "T": 500.0, "tstat_type": 2,
        "cell_list": 2.7, "use_clist": 1, "elec_type": 0}
o = oracle.Oracle(case); o.forces(0)
e = api.Engine(api.Model.from_case(case), use_graph=int(sys.argv[1]) if len(sys.argv)>1 else 1)
for st in range(6):
    s, so = e.state(), o.state()
    print(st, {a: "%.2e" % rel_err(s[a], so[b]) for a, b in (("x","x"),("vx","vx"),("fx","fx"),("U","U"),("radius","rad"))}, e.stats()["engVdW"], o.stats()["engVdW"], np.abs(so["fx"]).max())
    e.step(1); o.step(1)
