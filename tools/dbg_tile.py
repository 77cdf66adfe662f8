import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from aztotmd_amd import api, inputs
from oracle import oracle
from util import rel_err
for name, kw in (("F1", dict(cell_size=2.2)), ("F1", {}), ("C2", {})):
    case = inputs.config(name)
    o = oracle.Oracle(case); o.forces(1); so = o.state()
    for v in (1, 2):
        e = api.Engine(api.Model.from_case(case), pair_variant=v, **kw)
        s = e.state()
        bad = np.where(np.abs(s["fx"] - so["fx"]) > 1e-9)[0]
        print(name, kw, "variant", v, "relerr", rel_err(s["fx"], so["fx"]), "nbad", len(bad), bad[:8], e.stats()["engVdW"], o.stats()["engVdW"])
