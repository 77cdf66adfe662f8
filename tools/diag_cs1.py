"""diagnostic: step the HIP engine and the oracle side by side on case study 1 and report where they part"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from aztotmd_amd import api
from oracle import oracle, parse
from util import materialise_case_study, case_from_parsed
d = materialise_case_study(1, "/tmp/cs1_diag", nstep=20)
case = case_from_parsed(parse.parse_dir(d))
e = api.Engine(api.Model.from_dir(d))
o = oracle.Oracle(case); o.forces(1)
for step in range(1, 21):
    e.step(1); o.step(1)
    s, so = e.state(), o.state()
    worst = 0
    for k, ko in (("vx", "vx"), ("vy", "vy"), ("vz", "vz"), ("U", "U"), ("radius", "rad")):
        err = np.abs(s[k] - so[ko]); i = int(err.argmax())
        print("step %2d %-6s max abs err %.3e at atom %d: hip %.17g oracle %.17g ; n(err > 1e-12 * max) = %d" % (step, k, err[i], i, s[k][i], so[ko][i], int((err > 1e-12 * np.abs(so[ko]).max()).sum())))
        worst = max(worst, err[i] / np.abs(so[ko]).max())
    if worst > 1e-10:
        i = int(np.abs(s["vx"] - so["vx"]).argmax())
        for k, ko in (("x", "x"), ("vx", "vx"), ("vy", "vy"), ("vz", "vz"), ("U", "U"), ("radius", "rad"), ("fx", "fx")):
            print("   atom %d %-6s hip %.17g oracle %.17g" % (i, k, s[k][i], so[ko][i]))
        break
