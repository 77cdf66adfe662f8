"""README.md's table from the bench lines of tools/r04_numbers.sh (profiles/r04_*_bench_line.json): python tools/r04_readme_table.py"""
import json, os
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
ROWS = [("C4", "C4: Ar, LJ rc 8.5 Å, lattice at rest (default `--steps 100 --warmup 10`)"), ("C4_driver", "C4, the driver's `--steps 20 --warmup 5` (fresh process)"),
        ("C4_long", "C4, steady state (`--steps 1000 --warmup 1000`)"), ("C4_every", "C4 with the reference's schedule (`--sort-every 1`: cells rebuilt every step)"),
        ("C4T", "**C4T: the same at 85 K** (Maxwell velocities; 44–49 K after equipartition)"), ("C4X", "C4X: the lattice in a box of exactly 42 × 8.5 Å"),
        ("C3", "C3: LJ + Fennell/DSF Coulomb"), ("C3T", "C3T: the same at 85 K"), ("B3", "B3: Born–Mayer–Huggins + Fennell (heats up)"),
        ("M4", "M4: bonded triatomics (bonds + angles), LJ + Fennell"), ("C2", "C2: Ar, LJ"), ("C2T", "C2T: the same at 85 K"),
        ("C1", "C1: dilute gas, radiative thermostat"), ("CS1", "**case study 1: the reference's shipped input, verbatim**"), ("CS2", "**case study 2, verbatim**"),
        ("S40", "S40: `surk` + radii on 2.7 Å cells, radiative thermostat"), ("S4", "S4: the same at case study 2's size"), ("E2", "E2: LJ + full Ewald sum (4 231 k-vectors)")]
print("| workload | atoms | ms/step | ns/day | cells rebuilt every | pair kernel (µs) | `aztot_step(1)` loop (ms/step) |")
print("|---|---|---|---|---|---|---|")
for key, label in ROWS:
    f = os.path.join(P, "r04_%s_bench_line.json" % key)
    if not os.path.exists(f):
        continue
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = d.get("kernels") or {}
    pk = d.get("roofline", {}).get("kernel")
    co = d.get("call_overhead") or {}
    print("| %s | %s | %.4g | %.0f | %s | %s | %s |" % (label, format(d["config"]["n_atoms"], ",").replace(",", " "), d["ms_per_step"], d["value"], d["config"].get("sort_interval"),
                                                  ("%.0f `k_%s`" % (k[pk]["avg_us"], pk)) if pk in k else "-", ("%.4g" % co["step1_ms_per_step"]) if co else "-"))
