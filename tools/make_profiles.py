"""Copy the rocprofv3 summaries that back bench.py's roofline numbers from gpurun_out/ (scratch) into profiles/.

  python tools/make_profiles.py <round-tag> <stats_dir> <fetch_dir> <write_dir> [workload] [n_gpus]
HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are collected in SEPARATE --pmc
passes (TCC slot limit), are in KiB, and on gfx950 FETCH_SIZE reports exactly half of the bytes actually fetched
(checked here on k_integrate2, whose traffic is known: 52 B/atom read, 24 B/atom written).
"""
import csv, glob, json, os, shutil, sys, collections

tag, stats_dir, fetch_dir, write_dir = sys.argv[1:5]
workload = sys.argv[5] if len(sys.argv) > 5 else "C4"
ngpu = int(sys.argv[6]) if len(sys.argv) > 6 else 1
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
os.makedirs(P, exist_ok=True)
for f in glob.glob(stats_dir + "/**/*kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(P, "%s_rocprofv3_kernel_stats.csv" % tag))
for f in glob.glob(stats_dir + "/bench_line.json"):
    shutil.copy(f, os.path.join(P, "%s_bench_line_under_rocprofv3.json" % tag))


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                k = row["Kernel_Name"].split("(")[0].replace("void aztot::", "").replace("aztot::", "")
                acc[k].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    rows.append({"kernel": k, "dispatches": max(nf, nw), "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
                 "hbm_read_bytes": 2.0 * f * 1024.0, "hbm_write_bytes": w * 1024.0, "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0})
with open(os.path.join(P, "%s_rocprofv3_pmc_hbm_traffic.csv" % tag), "w", newline="") as f:
    wri = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    wri.writeheader()
    wri.writerows(rows)
# FP64 work per launch from the SQ instruction counters (optional 5th pass of tools/profile_run.sh): wave-instructions x 64 lanes,
# FMA counted twice; MFMA_MOPS in units of 512 FLOP (counter_defs.yaml: TOTAL_64_OPS)
fp64_dir = os.path.join(os.path.dirname(fetch_dir.rstrip("/")), "fp64")
flop = {}
if os.path.isdir(fp64_dir):
    c = {n: means(fp64_dir, "SQ_INSTS_VALU_%s_F64" % n) for n in ("FMA", "ADD", "MUL", "TRANS", "MFMA_MOPS")}
    c["MFMA_MOPS_F32"] = means(fp64_dir, "SQ_INSTS_VALU_MFMA_MOPS_F32")
    for k in set().union(*[set(v) for v in c.values()]):
        g = lambda n: c[n].get(k, (0.0, 0))[0]
        flop[k] = (2.0 * g("FMA") + g("ADD") + g("MUL") + g("TRANS")) * 64.0 + 512.0 * g("MFMA_MOPS")
    for f in ("sq_summary.txt", "fp64_summary.txt"):
        src = os.path.join(os.path.dirname(fp64_dir), f)
        if os.path.exists(src):
            shutil.copy(src, os.path.join(P, "%s_rocprofv3_pmc_%s" % (tag, f.replace("_summary", "_counters"))))
tp = os.path.join(P, "pmc_traffic.json")
rec = json.load(open(tp)) if os.path.exists(tp) else {}
# which build the counters were collected on: the bench line printed under rocprofv3 carries aztot_version (with the digest of the library's sources);
# bench.py replays an entry only on that very build
libs = set()
for d in (fetch_dir, write_dir, fp64_dir):
    try:
        libs.add(json.loads(open(os.path.join(d, "bench_line.json")).read().strip().splitlines()[-1])["library"])
    except Exception:
        pass
library = libs.pop() if len(libs) == 1 else None
seen = {}        # several instantiations share a bench name (k_pair_list<.., ENG = true / false>): the one launched most often carries the run
for r in rows:
    k = r["kernel"]
    # bench.py's names: k_pair_list -> pair_list ; k_pair_tile<MODE, VDW, CLEANUP, BUILD>: BUILD -> build_lists, CLEANUP -> pair_cleanup, else pair_tile
    if k.startswith("k_pair_list<"):
        name = "pair_list"
    elif k.startswith("k_build_lists"):
        name = "build_lists"
    elif k.startswith("k_pair_tile<"):
        a = [x.strip() for x in k[k.index("<") + 1:k.rindex(">")].split(",")]
        name = "pair_cleanup" if (len(a) > 2 and a[2] == "true") else "pair_tile"
    else:
        name = "pair_atom" if k == "k_pair_atom" else None
    if name and r["dispatches"] > seen.get(name, 0):
        seen[name] = r["dispatches"]
        rec["%s:%s:%d" % (workload, name, ngpu)] = {"hbm_bytes_per_launch": r["hbm_bytes_per_launch"], "fp64_flop_per_launch": flop.get(r["kernel"]), "round": tag, "kernel": r["kernel"], "library": library,
                                                    "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH doubled per MI355X_MICROARCH.md"}
json.dump(rec, open(tp, "w"), indent=1)
print(open(os.path.join(P, "%s_rocprofv3_pmc_hbm_traffic.csv" % tag)).read())
