"""Per-step timeline from a rocprofv3 --kernel-trace CSV: mean duration of every kernel and mean idle gap in front of it.
    python tools/timeline.py <dir with *_kernel_trace.csv> [skip_first_n_dispatches]"""
import csv, glob, sys, collections
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void aztot::", "").replace("aztot::", "")))
rows.sort()
rows = rows[skip:]
dur, gap, cnt = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(int)
for i in range(1, len(rows)):
    s, e, n = rows[i]
    dur[n] += e - s
    gap[n] += max(0, s - rows[i - 1][1])
    cnt[n] += 1
span = rows[-1][1] - rows[0][0]
print("dispatches %d, span %.1f us" % (len(rows), span / 1e3))
tot_d = tot_g = 0
for n in sorted(cnt, key=lambda k: -dur[k]):
    print("%-50s n=%5d  dur %7.2f us  gap-before %6.2f us" % (n[:50], cnt[n], dur[n] / cnt[n] / 1e3, gap[n] / cnt[n] / 1e3))
    tot_d += dur[n]; tot_g += gap[n]
print("busy %.1f %%, idle between kernels %.1f %%" % (100.0 * tot_d / span, 100.0 * tot_g / span))
