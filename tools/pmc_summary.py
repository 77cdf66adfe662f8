"""Summarise rocprofv3 --pmc CSV output: per kernel name, mean of each counter over dispatches."""
import csv, glob, sys, collections
d = sys.argv[1]
files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        k = row.get("Kernel_Name", "?")
        short = k.split("(")[0].replace("void aztot::", "").replace("aztot::", "")
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, c in sorted(acc.items()):
    if filt and filt not in k:
        continue
    print(k[:70])
    for name, vals in sorted(c.items()):
        print("    %-28s n=%-4d mean=%.4g" % (name, len(vals), sum(vals) / len(vals)))
