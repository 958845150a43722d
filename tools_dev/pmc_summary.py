"""average PMC counters per kernel from a rocprofv3 --pmc run: python tests/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys, collections
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} n={len(v):5d} mean={sum(v)/len(v):14.1f}")
