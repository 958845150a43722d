"""per (kernel, grid) durations from a rocprofv3 kernel_trace.csv: python tests/prof_trace.py <dir-or-csv> [filter]"""
import csv, glob, os, sys, collections
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True))[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(p)):
    if flt in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:48], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(d):
    v = sorted(d[k])
    print(f"{k[0]:48s} wgs={k[1]:6d} n={len(v):5d} med={v[len(v)//2]:7.2f} min={v[0]:7.2f} p90={v[int(len(v)*0.9)]:7.2f}")
