"""print a rocprofv3 kernel_stats.csv as a short table: python tests/prof_summary.py <dir-or-csv> [n]"""
import csv, glob, os, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = sorted(glob.glob(os.path.join(p, "**", "*kernel_stats.csv"), recursive=True))[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in list(csv.DictReader(open(p)))[:n]:
    print(f"{r['Name'][:64]:64s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:8.2f} min_us={float(r['MinNs'])/1e3:8.2f} pct={float(r['Percentage']):5.1f}")
