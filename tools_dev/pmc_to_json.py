"""profiles/r01_pmc_fetch_size.json from a rocprofv3 --pmc FETCH_SIZE run: python tools_dev/pmc_to_json.py <dir> "<command line that was profiled>" > out.json
bytes = FETCH_SIZE (KiB) * 1024 * 2 (gfx950 correction of MI355X_MICROARCH.md, HBM/rocprofv3 section)"""
import csv, glob, json, os, re, sys, collections
d, source = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            acc[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024 * 2)
kernels = {k: {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))} for k, v in sorted(acc.items())}
dom = [x for k, v in acc.items() if re.match(r"void k_mmvq_kq<12, 1, ", k) for x in v]      # Q4_K (type 12), NT = 1: all three instances
out = {"source": source,
       "correction": "bytes = FETCH_SIZE (KiB) * 1024 * 2 — gfx950 tallies 128-byte requests of a wide coalesced read at 64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section)",
       "dominant_kernel": {"name": "k_mmvq_kq<Q4_K,NT=1>", "launches": len(dom), "avg_fetch_bytes_per_launch": int(sum(dom) / max(1, len(dom)))},
       "kernels": kernels}
print(json.dumps(out, indent=1))
