"""profiles/r03_pmc_fetch_size_<workload>.json from a rocprofv3 --pmc FETCH_SIZE run:
    python tools_dev/pmc_to_json.py <dir> "<command line that was profiled>" > profiles/r03_pmc_fetch_size_70b.json
bytes = FETCH_SIZE (KiB) * 1024 * 2 (gfx950 correction of MI355X_MICROARCH.md, HBM/rocprofv3 section).
The file carries the sha of the kernel sources it was measured on (bench.kernel_source_sha): bench.py quotes it only when that matches."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

TYPES = {"2": "Q4_0", "3": "Q4_1", "6": "Q5_0", "7": "Q5_1", "8": "Q8_0", "12": "Q4_K", "13": "Q5_K", "14": "Q6_K"}
d, source = sys.argv[1], sys.argv[2]
raw = collections.defaultdict(list)
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            raw[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024 * 2)
keyed = collections.defaultdict(list)
for name, v in raw.items():
    m = re.match(r"void k_mmvq_kq<(\d+), (\d+), ", name)
    if m and m.group(1) in TYPES:
        keyed[f"{TYPES[m.group(1)]}_nt{m.group(2)}"] += v              # all G / epilogue instances of one (type, tokens per pass)
        continue
    m = re.match(r"void k_mmvq_kq2<(\d+), (\d+), (\d+), ", name)     # two weight types in one launch: counted with the first type (Q4_K), like bench.py's launch statistics
    if m and m.group(1) in TYPES:
        keyed[f"{TYPES[m.group(1)]}_nt{m.group(3)}"] += v
        continue
    m = re.match(r"void k_mmq_kq<(\d+), ", name)
    if m and m.group(1) in TYPES:
        keyed[f"{TYPES[m.group(1)]}_matrix_cores"] += v
        continue
    m = re.match(r"void k_mmvq_q0r<(\d+), (\d+), ", name)
    if m and m.group(1) in TYPES:
        keyed[f"{TYPES[m.group(1)]}_nt{m.group(2)}"] += v
kernels = {k: {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))} for k, v in sorted(keyed.items())}
out = {"source": source,
       "kernel_src_sha16": B.kernel_source_sha(),
       "correction": "bytes = FETCH_SIZE (KiB) * 1024 * 2 — gfx950 tallies 128-byte requests of a wide coalesced read at 64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section)",
       "kernels": kernels,
       "by_symbol": {k: {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))} for k, v in sorted(raw.items())}}
print(json.dumps(out, indent=1))
