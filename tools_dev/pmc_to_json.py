"""profiles/rNN_pmc_fetch_size_<workload>.json from a rocprofv3 --pmc FETCH_SIZE run of bench.py:
    python tools_dev/pmc_to_json.py <dir> "<command line that was profiled>" <bench json of THAT run> > profiles/r05_pmc_fetch_size_70b.json
bytes = FETCH_SIZE (KiB) * 1024 * 2 (gfx950 correction of MI355X_MICROARCH.md, HBM/rocprofv3 section).
Per kernel family (bench.py: roofline.by_family) the file holds the bytes fetched per launch AND the algorithmic bytes per launch of the very same run (the profiled bench's own
launch accounting), i.e. bytes fetched per algorithmic byte under the pass widths that schedule really uses.  It carries the sha of the kernel sources it was measured on
(bench.kernel_source_sha): bench.py quotes it only when that matches."""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

TYPES = {"2": "Q4_0", "3": "Q4_1", "6": "Q5_0", "7": "Q5_1", "8": "Q8_0", "10": "Q2_K", "11": "Q3_K", "12": "Q4_K", "13": "Q5_K", "14": "Q6_K"}
d, source = sys.argv[1], sys.argv[2]
bench_json = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1]) if len(sys.argv) > 3 else None          # (the bench's DETAIL file: the full result)
raw = collections.defaultdict(list)
if os.path.isfile(d) and d.endswith(".json"):                                        # re-derive the family tables of an existing file from its by_symbol section
    for name, v in json.load(open(d))["by_symbol"].items():
        raw[name] = [float(v["avg_fetch_bytes_per_launch"])] * int(v["launches"])
for f in ([] if os.path.isfile(d) else sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            raw[re.sub(r"\(.*", "", r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024 * 2)


def family(name):
    """bench.py's family of a kernel symbol, and a finer key (type, width class) for the mat kernels"""
    m = re.match(r"void k_mmq_kq<(\d+), (\d+), (\d+), (\d+), (true|false), (\d+)", name)
    if m:
        width = "nt33-64" if m.group(6) == "4" else "nt17-32" if m.group(6) == "2" else "nt3-8" if m.group(5) == "true" else "nt9-16"
        return "matrix_cores", f"{TYPES.get(m.group(1), m.group(1))}_matrix_cores_{width}"
    m = re.match(r"void k_mmq_kq2t<(\d+), (\d+), ", name)                           # two weight types in one matrix-core launch (q + k + v): counted with the first type
    if m:
        return "matrix_cores", f"{TYPES.get(m.group(1), m.group(1))}+{TYPES.get(m.group(2), m.group(2))}_matrix_cores_two_types"
    m = re.match(r"void k_mmq_g2<(\d+), (\d+)>", name)                               # two row groups per wave: Q4_K, <= 8 tokens
    if m:
        return "matrix_cores", "Q4_K_matrix_cores_nt3-8"
    m = re.match(r"void k_mmvq_kq2<(\d+), (\d+), (\d+), ", name)                      # two weight types in one launch: counted with the first type, like bench.py's accounting
    if m:
        return ("matvec_nt1" if m.group(3) == "1" else "matvec_multi"), f"{TYPES.get(m.group(1), m.group(1))}_nt{m.group(3)}"
    m = re.match(r"void k_mmvq_(?:kq|q0r|q0|leg)<(\d+), (\d+)", name)
    if m:
        return ("matvec_nt1" if m.group(2) == "1" else "matvec_multi"), f"{TYPES.get(m.group(1), m.group(1))}_nt{m.group(2)}"
    for sym, fam in (("k_norm_quant", "norm_quant"), ("k_quantize_q8", "quantize"), ("k_act_quant", "act_quant"), ("k_attention", "attention")):
        if sym in name:
            return fam, None
    if any(x in name for x in ("k_get_rows", "k_rope_kv", "k_topk", "k_norm", "k_add", "k_silu", "k_gelu", "k_clamp", "k_persimmon")):
        return "other", None
    return None, None


# Launches of a cancelled run return on entry (the abort word) and fetch next to nothing — but the profiled process's accounting prices every launch with its full algorithmic
# bytes, so a pass with many of them reads as "fetched LESS than the algorithmic bytes" (round 5's 0.998 and this round's first pass, 1.0001, were such averages: round 6's
# second pass happened to hold no aborted launch and shows the real ratio, 1.03 for the Q4_K tiles and 1.11 for Q6_K: the tile-major copies are that much larger than the GGUF
# blocks).  So: per symbol, launches below a third of the symbol's median are set apart (`launches_aborted`), and the ratio is taken over the launches that did the work.
fam, keyed, fam_work = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(list)
for name, v in raw.items():
    f, k = family(name)
    med = sorted(v)[len(v) // 2]
    work = [x for x in v if x >= med / 3.0] if f in ("matrix_cores", "matvec_nt1", "matvec_multi") else v
    if f:
        fam[f] += v; fam_work[f] += work
    if k:
        keyed[k] += v
families = {}
for f, v in sorted(fam.items()):
    e = {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))}
    w = fam_work[f]
    e["launches_aborted"] = len(v) - len(w)
    e["avg_fetch_bytes_per_launch_that_did_the_work"] = int(sum(w) / max(1, len(w)))
    bf = (bench_json or {}).get("roofline", {}).get("process_eager_launches", {}).get(f)
    if bf and bf.get("alg_bytes_per_launch"):
        # the profiled process's own accounting of the SAME launches (PI_NO_GRAPHS=1: every launch is eager and counted from the first one on, both contexts)
        e["alg_bytes_per_launch_same_run"] = bf["alg_bytes_per_launch"]; e["launches_accounted_same_run"] = bf["launches"]
        e["fetch_over_alg_all_launches"] = round(e["avg_fetch_bytes_per_launch"] / bf["alg_bytes_per_launch"], 4)
        e["fetch_over_alg"] = round(e["avg_fetch_bytes_per_launch_that_did_the_work"] / bf["alg_bytes_per_launch"], 4)      # (the family's average algorithmic bytes per launch: aborted launches are a random sample of its mix)
    families[f] = e
out = {"source": source,
       "kernel_src_sha16": B.kernel_source_sha(),
       "correction": "bytes = FETCH_SIZE (KiB) * 1024 * 2 — gfx950 tallies 128-byte requests of a wide coalesced read at 64 B (MI355X_MICROARCH.md, HBM/rocprofv3 section)",
       "families": families,
       "kernels": {k: {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))} for k, v in sorted(keyed.items())},
       "by_symbol": {k: {"launches": len(v), "avg_fetch_bytes_per_launch": int(sum(v) / len(v))} for k, v in sorted(raw.items())}}
print(json.dumps(out, indent=1))
