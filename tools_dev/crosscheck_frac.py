"""roofline.frac of the bench line against the committed rocprofv3 summaries: sum over the matrix-core kernels of (calls x PMC bytes per launch of the symbol) / sum of their durations
    python3 tools_dev/crosscheck_frac.py [profiles/r06_bench_70b_default_eager_kernel_stats.csv] [profiles/r06_pmc_fetch_size_70b.json] [profiles/r06_bench_70b_n1.json]"""
import csv, json, re, sys
stats = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06_bench_70b_default_eager_kernel_stats.csv"
pmc_all = json.load(open(sys.argv[2] if len(sys.argv) > 2 else "profiles/r06_pmc_fetch_size_70b.json"))
pmc = pmc_all["by_symbol"]
line = json.load(open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r06_bench_70b_n1.json"))
tot_b = tot_t = 0
for r in csv.DictReader(open(stats)):
    name = re.sub(r"\(.*", "", r["Name"])
    if "k_mmq" in name and name in pmc:
        tot_b += int(r["Calls"]) * pmc[name]["avg_fetch_bytes_per_launch"]; tot_t += int(r["TotalDurationNs"])
ratio = pmc_all["families"]["matrix_cores"].get("fetch_over_alg") or 1.0
print(f"all k_mmq_* (kernel-trace durations x PMC bytes per launch): {tot_b / tot_t:.0f} GB/s FETCHED = {tot_b / tot_t / 8000:.3f} of the HBM peak; "
      f"over the algorithmic bytes (fetched / {ratio}: the tile-major copies are larger than the GGUF blocks): {tot_b / tot_t / ratio:.0f} GB/s = {tot_b / tot_t / ratio / 8000:.3f}; "
      f"the bench line (algorithmic bytes): frac {line['roofline']['frac']} ({line['roofline']['family']})")
