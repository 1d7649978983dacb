"""Per-kernel summary of rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one directory per pass) as CSV on stdout,
and, with --json, the k_gat_aggregate row as HBM bytes per launch (the figure bench.py reports as roofline.traffic).

    python tools/pmc_summary.py gpurun_out/r01_pmc_fetch gpurun_out/r01_pmc_write > profiles/r01_pmc_fetch_write_perf_gat.csv
    python tools/pmc_summary.py --json gpurun_out/r01_pmc_fetch gpurun_out/r01_pmc_write > profiles/pmc_traffic.json
"""
import csv, glob, json, re, sys
from collections import defaultdict

args = [a for a in sys.argv[1:] if a != "--json"]
vals = defaultdict(list)
for d in args:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"^(void )?(aig::\(anonymous namespace\)::)?", "", r["Kernel_Name"])
            name = re.sub(r"[(<].*", "", name)
            vals[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
if "--json" in sys.argv:
    f, w = vals[("k_gat_aggregate", "FETCH_SIZE")], vals[("k_gat_aggregate", "WRITE_SIZE")]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    print(json.dumps({
        "note": "HBM-side bytes per launch from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes over "
                "tools/perf_gat.py, the synth100k real and repaired-fake graphs); FETCH_SIZE doubled as gfx950 tallies "
                "128-B read requests at 64 B (calibration in the same pass: k_wgrad128 reads 2 x 51.2 MB and reports about half)",
        "synth100k": {"k_gat_aggregate": {"traffic_bytes_per_launch": int((2 * fm + wm) * 1024),
                                          "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1),
                                          "dispatches": len(f)}}}, indent=1))
else:
    print("kernel,counter,mean_KB,min_KB,max_KB,dispatches")
    for (k, c), v in sorted(vals.items()):
        print(f"{k},{c},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f},{len(v)}")
