"""Per-kernel summary of rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE, one directory per pass).

    python tools/pmc_summary.py DIR [DIR ...]                      CSV on stdout: kernel, counter, mean / min / max KB
    python tools/pmc_summary.py --json real=DIR,DIR fake=DIR,DIR   JSON: HBM-side bytes per launch by graph kind

The JSON is what bench.py reports as `traffic` (profiles/pmc_traffic.json): FETCH_SIZE doubled + WRITE_SIZE."""
import csv, glob, json, re, sys
from collections import defaultdict

KERNELS = ["k_gat_aggregate", "k_gat_bwd_dst", "k_gat_bwd_src", "k_pool_walk", "k_linear128", "k_wgrad128", "k_gat1_fwd",
           "k_gat1_bwd"]
ALIAS = {"k_linear128_x6": "k_linear128", "k_wgrad128_x6": "k_wgrad128"}      # the kernels behind aig_linear128 / aig_wgrad


def read(dirs):
    vals = defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                name = re.sub(r"^(void )?(aig::\(anonymous namespace\)::)?", "", r["Kernel_Name"])
                name = re.sub(r"[(<].*", "", name)
                name = ALIAS.get(name, name)
                vals[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    return vals


if "--json" in sys.argv:
    out = {"note": "bytes that left the L2 towards memory per launch, from rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE "
                   "(separate passes over tools/perf_critic.py: the critic's forward + backward pass on a synth100k graph - "
                   "'real' - and on what the untrained generator wires for it, repaired - 'fake'); FETCH_SIZE doubled, as "
                   "gfx950 tallies 128-B read requests at 64 B (calibration in the same pass: k_wgrad128 reads 2 x 51.2 MB "
                   "and reports about half) - an upper bound for kernels whose reads are narrower",
           "synth100k": {"by_kind": {}}}
    for spec in [a for a in sys.argv[1:] if "=" in a]:
        kind, dirs = spec.split("=")
        vals = read(dirs.split(","))
        rows = {}
        for k in KERNELS:
            f, w = vals.get((k, "FETCH_SIZE")), vals.get((k, "WRITE_SIZE"))
            if not f or not w:
                continue
            fm, wm = sum(f) / len(f), sum(w) / len(w)
            rows[k] = {"traffic_bytes_per_launch": int((2 * fm + wm) * 1024), "fetch_size_kb_mean": round(fm, 1),
                       "write_size_kb_mean": round(wm, 1), "dispatches": len(f)}
        out["synth100k"]["by_kind"][kind] = rows
    print(json.dumps(out, indent=1))
else:
    vals = read([a for a in sys.argv[1:]])
    print("kernel,counter,mean_KB,min_KB,max_KB,dispatches")
    for (k, c), v in sorted(vals.items()):
        print(f"{k},{c},{sum(v) / len(v):.1f},{min(v):.1f},{max(v):.1f},{len(v)}")
