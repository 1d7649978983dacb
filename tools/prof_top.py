"""Print the top kernels of a rocprofv3 --stats CSV directory (argument: directory)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:72]:72s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:8.1f} min={float(r['MinNs'])/1e3:8.1f} max={float(r['MaxNs'])/1e3:8.1f} pct={r['Percentage']}")
