"""Tuning aid: medians of one kernel's durations from a rocprofv3 kernel trace, for the last GROUPS groups of
PER consecutive launches (a timing script that runs PER launches per problem size, sizes in order).
    python3 tools/prof_by_order.py OUTDIR KERNEL_SUBSTRING PER GROUPS"""
import csv, glob, sys
out, name, per, groups = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if name in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-per * groups:]
for gi in range(groups):
    v = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[gi * per:(gi + 1) * per])
    print(f"group {gi}: {len(v)} launches, median {v[len(v) // 2]:.1f} us, min {v[0]:.1f} us")
