"""Tuning aid: from a rocprofv3 kernel trace of bench.py, the launches of one replay of the gradient-penalty term (from
its forward kernel to the head's weight gradient, same queue): start offset, duration, name.
    python3 tools/gp_trace.py OUTDIR [which_replay]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
starts = [i for i, r in enumerate(rows) if 'k_gpt_fwd' in r['Kernel_Name'] or 'k_gp_enc' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(starts) // 2
i0 = starts[which]; q = rows[i0].get('Queue_Id'); t0 = rows[i0]['s']
print("replays:", len(starts), "showing", which, "queue", q)
last = t0
for r in rows[i0:i0 + 400]:
    if r.get('Queue_Id') != q: continue
    name = r['Kernel_Name'].replace('aig::(anonymous namespace)::', '').replace('void ', '')[:64]
    print(f"  t={(r['s'] - t0) / 1e3:7.1f} us  gap {(r['s'] - last) / 1e3:5.1f}  dur {(r['e'] - r['s']) / 1e3:6.1f}  {name}")
    last = r['e']
    if 'k_head_gr_wgrad' in r['Kernel_Name']: break
