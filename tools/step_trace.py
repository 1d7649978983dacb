"""Tuning aid: from a rocprofv3 kernel trace of `bench.py --no-kernel-timing ...`, how busy the GPU is inside the
training steps: per step (delimited by the generator's Adam launch, the last kernel of a step) the wall time,
the union of all kernel intervals (GPU busy with anything), the union of the "wide" kernels (>= 64 workgroups)
and the sum of their durations.
    python3 tools/step_trace.py OUTDIR [first_step last_step]"""
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
    gx = int(r.get('Grid_Size_X', r.get('Grid_Size', 0)) or 0) * max(int(r.get('Grid_Size_Y', 1) or 1), 1)
    wx = max(int(r.get('Workgroup_Size_X', r.get('Workgroup_Size', 1)) or 1), 1) * max(int(r.get('Workgroup_Size_Y', 1) or 1), 1)
    r['wgs'] = gx // wx
rows.sort(key=lambda r: r['s'])
adam = [r for r in rows if 'k_adam_sum3' in r['Kernel_Name'] or 'multi_tensor_apply' in r['Kernel_Name'] and 'Adam' in r['Kernel_Name'] or 'FusedAdam' in r['Kernel_Name'] or 'fused_adam' in r['Kernel_Name'].lower()]
print("adam-like launches:", len(adam), "of", len(rows), "kernels")
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot
# D has 3 Adam launches per step, G one more (no gradient: may be skipped); group by gaps instead: steps are
# delimited by every 3rd D-Adam launch
bounds = [a_['e'] for a_ in adam]
per = int(sys.argv[4]) if len(sys.argv) > 4 else 3
steps = [(bounds[i], bounds[i + per]) for i in range(0, len(bounds) - per, per)]
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else len(steps)
for si, (s0, s1) in enumerate(steps[lo:hi], lo):
    ks = [r for r in rows if r['s'] >= s0 and r['e'] <= s1]
    wide = [r for r in ks if r['wgs'] >= 64]
    print(f"step {si}: wall {(s1 - s0) / 1e6:.3f} ms, kernels {len(ks)}, busy(any) {union([(r['s'], r['e']) for r in ks]) / 1e6:.3f}, "
          f"busy(wide) {union([(r['s'], r['e']) for r in wide]) / 1e6:.3f}, sum(wide) {sum(r['e'] - r['s'] for r in wide) / 1e6:.3f}, "
          f"sum(narrow) {sum(r['e'] - r['s'] for r in ks if r['wgs'] < 64) / 1e6:.3f}")
if len(sys.argv) > 5:
    si = int(sys.argv[5]); s0, s1 = steps[si]
    ks = [r for r in rows if r['s'] >= s0 and r['e'] <= s1]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in ks:
        agg[r['Kernel_Name'][:70]][0] += 1; agg[r['Kernel_Name'][:70]][1] += r['e'] - r['s']
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"   {k:70s} {v[0]:4d} {v[1] / 1e3:9.1f} us")
if len(sys.argv) > 6:   # compressed timeline of that step: wide kernels by name, runs of narrow ones collapsed
    cols = rows[0].keys()
    qk = 'Queue_Id' if 'Queue_Id' in cols else None
    cur = None
    def flush():
        global cur
        if cur: print(f"   t={cur[0] / 1e3:8.1f} us  q{cur[4]}  [{cur[2]} narrow launches, {cur[3] / 1e3:.1f} us busy, span {(cur[1] - cur[5]) / 1e3:.1f} us]  " + " ".join(cur[6]))
        cur = None
    for r in ks:
        q = r[qk] if qk else '?'
        if r['wgs'] >= 64 and r['e'] - r['s'] > 8000:
            flush()
            print(f"   t={(r['s'] - s0) / 1e3:8.1f} us  q{q}  {(r['e'] - r['s']) / 1e3:7.1f} us  {r['Kernel_Name'].replace('aig::(anonymous namespace)::', '').replace('void ', '')[:60]}")
        else:
            nm = r['Kernel_Name'].replace('aig::(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0][-24:] + f":{(r['e'] - r['s']) / 1e3:.0f}"
            if cur and cur[4] == q: cur[1] = r['e']; cur[2] += 1; cur[3] += r['e'] - r['s']; cur[6].append(nm)
            else:
                flush(); cur = [r['s'] - s0, r['e'], 1, r['e'] - r['s'], q, r['s'], [nm]]
    flush()
if len(sys.argv) > 7:   # wide-kernel busy fraction per window of W us, with the widest kernels of each window
    W = float(sys.argv[7]) * 1e3
    t = s0
    while t < s1:
        win = [(max(r['s'], t), min(r['e'], t + W), r) for r in ks if r['e'] > t and r['s'] < t + W]
        wide_iv = [(a_, b_) for a_, b_, r in win if r['wgs'] >= 64]
        names = collections.Counter()
        for a_, b_, r in win:
            names[r['Kernel_Name'].replace('aig::(anonymous namespace)::', '').replace('void ', '').split('(')[0][:28]] += b_ - a_
        top = ", ".join(f"{k}:{v / 1e3:.0f}" for k, v in names.most_common(4))
        print(f"  w t={(t - s0) / 1e3:7.0f} us  wide-busy {union(wide_iv) / W * 100:5.1f} %   {top}")
        t += W
if len(sys.argv) > 9:   # every kernel between two offsets (us) of that step
    lo_t, hi_t = float(sys.argv[8]) * 1e3 + s0, float(sys.argv[9]) * 1e3 + s0
    qk = 'Queue_Id' if 'Queue_Id' in rows[0] else None
    for r in ks:
        if r['e'] >= lo_t and r['s'] <= hi_t:
            print(f"  d t={(r['s'] - s0) / 1e3:8.1f} +{(r['e'] - r['s']) / 1e3:6.1f} us q{r[qk] if qk else '?'} wgs={r['wgs']:5d} {r['Kernel_Name'].replace('aig::(anonymous namespace)::', '').replace('void ', '')[:70]}")
