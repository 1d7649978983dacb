#!/bin/bash
# kernel stats of a short default bench run: the selection kernel's and the class-scoring kernel's rows
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -o p -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-side-workloads --no-kernel-timing > /tmp/pq.log 2>&1 || { tail -5 /tmp/pq.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/pq/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_topk_select', 'k_score_classes', 'k_emit', 'k_pair_table', 'k_zbias')):
        print(r['Name'].replace('aig::(anonymous namespace)::','').replace('void ','')[:60].ljust(60), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f} us avg", f"min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
