#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0 1 2; do
export AIG_REACH_ONLY=$o
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp$o -o p -- python3 tools/perf_penalties.py > /tmp/pp.log 2>&1 || { tail -5 /tmp/pp.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/pp$o/**/*kernel_stats.csv', recursive=True)[0]
for r in [r for r in csv.DictReader(open(f)) if 'k_reach' in r['Name'] or 'k_pen' in r['Name'] or 'sweep' in r['Name']]:
    print($o, r['Name'][:60].ljust(60), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
done
