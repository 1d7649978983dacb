#!/bin/bash
# Tuning aid: kernel times of GpRows.rebuild at 100k nodes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o p -- python3 tools/perf_gp_rows.py > /tmp/pr.log 2>&1 || { tail -5 /tmp/pr.log; exit 1; }
grep GpRows /tmp/pr.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/pr/**/*kernel_stats.csv', recursive=True)[0]
for r in [r for r in csv.DictReader(open(f)) if int(r['Calls']) in (13, 14, 26, 27, 39, 52, 65)][:24]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), f"{float(r['AverageNs'])/1e3:8.1f} us")
PY
