#!/bin/bash
# kernel stats of bulk sampling on the 100k-node workload (tools/sample_breakdown.py: 8 blocks of 10 draws)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -o p -- python3 tools/sample_breakdown.py > /tmp/ps.log 2>&1 || { tail -5 /tmp/ps.log; exit 1; }
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/ps/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:32]:
    print(r['Name'].replace('aig::(anonymous namespace)::','').replace('void ','')[:72].ljust(72), r['Calls'].rjust(5), f"{float(r['TotalDurationNs'])/1e3/8:9.1f} us/block", f"{float(r['AverageNs'])/1e3:8.1f} us avg", r['Percentage'])
PY
