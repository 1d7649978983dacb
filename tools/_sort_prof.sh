cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o p -- python3 tools/sort_time.py > /tmp/sp.log 2>&1 || { tail -5 /tmp/sp.log; exit 1; }
f=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'k_rs_' in n or 'rocprim' in n:
        m = re.search(r'(k_rs_\w+<[^>]*>|k_rs_\w+|onesweep\w*|merge_sort\w*|block_sort\w*|radix\w*)', n)
        print(f"{(m.group(1) if m else n[:50]):60s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:7.1f} max {float(r['MaxNs'])/1e3:7.1f}")
PY
