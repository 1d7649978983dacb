cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for o in 0; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$o -o lin -- python3 $GRAFT_REPO_ROOT/tools/prof_linear.py > /tmp/prof_$o.log 2>&1 || { tail -5 /tmp/prof_$o.log; exit 1; }
  echo "OLD=$o"; python - <<PY
import csv, glob, collections
f = glob.glob('/tmp/prof_$o/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'linear128' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for gi, n in enumerate((26411, 100000, 1000000)):
    for act in ('<0>', '<2>'):
        v = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[gi*30:(gi+1)*30] if act in r['Kernel_Name'])
        print(n, act, len(v), 'median %.1f us min %.1f  -> %.1f TFLOP/s' % (v[len(v)//2], v[0], 2*n*128*128/v[len(v)//2]*1e-6))
PY
done
