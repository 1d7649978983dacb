#!/bin/bash
# HBM-side traffic of the critic's kernels per graph kind (run on the GPU box): separate counter passes, as the
# counters do not fit one pass.   tools/pmc_passes.sh OUTDIR
out=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$out"
for kind in real fake; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 150 rocprofv3 --pmc $ctr --output-format csv -d "$out/${ctr}_$kind" -o p -- python3 tools/perf_critic.py $kind 5 > "$out/${ctr}_$kind.log" 2>&1 || { tail -5 "$out/${ctr}_$kind.log"; exit 1; }
  done
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats_$kind" -o p -- python3 tools/perf_critic.py $kind 10 > "$out/stats_$kind.log" 2>&1 || { tail -5 "$out/stats_$kind.log"; exit 1; }
  tail -8 "$out/stats_$kind.log"
done
