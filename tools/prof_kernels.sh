#!/bin/bash
# Tuning aid: per-kernel averages of a short program under rocprofv3 (run on the GPU box).
#   tools/prof_kernels.sh OUTDIR TOPN python3 tools/perf_gat.py real
out=$1; top=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- "$@" > "$out.log" 2>&1 || { tail -5 "$out.log"; exit 1; }
python3 tools/prof_top.py "$out" "$top"
