#!/bin/bash
# kernel trace of the iscas85 bench loop, digested per step by tools/step_trace.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tri -o p -- python3 bench.py --workload iscas85 --no-kernel-timing --no-cpu-baseline --no-side-workloads --sample-iters 1 > /tmp/tri.log 2>&1 || { tail -5 /tmp/tri.log; exit 1; }
python3 tools/step_trace.py /tmp/tri 8 9 3 8 1 > gpurun_out/r3_trace_iscas_step.txt 2>&1
grep '^{' /tmp/tri.log | cut -c1-200
head -60 gpurun_out/r3_trace_iscas_step.txt
