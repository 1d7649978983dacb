#!/bin/bash
# kernel trace of the default bench loop, digested per step by tools/step_trace.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python3 bench.py --no-kernel-timing --no-cpu-baseline --no-side-workloads --sample-iters 1 > /tmp/tr.log 2>&1 || { tail -5 /tmp/tr.log; exit 1; }
python3 tools/step_trace.py /tmp/tr 8 9 3 8 1 > gpurun_out/r3_trace_step.txt 2>&1
python3 tools/step_trace.py /tmp/tr 4 12 > gpurun_out/r3_trace_steps.txt 2>&1
python3 tools/step_trace.py /tmp/tr 8 9 3 8 1 250 0 9000 > gpurun_out/r3_trace_all.txt 2>&1
grep '^{' /tmp/tr.log | cut -c1-300
cat gpurun_out/r3_trace_steps.txt
