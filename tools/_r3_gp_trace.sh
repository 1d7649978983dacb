#!/bin/bash
# kernel trace of the default bench loop; one replay of the penalty term by tools/gp_trace.py, one step kernel by
# kernel by tools/step_trace.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trg -o p -- python3 bench.py --no-kernel-timing --no-cpu-baseline --no-side-workloads --sample-iters 1 > /tmp/trg.log 2>&1 || { tail -5 /tmp/trg.log; exit 1; }
python3 tools/gp_trace.py /tmp/trg > gpurun_out/r3_gp_trace.txt 2>&1
python3 tools/step_trace.py /tmp/trg 8 9 3 8 1 1000 0 9000 > gpurun_out/r3_step_detail.txt 2>&1
grep -c . gpurun_out/r3_step_detail.txt
