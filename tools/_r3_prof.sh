#!/bin/bash
# kernel trace summary of the default bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
tag=${1:-r3}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-side-workloads > /tmp/pb.log 2>&1 || { tail -5 /tmp/pb.log; exit 1; }
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${tag}_bench_synth100k_kernel_stats.csv
grep '^{' /tmp/pb.log > gpurun_out/prof/${tag}_bench_synth100k_profiled.json
echo done
