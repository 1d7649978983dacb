#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
tag=${1:-r3}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbk -o p -- python3 bench.py --workload bulk_k256 --score-dtype bf16 --steps 16 --warmup 3 > /tmp/pbk.log 2>&1 || { tail -5 /tmp/pbk.log; exit 1; }
cp $(find /tmp/pbk -name "*kernel_stats.csv" | head -1) gpurun_out/prof/${tag}_bench_bulk_k256_bf16_kernel_stats.csv
grep '^{' /tmp/pbk.log > gpurun_out/prof/${tag}_bench_bulk_k256_bf16_profiled.json
timeout -k 10 200 python3 bench.py --workload bulk_k256 --score-dtype bf16 --steps 16 --warmup 3 > gpurun_out/prof/${tag}_bench_bulk_k256_bf16.json 2>/dev/null
echo done
