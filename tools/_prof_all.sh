cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/prof/bench_synth100k.log 2>&1 || { tail -5 gpurun_out/prof/bench_synth100k.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-side-workloads > /tmp/pb.log 2>&1 || { tail -5 /tmp/pb.log; exit 1; }
cp /tmp/pb/*/p_kernel_stats.csv gpurun_out/prof/bench_synth100k_kernel_stats.csv 2>/dev/null || cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) gpurun_out/prof/bench_synth100k_kernel_stats.csv
for dt in bf16 fp32; do timeout -k 10 200 python3 bench.py --workload bulk_k256 --score-dtype $dt --steps 16 --warmup 3 > gpurun_out/prof/bench_bulk_$dt.log 2>&1 || exit 1; done
timeout -k 10 200 python3 bench.py --workload synth1m --no-cpu-baseline --no-side-workloads > gpurun_out/prof/bench_synth1m.log 2>&1 || exit 1
timeout -k 10 200 python3 bench.py --workload iscas85 --no-cpu-baseline > gpurun_out/prof/bench_iscas85.log 2>&1 || exit 1
echo done
