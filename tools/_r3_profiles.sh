#!/bin/bash
# Evidence for profiles/: everything lands in gpurun_out/r03/ under the names it gets in profiles/ (the r03_ prefix is the naming scheme of that directory).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/r03
mkdir -p $o
step() { echo "== $1"; }
step memset-probe
timeout -k 10 120 python3 tools/graph_memset_probe2.py > $o/r03_graph_memset_probe.txt 2>&1 || { tail -5 $o/r03_graph_memset_probe.txt; exit 1; }
step stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb -o p -- python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline --no-side-workloads > /tmp/pb.log 2>&1 || { tail -5 /tmp/pb.log; exit 1; }
cp $(find /tmp/pb -name "*kernel_stats.csv" | head -1) $o/r03_bench_synth100k_kernel_stats.csv
grep '^{' /tmp/pb.log > $o/r03_bench_synth100k_profiled.json
step bulk-stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pbk -o p -- python3 bench.py --workload bulk_k256 --score-dtype bf16 --steps 16 --warmup 3 > /tmp/pbk.log 2>&1 || { tail -5 /tmp/pbk.log; exit 1; }
cp $(find /tmp/pbk -name "*kernel_stats.csv" | head -1) $o/r03_bench_bulk_k256_kernel_stats.csv
step pmc
bash tools/pmc_passes.sh /tmp/pmc || exit 1
python3 tools/pmc_summary.py --json real=/tmp/pmc/FETCH_SIZE_real,/tmp/pmc/WRITE_SIZE_real fake=/tmp/pmc/FETCH_SIZE_fake,/tmp/pmc/WRITE_SIZE_fake > $o/pmc_traffic.json || exit 1
for k in real fake; do
  python3 tools/pmc_summary.py /tmp/pmc/FETCH_SIZE_$k /tmp/pmc/WRITE_SIZE_$k > $o/r03_pmc_fetch_write_critic_$k.csv
  cp $(find /tmp/pmc/stats_$k -name "*kernel_stats.csv" | head -1) $o/r03_critic_${k}_kernel_stats.csv
done
cp $o/pmc_traffic.json profiles/pmc_traffic.json; cp $o/r03_bench_synth100k_kernel_stats.csv profiles/   # (the bench lines below read them)
step bench-lines
timeout -k 10 400 python3 bench.py > $o/r03_bench_default_with_cpu_baseline.json 2> $o/default.err || { tail -5 $o/default.err; exit 1; }
timeout -k 10 200 python3 bench.py --workload iscas85 --no-cpu-baseline > $o/r03_bench_iscas85.json 2>/dev/null || exit 1
timeout -k 10 300 python3 bench.py --workload synth1m --no-cpu-baseline --no-side-workloads > $o/r03_bench_synth1m.json 2>/dev/null || exit 1
for dt in bf16 fp32; do timeout -k 10 200 python3 bench.py --workload bulk_k256 --score-dtype $dt --steps 16 --warmup 3 > $o/r03_bench_bulk_k256_$dt.json 2>/dev/null || exit 1; done
for f in $o/r03_bench_*.json; do echo $f; cut -c1-260 $f; done
echo done
