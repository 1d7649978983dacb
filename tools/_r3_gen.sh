#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_generator_gpu.py tests/test_graphset_gpu.py tests/test_cli_gpu.py tests/test_large_graphs_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r3_gen_tests.log 2>&1; rc=$?; tail -25 gpurun_out/r3_gen_tests.log
[ $rc -eq 0 ] || exit $rc
for dt in bf16 fp32; do
timeout -k 10 200 python bench.py --workload bulk_k256 --score-dtype $dt --no-cpu-baseline > gpurun_out/r3_bulk_$dt.json 2> gpurun_out/r3_bulk_$dt.err || { tail -5 gpurun_out/r3_bulk_$dt.err; exit 1; }
cut -c1-200 gpurun_out/r3_bulk_$dt.json
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-workloads > gpurun_out/r3_bench_d.json 2> gpurun_out/r3_bench_d.err || { tail -5 gpurun_out/r3_bench_d.err; exit 1; }
cut -c1-400 gpurun_out/r3_bench_d.json; grep -o "sample_nodes_per_sec[^,]*,[^,]*,[^,]*" gpurun_out/r3_bench_d.json
