#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_discriminator_gpu.py tests/test_train_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider > gpurun_out/r3_gp_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_gp_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err || { tail -5 gpurun_out/r3_bench_e.err; exit 1; }
cut -c1-330 gpurun_out/r3_bench_e.json
grep -o '"other_workloads.*ms_per_step": [0-9.]*' gpurun_out/r3_bench_e.json | tail -c 60
timeout -k 10 300 python tools/probe_times.py > gpurun_out/r3_probe_times3.txt 2>&1; tail -22 gpurun_out/r3_probe_times3.txt
