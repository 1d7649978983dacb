#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_discriminator_gpu.py tests/test_train_gpu.py -x -q -m gpu --timeout 300 --timeout-method thread -p no:cacheprovider -k "gradient_penalty or closed_form or bench_step or follows_reference" > gpurun_out/r3_gp_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r3_gp_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline --no-side-workloads > gpurun_out/r3_bench_e.json 2> gpurun_out/r3_bench_e.err || { tail -5 gpurun_out/r3_bench_e.err; exit 1; }
cut -c1-330 gpurun_out/r3_bench_e.json
bash tools/_r3_trace.sh > /dev/null 2>&1
grep -n "gpr_\|k_linear128_x6<2>" gpurun_out/r3_trace_step.txt | head
