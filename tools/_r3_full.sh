#!/bin/bash
# the whole GPU suite as the round's end runs it, then the default bench line
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests/ -x -q -m gpu --timeout 900 --timeout-method thread > gpurun_out/r3_full_tests.log 2>&1
rc=$?
tail -6 gpurun_out/r3_full_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r3_bench_full.json 2> gpurun_out/r3_bench_full.err
rc=$?
tail -2 gpurun_out/r3_bench_full.err
exit $rc
