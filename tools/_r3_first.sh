#!/bin/bash
# round-3 first GPU pass: new bench-size step tests, the parallel tests, then the two-rank bench once
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -m gpu tests/test_parallel_gpu.py tests/test_train_gpu.py -x -q --timeout 900 --timeout-method thread -k "parallel or rccl or bench_step or two_rank" > gpurun_out/r3_first_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r3_first_tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/_two_rank.sh r3 --workload synth100k
