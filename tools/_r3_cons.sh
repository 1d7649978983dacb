#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest -m gpu tests/test_graphset_gpu.py tests/test_constraints_gpu.py tests/test_large_graphs_gpu.py tests/test_level_sweep_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_cons_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r3_cons_tests.log
exit $rc
