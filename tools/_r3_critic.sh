#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest -m gpu tests/test_discriminator_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_critic_tests.log 2>&1
rc=$?
tail -25 gpurun_out/r3_critic_tests.log
exit $rc
