#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests/test_discriminator_gpu.py -x -q -m gpu --timeout 180 --timeout-method thread -p no:cacheprovider -k "gradient_penalty or closed_form" > gpurun_out/r3_gp_tests.log 2>&1; rc=$?
tail -30 gpurun_out/r3_gp_tests.log
exit $rc
