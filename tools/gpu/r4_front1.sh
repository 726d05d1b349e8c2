#!/bin/bash
# first GPU pass over the prompt front-end kernels + conditioner engine
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_frontend_gpu.py -x -q > gpurun_out/r04_front_tests.log 2>&1
rc=$?
tail -30 gpurun_out/r04_front_tests.log
exit $rc
