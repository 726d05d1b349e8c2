#!/bin/bash
# engine tests (prefill / paged / refill paths) + the first-token split
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests/test_engines_gpu.py tests/test_frontend_gpu.py -x -q > gpurun_out/r04_eng_tests.log 2>&1 || { tail -40 gpurun_out/r04_eng_tests.log; exit 1; }
tail -3 gpurun_out/r04_eng_tests.log
python tools/first_token_split.py > gpurun_out/r04_first_token_split.txt 2>&1 || { tail -20 gpurun_out/r04_first_token_split.txt; exit 1; }
tail -6 gpurun_out/r04_first_token_split.txt
