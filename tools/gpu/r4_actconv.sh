#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "act_conv" > gpurun_out/r04_actconv_tests.log 2>&1 || { tail -40 gpurun_out/r04_actconv_tests.log; exit 1; }
tail -3 gpurun_out/r04_actconv_tests.log
ITTS_FUSE_ACT_CONV=0 python tools/vocoder_time.py > gpurun_out/r04_voc_unfused.txt 2>&1 || { tail -20 gpurun_out/r04_voc_unfused.txt; exit 1; }
tail -3 gpurun_out/r04_voc_unfused.txt
python tools/vocoder_time.py > gpurun_out/r04_voc_fused.txt 2>&1 || { tail -20 gpurun_out/r04_voc_fused.txt; exit 1; }
tail -3 gpurun_out/r04_voc_fused.txt
