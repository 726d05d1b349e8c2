set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "aa_snake" > gpurun_out/r4_voc_kernels.log 2>&1 || { tail -40 gpurun_out/r4_voc_kernels.log; exit 1; }
tail -2 gpurun_out/r4_voc_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -x -q -k "bigvgan or vocoder or config5" > gpurun_out/r4_voc_engines.log 2>&1 || { tail -60 gpurun_out/r4_voc_engines.log; exit 1; }
tail -2 gpurun_out/r4_voc_engines.log
timeout -k 10 300 python3 tools/vocoder_time.py 2>&1 | tail -12
echo ALLDONE
