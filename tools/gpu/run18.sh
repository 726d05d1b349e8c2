set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "conv" > gpurun_out/t18.log 2>&1 || { tail -40 gpurun_out/t18.log; exit 1; }
tail -2 gpurun_out/t18.log
echo ALLDONE
