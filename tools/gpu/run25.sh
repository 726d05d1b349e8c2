set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py -x -q -m gpu -k "ragged or conv or aa_snake or bigvgan" > gpurun_out/t25.log 2>&1; rc=$?
tail -25 gpurun_out/t25.log
exit $rc
