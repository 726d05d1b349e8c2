set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv or gemm or vocoder or bigvgan" > gpurun_out/t_conv.log 2>&1; rc=$?
tail -3 gpurun_out/t_conv.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids"
echo ALLDONE
