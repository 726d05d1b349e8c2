set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv or gemm or vocoder or bigvgan" > gpurun_out/t_conv.log 2>&1; rc=$?
tail -5 gpurun_out/t_conv.log
[ $rc -eq 0 ] || exit $rc
for c in 0 10; do
echo "== conv cfg $c"
ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 2>&1 | grep -v "Warning\|amdgpu.ids"
done
echo "== mask 16 (no epilogue)"
ITTS_CONV_EXP=16 timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids"
echo ALLDONE
