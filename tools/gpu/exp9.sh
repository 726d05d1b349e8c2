set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv or gemm or vocoder or bigvgan" > gpurun_out/t_conv.log 2>&1; rc=$?
tail -3 gpurun_out/t_conv.log
[ $rc -eq 0 ] || exit $rc
for c in 0 10; do
echo "== conv cfg $c"
ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids\|C48\|C24"
done
echo "== C768 k11"
timeout -k 10 200 python tools/timeline_conv.py 768 11 1 560 2>&1 | grep -A7 "chunk1_us"
echo ALLDONE
