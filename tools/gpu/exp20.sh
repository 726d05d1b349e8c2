set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "conv" > gpurun_out/t_conv.log 2>&1; rc=$?
tail -2 gpurun_out/t_conv.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep "C48\|C24\|C96"
timeout -k 10 200 python tools/timeline_narrow.py 48 7 1 71680 2>&1 | grep "us_per_launch\|workgroup life\|tiles_per"
echo ALLDONE
