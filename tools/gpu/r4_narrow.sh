set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "conv" > gpurun_out/r4_narrow_tests.log 2>&1 || { tail -40 gpurun_out/r4_narrow_tests.log; exit 1; }
tail -2 gpurun_out/r4_narrow_tests.log
echo "== third form (LDS-staged rows)"; timeout -k 10 300 python3 tools/microbench_gemm.py 10 narrow 2>&1 | grep narrow
echo "== second form"; ITTS_CONV_CFG=31 timeout -k 10 300 python3 tools/microbench_gemm.py 10 narrow 2>&1 | grep narrow
echo ALLDONE
