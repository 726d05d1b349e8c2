set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -40 gpurun_out/final_gpu_tests.log; exit 1; }
tail -2 gpurun_out/final_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1 || { tail -30 gpurun_out/final_smoke.log; exit 1; }
tail -1 gpurun_out/final_smoke.log
echo ALLDONE_E
