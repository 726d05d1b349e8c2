set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "rest_api or on_disk" > gpurun_out/t9.log 2>&1 || { tail -40 gpurun_out/t9.log; exit 1; }
tail -3 gpurun_out/t9.log
echo ALLDONE
