set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "paged or skinny or attn" > gpurun_out/r4_paged_kernels.log 2>&1 || { tail -40 gpurun_out/r4_paged_kernels.log; exit 1; }
tail -2 gpurun_out/r4_paged_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_engines_gpu.py -x -q > gpurun_out/r4_paged_engines.log 2>&1 || { tail -60 gpurun_out/r4_paged_engines.log; exit 1; }
tail -2 gpurun_out/r4_paged_engines.log
rm -f gpurun_out/token_time.txt
ITTS_PAGED_KV=0 timeout -k 10 300 python3 tools/decode_token_time.py contiguous 2>&1 | grep "us/token"
ITTS_PAGED_KV=1 timeout -k 10 300 python3 tools/decode_token_time.py paged 2>&1 | grep "us/token"
ITTS_PAGED_KV=0 timeout -k 10 300 python3 tools/decode_token_time.py contiguous 2>&1 | grep "us/token"
ITTS_PAGED_KV=1 timeout -k 10 300 python3 tools/decode_token_time.py paged 2>&1 | grep "us/token"
echo ALLDONE
