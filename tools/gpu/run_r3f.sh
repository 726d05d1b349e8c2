set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py -x -q -m gpu -k "attn or decode or beam or finished or sampling_loop" > gpurun_out/t_r3f.log 2>&1 || { tail -40 gpurun_out/t_r3f.log; exit 1; }
tail -3 gpurun_out/t_r3f.log
timeout -k 10 300 python3 tools/decode_token_time.py "" qofp > gpurun_out/token_time_f.log 2>&1 || { tail -30 gpurun_out/token_time_f.log; exit 1; }
cat gpurun_out/token_time_f.log
