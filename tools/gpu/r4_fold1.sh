set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -k "skinny or embed_step or packed or lora" > gpurun_out/r4_fold1_kernels.log 2>&1 || { tail -40 gpurun_out/r4_fold1_kernels.log; exit 1; }
tail -3 gpurun_out/r4_fold1_kernels.log
timeout -k 10 900 python3 -m pytest tests/test_engines_gpu.py -x -q -k "folded or packed_and_row or gpt_decode_logits or refill or lora" > gpurun_out/r4_fold1_engines.log 2>&1 || { tail -40 gpurun_out/r4_fold1_engines.log; exit 1; }
tail -3 gpurun_out/r4_fold1_engines.log
rm -f gpurun_out/token_time.txt
timeout -k 10 600 python3 tools/decode_token_time.py mode=launch mode=fold,rows=0:0,wide=0 mode=fold,rows=16:16,wide=0 mode=fold,rows=16:16,wide=1 mode=fold,rows=0:16,wide=1 mode=fold,rows=16:0,wide=0 2>&1 | tail -20
echo ALLDONE
