set -o pipefail
cd $GRAFT_REPO_ROOT
for e in 0 32; do for c in 0 10 13; do
echo "== conv exp $e cfg $c"
ITTS_CONV_EXP=$e ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids\|C48\|C24"
done; done
echo "== C768 k11"
timeout -k 10 200 python tools/timeline_conv.py 768 11 1 560 2>&1 | grep -A7 "chunk1_us"
echo ALLDONE
