set -o pipefail
cd $GRAFT_REPO_ROOT
for e in 0 7 15 31 16; do
echo "== conv exp mask $e"
ITTS_CONV_EXP=$e timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids"
done
for c in 10 11 12; do
echo "== conv cfg $c"
ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids"
done
echo ALLDONE
