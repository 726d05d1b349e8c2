set -o pipefail
cd $GRAFT_REPO_ROOT
for e in 0 32; do for c in 0 10; do
echo "== conv exp $e cfg $c"
ITTS_CONV_EXP=$e ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids\|C48\|C24"
done; done
echo ALLDONE
