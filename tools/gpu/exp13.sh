set -o pipefail
cd $GRAFT_REPO_ROOT
for c in 0 2; do
echo "== conv cfg $c"
ITTS_CONV_CFG=$c timeout -k 10 200 python tools/microbench_gemm.py 10 "voc C" 2>&1 | grep -v "Warning\|amdgpu.ids"
done
echo ALLDONE
