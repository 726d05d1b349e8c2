set -o pipefail
cd $GRAFT_REPO_ROOT
for e in 0 1 2 3; do
echo "== conv exp $e"
ITTS_CONV_EXP=$e timeout -k 10 200 python tools/microbench_gemm.py 10 voc 2>&1 | grep -v Warning
done
echo ALLDONE
