set -o pipefail
cd $GRAFT_REPO_ROOT
for e in 0 31 7; do
echo "== C768 k11 mask $e"
ITTS_CONV_EXP=$e timeout -k 10 200 python tools/timeline_conv.py 768 11 1 560 2>&1 | grep -v "Warning\|amdgpu.ids"
done
for e in 0 31; do
echo "== C96 k7 mask $e"
ITTS_CONV_EXP=$e timeout -k 10 200 python tools/timeline_conv.py 96 7 1 35840 2>&1 | grep -v "Warning\|amdgpu.ids"
done
echo ALLDONE
