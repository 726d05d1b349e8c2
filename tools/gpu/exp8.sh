set -o pipefail
cd $GRAFT_REPO_ROOT
echo "== C768 k3 d5"
timeout -k 10 200 python tools/timeline_conv.py 768 3 5 560 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== C768 k11"
timeout -k 10 200 python tools/timeline_conv.py 768 11 1 560 2>&1 | grep -v "Warning\|amdgpu.ids"
echo "== C384 k7 d3"
timeout -k 10 200 python tools/timeline_conv.py 384 7 3 2240 2>&1 | grep -v "Warning\|amdgpu.ids"
echo ALLDONE
