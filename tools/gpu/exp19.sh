set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python tools/timeline_narrow.py 24 7 1 143360 2>&1 | grep -v "Warning\|amdgpu.ids"
timeout -k 10 200 python tools/timeline_narrow.py 48 7 1 71680 2>&1 | grep -v "Warning\|amdgpu.ids"
echo ALLDONE
