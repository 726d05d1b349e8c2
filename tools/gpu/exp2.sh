set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/timeline_conv.py 96 7 1 35840 || exit 1
timeout -k 10 120 python tools/timeline_conv.py 192 7 1 8960 || exit 1
timeout -k 10 120 python tools/timeline_conv.py 768 11 1 560 || exit 1
echo ALLDONE
