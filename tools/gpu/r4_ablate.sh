set -o pipefail
cd $GRAFT_REPO_ROOT
export ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_hip_diag.so
for A in 0 1 2 3; do echo "== ablate $A"; ITTS_ABLATE=$A timeout -k 10 200 python3 tools/probes/ab_fold_gemm.py 2>&1 | grep "us per" | tail -1; done
echo ALLDONE
