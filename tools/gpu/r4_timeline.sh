set -o pipefail
cd $GRAFT_REPO_ROOT
export ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_hip_diag.so
timeout -k 10 300 python3 tools/timeline_skinny.py --mode fold --out gpurun_out/r04_timeline_fold.json > /dev/null 2> gpurun_out/tl_fold.log || { tail gpurun_out/tl_fold.log; exit 1; }
timeout -k 10 300 python3 tools/timeline_skinny.py --mode launch --out gpurun_out/r04_timeline_launch.json > /dev/null 2> gpurun_out/tl_launch.log || { tail gpurun_out/tl_launch.log; exit 1; }
echo ALLDONE
