set -o pipefail
bash tools/gpu/prof1.sh || exit 1
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/r02_side_configs.log 2>&1; tail -1 gpurun_out/r02_side_configs.log > gpurun_out/r02_side_configs.json; cat gpurun_out/r02_side_configs.json
for sh in "768 11 1 560" "768 3 5 560" "192 7 1 8960"; do
  timeout -k 10 200 python tools/timeline_conv.py $sh 2>/dev/null | grep -v "Warning\|amdgpu.ids"
done > gpurun_out/r02_timeline_conv.json
tail -30 gpurun_out/r02_timeline_conv.json
echo ALLDONE_A
