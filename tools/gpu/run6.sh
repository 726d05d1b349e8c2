set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "sampl" > gpurun_out/t6_kern.log 2>&1 || { tail -30 gpurun_out/t6_kern.log; exit 1; }
tail -2 gpurun_out/t6_kern.log
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py -m gpu -x -q -k "config3 or config4" > gpurun_out/t6_eng.log 2>&1 || { tail -40 gpurun_out/t6_eng.log; exit 1; }
tail -2 gpurun_out/t6_eng.log
timeout -k 10 200 python tools/timeline_sample.py gpurun_out/timeline_sample.json || exit 1
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b6.json 2> gpurun_out/b6.log || { tail -30 gpurun_out/b6.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b6.json"))
print(j["value"], j["phases_ms"], j["decode_step"]["us"], j["first_token_ms"])
PY
echo ALLDONE
