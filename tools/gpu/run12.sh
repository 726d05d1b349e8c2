set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/t12_kern.log 2>&1 || { tail -30 gpurun_out/t12_kern.log; exit 1; }
tail -2 gpurun_out/t12_kern.log
timeout -k 10 900 python -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/t12_eng.log 2>&1 || { tail -40 gpurun_out/t12_eng.log; exit 1; }
tail -2 gpurun_out/t12_eng.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b12.json 2> gpurun_out/b12.log || { tail -30 gpurun_out/b12.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b12.json"))
print(j["value"], j["phases_ms"], j["decode_step"]["us"], j["first_token_ms"])
PY
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/side12.log 2>&1; tail -3 gpurun_out/side12.log
echo ALLDONE
