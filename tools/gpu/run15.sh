set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "beam" > gpurun_out/t15_beam.log 2>&1 || { tail -40 gpurun_out/t15_beam.log; exit 1; }
tail -2 gpurun_out/t15_beam.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b15.json 2> gpurun_out/b15.log || { tail -30 gpurun_out/b15.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b15.json"))
print(j["value"], j["decode_step"]["us"], j["beam_sample"])
PY
echo ALLDONE
