set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "beam or attn" > gpurun_out/t16_beam.log 2>&1 || { tail -40 gpurun_out/t16_beam.log; exit 1; }
tail -2 gpurun_out/t16_beam.log
timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b16.json 2> gpurun_out/b16.log || { tail -30 gpurun_out/b16.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b16.json"))
print(j["value"], j["decode_step"]["us"], j["beam_sample"])
PY
ITTS_BEAM_ATTN=rows timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b16r.json 2> gpurun_out/b16r.log || { tail -30 gpurun_out/b16r.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b16r.json"))
print("rows:", j["beam_sample"])
PY
echo ALLDONE
