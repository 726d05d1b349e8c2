set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "bigvgan or vocoder or config5 or public_api" > gpurun_out/t22.log 2>&1; rc=$?
tail -2 gpurun_out/t22.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency --no-beam > gpurun_out/b22.json 2> gpurun_out/b22.log || { tail -30 gpurun_out/b22.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/b22.json"))
print(j["value"], j["phases_ms"])
v=j["roofline"]["vocoder"]
print({k:v[k] for k in v if k!="stages"})
for s in v.get("stages",[]): print(s)
PY
echo ALLDONE
