# Validation run of a round: the full -m gpu suite, smoke(), then the default bench line.
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/final_tests.log 2>&1 || { tail -60 gpurun_out/final_tests.log; exit 1; }
tail -3 gpurun_out/final_tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1 || { tail -30 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
timeout -k 10 600 python3 bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.log || { tail -30 gpurun_out/final_bench.log; exit 1; }
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/final_bench.json"))
for k in ("value","ms_per_step","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests","cpu_baseline"):
    print(k, j.get(k))
print(json.dumps(j.get("accuracy"))[:900])
r=j["roofline"]; print({k:r[k] for k in ("achieved","frac","traffic","avg_launch_us") if k in r})
PY
