set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/final_gpu_tests.log 2>&1 || { tail -40 gpurun_out/final_gpu_tests.log; exit 1; }
tail -3 gpurun_out/final_gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" > gpurun_out/final_smoke.log 2>&1 || { tail -30 gpurun_out/final_smoke.log; exit 1; }
tail -2 gpurun_out/final_smoke.log
timeout -k 10 600 python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.log || { tail -30 gpurun_out/r02_bench.log; exit 1; }
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r02_bench.json"))
for k in ("value","ms_per_step","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests"):
    print(k, j.get(k))
print(json.dumps(j["roofline"])[:400])
PY
echo ALLDONE_C
