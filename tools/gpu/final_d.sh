set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.log || { tail -30 gpurun_out/r02_bench.log; exit 1; }
python3 - <<'PY'
import json
j=json.load(open("gpurun_out/r02_bench.json"))
for k in ("value","ms_per_step","first_token_ms_p50","phases_ms","decode_step","beam_sample"):
    print(k, j.get(k))
print(j["roofline"]["frac"], j["roofline"]["avg_launch_us"], len(j["roofline"]["vocoder"].get("stages",[])), j["cpu_baseline"]["value"])
PY
echo ALLDONE_D
