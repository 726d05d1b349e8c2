set -o pipefail
R=${R:-r04}
cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${R}_gputests.log 2>&1 || { tail -60 gpurun_out/${R}_gputests.log; exit 1; }
tail -3 gpurun_out/${R}_gputests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1 || { tail -30 gpurun_out/${R}_smoke.log; exit 1; }
tail -2 gpurun_out/${R}_smoke.log
timeout -k 10 900 python3 bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.log || { tail -30 gpurun_out/${R}_bench.log; exit 1; }
R=$R python3 - <<'PY'
import json, os
j=json.load(open(f"gpurun_out/{os.environ['R']}_bench.json"))
for k in ("value","ms_per_step","value_new_prompt","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests","cpu_baseline","accuracy"):
    print(k, j.get(k))
r=j["roofline"]; print({k:v for k,v in r.items() if k not in ("vocoder","latent_pass_mfma")})
print(r["vocoder"]["ms"], [ (s["stage"], s["ms"]) for s in r["vocoder"].get("stages",[])])
PY
echo ALLDONE
