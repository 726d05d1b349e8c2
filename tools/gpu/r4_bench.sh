set -o pipefail
R=${R:-r04}
cd $GRAFT_REPO_ROOT
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print(len(os.sched_getaffinity(0)))"
timeout -k 10 900 python3 bench.py $BENCH_ARGS > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.log || { tail -30 gpurun_out/${R}_bench.log; exit 1; }
R=$R python3 - <<'PY'
import json, os
j=json.load(open(f"gpurun_out/{os.environ['R']}_bench.json"))
for k in ("value","ms_per_step","value_new_prompt","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests","cpu_baseline"):
    print(k, j.get(k))
r=j.get("roofline")
if r:
    print({k:v for k,v in r.items() if k not in ("vocoder","latent_pass_mfma")})
    print(r["vocoder"]["ms"], [ (s["stage"], s["ms"]) for s in r["vocoder"].get("stages",[])])
PY
echo ALLDONE
