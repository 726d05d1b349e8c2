# full checkpoint of a round: GPU tests, smoke, PMC traffic of the dominant kernel, bench line, rocprof stats of the bench command
set -o pipefail
R=${R:-r04}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/${R}_gputests.log 2>&1 || { tail -60 gpurun_out/${R}_gputests.log; exit 1; }
tail -2 gpurun_out/${R}_gputests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${R}_smoke.log 2>&1 || { tail -30 gpurun_out/${R}_smoke.log; exit 1; }
tail -1 gpurun_out/${R}_smoke.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o g -- python3 tools/pmc_decode_gemm.py > gpurun_out/pmc_fetch.log 2>&1 || { tail -20 gpurun_out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o g -- python3 tools/pmc_decode_gemm.py > gpurun_out/pmc_write.log 2>&1 || { tail -20 gpurun_out/pmc_write.log; exit 1; }
ALG=$(grep -o "per launch [0-9]*" gpurun_out/pmc_write.log | grep -o "[0-9]*")
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write $ALG gpurun_out/${R}_pmc_gemm_skinny.json | tee gpurun_out/${R}_pmc_gemm_skinny.txt
cp gpurun_out/${R}_pmc_gemm_skinny.json profiles/ 2>/dev/null
cp "$(find gpurun_out/pmc_fetch -name "*counter_collection.csv" | head -1)" gpurun_out/${R}_pmc_fetch_size.csv
cp "$(find gpurun_out/pmc_write -name "*counter_collection.csv" | head -1)" gpurun_out/${R}_pmc_write_size.csv
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrency --no-beam --no-accuracy > gpurun_out/prof_${R}_bench.json 2> gpurun_out/prof_${R}_bench.log || { tail -20 gpurun_out/prof_${R}_bench.log; exit 1; }
cp "$(find gpurun_out/prof_${R} -name "*kernel_stats.csv" | head -1)" gpurun_out/${R}_bench_kernel_stats.csv
cp gpurun_out/${R}_bench_kernel_stats.csv profiles/
rm -rf gpurun_out/prof_${R}
timeout -k 10 900 python3 bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.log || { tail -30 gpurun_out/${R}_bench.log; exit 1; }
R=$R python3 - <<'PY'
import json, os
j=json.load(open(f"gpurun_out/{os.environ['R']}_bench.json"))
for k in ("value","ms_per_step","value_new_prompt","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests"):
    print(k, j.get(k))
print("cpu", {k:v for k,v in j["cpu_baseline"].items() if k!="reference_in_build_container"})
r=j.get("roofline")
print({k:v for k,v in r.items() if k not in ("vocoder","latent_pass_mfma")})
print(r["vocoder"]["ms"], [ (s["stage"][:8], s["ms"]) for s in r["vocoder"].get("stages",[])])
print(json.dumps(j.get("kernel_breakdown"))[:1200])
print("accuracy", json.dumps(j.get("accuracy"))[:600])
PY
echo ALLDONE
