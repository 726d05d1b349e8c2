set -o pipefail
R=${R:-r03}   # round tag of the output files
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o g -- python3 tools/pmc_decode_gemm.py > gpurun_out/pmc_fetch.log 2>&1 || { tail -20 gpurun_out/pmc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o g -- python3 tools/pmc_decode_gemm.py > gpurun_out/pmc_write.log 2>&1 || { tail -20 gpurun_out/pmc_write.log; exit 1; }
ALG=$(grep -o "per launch [0-9]*" gpurun_out/pmc_write.log | grep -o "[0-9]*")
python3 tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write $ALG gpurun_out/${R}_pmc_gemm_skinny.json | tee gpurun_out/${R}_pmc_gemm_skinny.txt
cp gpurun_out/${R}_pmc_gemm_skinny.json profiles/ 2>/dev/null
timeout -k 10 600 python3 bench.py > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.log || { tail -30 gpurun_out/${R}_bench.log; exit 1; }
R=$R python3 - <<'PY'
import json, os
j=json.load(open(f"gpurun_out/{os.environ['R']}_bench.json"))
for k in ("value","ms_per_step","first_token_ms","phases_ms","decode_step","beam_sample","concurrent_requests","cpu_baseline"):
    print(k, j.get(k))
print(json.dumps(j["roofline"])[:1500])
PY
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${R}_bench_config4.json 2> gpurun_out/${R}_bench_config4.log || { tail -30 gpurun_out/${R}_bench_config4.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/${R}_bench_config4.json')); print('config4', j['value'], j['ms_per_step'], j['phases_ms'], j['config']['audio_seconds_per_step_per_gpu'])"
ITTS_BENCH_ONE_DEVICE=1 ITTS_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${R}_bench_2ranks_1gpu.json 2> gpurun_out/${R}_bench_2ranks_1gpu.log || { tail -30 gpurun_out/${R}_bench_2ranks_1gpu.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/${R}_bench_2ranks_1gpu.json')); print('2ranks', j['value'], j['n_gpus'], j['ranks_seen'], j['per_rank'], j['weight_broadcast'], j['tail_imbalance'])"
echo ALLDONE
