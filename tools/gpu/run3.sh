set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "skinny or ln_reduce or sampl or beam" > gpurun_out/t3_kern.log 2>&1 || { tail -30 gpurun_out/t3_kern.log; exit 1; }
tail -2 gpurun_out/t3_kern.log
timeout -k 10 900 python -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/t3_eng.log 2>&1 || { tail -40 gpurun_out/t3_eng.log; exit 1; }
tail -2 gpurun_out/t3_eng.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency > gpurun_out/b3_tail.json 2> gpurun_out/b3_tail.log || { tail -30 gpurun_out/b3_tail.log; exit 1; }
ITTS_DECODE_MODE=launch timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency > gpurun_out/b3_launch.json 2> gpurun_out/b3_launch.log || { tail -30 gpurun_out/b3_launch.log; exit 1; }
python - <<'PY'
import json
for n in ("tail","launch"):
    j=json.load(open(f"gpurun_out/b3_{n}.json"))
    print(n, j["value"], j["phases_ms"], j["decode_step"]["us"], j.get("roofline",{}).get("avg_launch_us"), j["first_token_ms"])
    print({k:(v["launches"],v["ms"]) for k,v in j["kernel_breakdown"].items()})
PY
timeout -k 10 300 python tools/timeline_skinny.py --mode tail --out gpurun_out/timeline_tail.json > /dev/null 2> gpurun_out/timeline_tail.log || { tail -30 gpurun_out/timeline_tail.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/timeline_tail.json"))
for k,v in j["kinds"].items(): print(k, v["kernel_span_us"], v["wg_median_life_us"], v["segments_us"])
print(j["between_skinny_launches_us"], j["step_span_us"])
PY
echo ALLDONE
