set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/t5_kern.log 2>&1 || { tail -30 gpurun_out/t5_kern.log; exit 1; }
tail -2 gpurun_out/t5_kern.log
timeout -k 10 900 python -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/t5_eng.log 2>&1 || { tail -40 gpurun_out/t5_eng.log; exit 1; }
tail -2 gpurun_out/t5_eng.log
timeout -k 10 200 python tools/timeline_sample.py gpurun_out/timeline_sample.json > /dev/null || exit 1
for pa in 1 0; do
ITTS_PACKED_ACT=$pa timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency > gpurun_out/b5_pa$pa.json 2> gpurun_out/b5_pa$pa.log || { tail -30 gpurun_out/b5_pa$pa.log; exit 1; }
done
python - <<'PY'
import json
for n in ("pa1","pa0"):
    j=json.load(open(f"gpurun_out/b5_{n}.json"))
    print(n, j["value"], j["phases_ms"], j["decode_step"]["us"], j.get("roofline",{}).get("avg_launch_us"), j["first_token_ms"])
    print({k:(v["launches"],v["ms"]) for k,v in j["kernel_breakdown"].items()})
print(open("gpurun_out/timeline_sample.json").read())
PY
timeout -k 10 300 python tools/timeline_skinny.py --mode launch --out gpurun_out/timeline_launch_pa.json > /dev/null 2> gpurun_out/timeline_launch_pa.log || { tail -30 gpurun_out/timeline_launch_pa.log; exit 1; }
python - <<'PY'
import json
j=json.load(open("gpurun_out/timeline_launch_pa.json"))
for k,v in j["kinds"].items(): print(k, v["kernel_span_us"], v["wg_median_life_us"], v["segments_us"])
print(j["between_skinny_launches_us"], j["step_span_us"])
PY
echo ALLDONE
