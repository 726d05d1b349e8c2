set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "skinny or ln_reduce or layernorm" > gpurun_out/t2_kern.log 2>&1 || { tail -30 gpurun_out/t2_kern.log; exit 1; }
tail -3 gpurun_out/t2_kern.log
timeout -k 10 900 python -m pytest tests/test_engines_gpu.py tests/test_configs_gpu.py -m gpu -x -q > gpurun_out/t2_eng.log 2>&1 || { tail -40 gpurun_out/t2_eng.log; exit 1; }
tail -3 gpurun_out/t2_eng.log
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency > gpurun_out/b2_tail.json 2> gpurun_out/b2_tail.log || { tail -30 gpurun_out/b2_tail.log; exit 1; }
ITTS_DECODE_MODE=launch timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency > gpurun_out/b2_launch.json 2> gpurun_out/b2_launch.log || { tail -30 gpurun_out/b2_launch.log; exit 1; }
ITTS_TAIL_ACQUIRE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/b2_tail_acq.json 2> gpurun_out/b2_tail_acq.log || { tail -30 gpurun_out/b2_tail_acq.log; exit 1; }
python - <<'PY'
import json
for n in ("tail","launch","tail_acq"):
    j=json.load(open(f"gpurun_out/b2_{n}.json"))
    print(n, j["value"], j["phases_ms"], j["decode_step"]["us"], j.get("roofline",{}).get("avg_launch_us"))
PY
timeout -k 10 300 python tools/timeline_skinny.py --mode tail --out gpurun_out/timeline_tail.json > /dev/null 2> gpurun_out/timeline_tail.log || { tail -30 gpurun_out/timeline_tail.log; exit 1; }
timeout -k 10 300 python tools/timeline_skinny.py --mode launch --out gpurun_out/timeline_launch.json > /dev/null 2> gpurun_out/timeline_launch.log || { tail -30 gpurun_out/timeline_launch.log; exit 1; }
echo ALLDONE
