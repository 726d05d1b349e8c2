set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_engines_gpu.py -m gpu -x -q -k "attn or decode or beam or sampling" > gpurun_out/t28.log 2>&1; rc=$?
tail -2 gpurun_out/t28.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/t28.log; exit $rc; }
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrency --no-roofline --no-beam > gpurun_out/b28.json 2> gpurun_out/b28.log || { tail -30 gpurun_out/b28.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/b28.json')); print('config3', j['value'], j['phases_ms']['prefilled->decoded'], j['decode_step']['us'])"
done
echo ALLDONE
