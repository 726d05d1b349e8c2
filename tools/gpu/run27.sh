set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/t27.log 2>&1; rc=$?
tail -3 gpurun_out/t27.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/t27.log; exit $rc; }
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_config4.json 2> gpurun_out/r02_bench_config4.log || { tail -30 gpurun_out/r02_bench_config4.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r02_bench_config4.json')); print('config4', j['value'], j['ms_per_step'], j['phases_ms'])"
timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline --no-beam > gpurun_out/b27.json 2> gpurun_out/b27.log || { tail -30 gpurun_out/b27.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/b27.json')); print('config3', j['value'], j['phases_ms'], j['decode_step']['us'])"
echo ALLDONE
