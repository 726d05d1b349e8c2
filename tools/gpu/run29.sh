set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_config4.json 2> gpurun_out/r02_bench_config4.log || { tail -30 gpurun_out/r02_bench_config4.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r02_bench_config4.json')); print('config4', j['value'], j['ms_per_step'], j['phases_ms'])"
echo ALLDONE
