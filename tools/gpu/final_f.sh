set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/r02_side_configs.log 2>&1; tail -1 gpurun_out/r02_side_configs.log > gpurun_out/r02_side_configs.json; cat gpurun_out/r02_side_configs.json
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_config4.json 2> gpurun_out/r02_bench_config4.log || { tail -30 gpurun_out/r02_bench_config4.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r02_bench_config4.json')); print('config4', j['value'], j['ms_per_step'], j['phases_ms'])"
ITTS_BENCH_ONE_DEVICE=1 ITTS_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/r02_bench_2ranks_1gpu.json 2> gpurun_out/r02_bench_2ranks_1gpu.log || { tail -30 gpurun_out/r02_bench_2ranks_1gpu.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/r02_bench_2ranks_1gpu.json')); print('2ranks', j['value'], j['n_gpus'], j['per_rank'], j['weight_broadcast'], j['tail_imbalance'])"
echo ALLDONE_F
