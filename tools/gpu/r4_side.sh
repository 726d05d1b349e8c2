# side configurations of a round: configs 2 / 5, config 4's shard, continuous batching on a queue, two ranks on one GPU
set -o pipefail
R=${R:-r04}
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 tools/bench_configs.py > gpurun_out/${R}_side_configs.json 2> gpurun_out/${R}_side.log || { tail -20 gpurun_out/${R}_side.log; exit 1; }
cat gpurun_out/${R}_side_configs.json
timeout -k 10 400 python3 bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${R}_bench_config4.json 2> gpurun_out/${R}_bench_config4.log || { tail -30 gpurun_out/${R}_bench_config4.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/${R}_bench_config4.json')); print('config4', j['value'], j['ms_per_step'], j['phases_ms'], j['config']['audio_seconds_per_step_per_gpu'])"
timeout -k 10 600 python3 tools/bench_refill.py 4 > gpurun_out/${R}_refill_config4.json 2> gpurun_out/${R}_refill.log || { tail -30 gpurun_out/${R}_refill.log; exit 1; }
cat gpurun_out/${R}_refill_config4.json | head -c 1500; echo
ITTS_BENCH_ONE_DEVICE=1 ITTS_BENCH_BACKEND=gloo timeout -k 10 500 python3 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/${R}_bench_2ranks_1gpu.json 2> gpurun_out/${R}_bench_2ranks_1gpu.log || { tail -30 gpurun_out/${R}_bench_2ranks_1gpu.log; exit 1; }
python3 -c "
import json; j=json.load(open('gpurun_out/${R}_bench_2ranks_1gpu.json')); print('2ranks', j['value'], j['n_gpus'], j['per_rank'], j['weight_broadcast'], j['tail_imbalance'])"
echo ALLDONE
