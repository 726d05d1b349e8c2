#!/bin/bash
# conditioner timing + per-kernel profile + first-token split
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/conditioner_time.py 300 300 > gpurun_out/r04_conditioner_time.txt 2>&1 || { tail -20 gpurun_out/r04_conditioner_time.txt; exit 1; }
cat gpurun_out/r04_conditioner_time.txt
rm -rf gpurun_out/prof_cond
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cond -- python3 tools/conditioner_time.py 300 50 > gpurun_out/r04_conditioner_prof.log 2>&1 || { tail -20 gpurun_out/r04_conditioner_prof.log; exit 1; }
f=$(find gpurun_out/prof_cond -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r04_conditioner_kernel_stats.csv
head -40 gpurun_out/r04_conditioner_kernel_stats.csv | cut -c1-160
true || python tools/first_token_split.py > gpurun_out/r04_first_token_split.txt 2>&1 || { tail -20 gpurun_out/r04_first_token_split.txt; exit 1; }
true
