#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_ac
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ac -- python3 tools/vocoder_time.py > gpurun_out/prof_ac.log 2>&1 || { tail -20 gpurun_out/prof_ac.log; exit 1; }
f=$(find gpurun_out/prof_ac -name '*kernel_stats.csv' | head -1)
grep -E "act_conv|aa_snake|narrow" "$f" | cut -c1-140
rm -rf gpurun_out/prof_ac
