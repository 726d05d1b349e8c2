#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_gap
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_gap -- python3 tools/phase_steps.py 3 > gpurun_out/prof_gap.log 2>&1 || { tail -20 gpurun_out/prof_gap.log; exit 1; }
f=$(find gpurun_out/prof_gap -name '*kernel_trace.csv' | head -1)
python3 tools/step_gaps.py "$f" > gpurun_out/r04_step_gaps.txt 2>&1
cat gpurun_out/r04_step_gaps.txt
tail -5 gpurun_out/phase_steps.txt
rm -rf gpurun_out/prof_gap
