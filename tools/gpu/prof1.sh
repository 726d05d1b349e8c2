set -o pipefail
R=${R:-r03}   # round tag of the output files
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R} -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-concurrency --no-beam --no-accuracy > gpurun_out/prof_${R}_bench.json 2> gpurun_out/prof_${R}_bench.log || { tail -20 gpurun_out/prof_${R}_bench.log; exit 1; }
find gpurun_out/prof_${R} -name "*stats*" | head
R=$R python3 - <<'PY'
import csv, glob, os
f=[x for x in glob.glob(f"gpurun_out/prof_{os.environ['R']}/**/*kernel_stats.csv", recursive=True)]
print(f)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% n={r["Calls"]:>7} avg={float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:150]}')
PY
cp "$(find gpurun_out/prof_${R} -name "*kernel_stats.csv" | head -1)" gpurun_out/${R}_bench_kernel_stats.csv
rm -rf gpurun_out/prof_${R}   # the per-dispatch trace is > 64 MiB: only the stats summary travels back
echo ALLDONE
