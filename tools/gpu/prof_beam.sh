set -o pipefail
R=${R:-r03}   # round tag of the output files
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_beam -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-concurrency --no-roofline > gpurun_out/prof_beam.json 2> gpurun_out/prof_beam.log || { tail -20 gpurun_out/prof_beam.log; exit 1; }
R=$R python3 - <<'PY'
import csv, glob
f=[x for x in glob.glob("gpurun_out/prof_beam/**/*kernel_stats.csv", recursive=True)]
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:60]:
    n=r["Name"]
    if "itts" in n:
        print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms n={r["Calls"]:>7} avg={float(r["AverageNs"])/1e3:8.2f} us  {n[:130]}')
PY
cp "$(find gpurun_out/prof_beam -name "*kernel_stats.csv" | head -1)" gpurun_out/${R}_beam_kernel_stats.csv
rm -rf gpurun_out/prof_beam   # the per-dispatch trace is > 64 MiB: only the stats summary travels back
echo ALLDONE
