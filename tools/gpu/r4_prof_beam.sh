set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ITTS_BEAMS=3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_tok -- python3 tools/decode_token_time.py "mode=fold,rows=32:32" > gpurun_out/prof_tok.log 2>&1 || { tail -20 gpurun_out/prof_tok.log; exit 1; }
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/prof_tok/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(f'{float(r["TotalDurationNs"])/1e6:9.2f} ms {100*float(r["TotalDurationNs"])/tot:5.1f}% n={r["Calls"]:>7} avg={float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:110]}')
PY
rm -rf gpurun_out/prof_tok
echo ALLDONE
