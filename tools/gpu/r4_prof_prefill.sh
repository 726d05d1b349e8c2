set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pf -- python3 tools/first_token_split.py > gpurun_out/prof_pf.log 2>&1 || { tail -20 gpurun_out/prof_pf.log; exit 1; }
python3 - <<'PY'
import csv, glob
f=glob.glob("gpurun_out/prof_pf/**/*kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per iteration", tot/1e6/24)
for r in rows[:22]:
    print(f'{float(r["TotalDurationNs"])/1e6/24:8.3f} ms/iter n/iter={int(r["Calls"])/24:7.1f} avg={float(r["AverageNs"])/1e3:8.2f} us  {r["Name"][:100]}')
# the launch sequence of the LAST iteration, run-length coded (where do the small copies sit?)
t=glob.glob("gpurun_out/prof_pf/**/*kernel_trace.csv", recursive=True)
tr=list(csv.DictReader(open(t[0])))
tr.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in tr]
n=len(names)
seq=names[n - n//24:]
out=[]
for k in seq:
    k=k[:60]
    if out and out[-1][0]==k: out[-1][1]+=1
    else: out.append([k,1])
for k,c in out: print(c, k)
PY
rm -rf gpurun_out/prof_pf
echo ALLDONE
