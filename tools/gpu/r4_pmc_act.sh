set -o pipefail
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export ACT_C=48
for L in hip hip_aa0; do
rm -rf gpurun_out/pmc_act
ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_$L.so rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d gpurun_out/pmc_act -o g -- python3 tools/microbench_act.py 3 > gpurun_out/pmc_act_$L.log 2>&1 || { tail -20 gpurun_out/pmc_act_$L.log; exit 1; }
L=$L python3 - <<'PY'
import csv, glob, collections, os
f=glob.glob("gpurun_out/pmc_act/**/*counter_collection.csv", recursive=True)
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:40]
    if "aa_snake" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="SQ_WAVE_CYCLES": n[k]+=1
for k,v in agg.items():
    print(os.environ["L"], k, n[k], {a: round(b/n[k]/1e6,2) for a,b in v.items()})
PY
done
rm -rf gpurun_out/pmc_act
echo ALLDONE
