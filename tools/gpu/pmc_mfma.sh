# MFMA-busy cycles per kernel (rocprofv3 --pmc, own pass) for the plain GEMM shapes (with hipBLASLt beside) and one vocoder pass.
set -o pipefail
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA_BUSY|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES" | head -8
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_g -o g -- python3 tools/probes/gemm_vs_blaslt.py 3 > gpurun_out/pmc_mfma_g.log 2>&1 || { tail -20 gpurun_out/pmc_mfma_g.log; exit 1; }
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma_v -o v -- python3 tools/vocoder_time.py > gpurun_out/pmc_mfma_v.log 2>&1 || { tail -20 gpurun_out/pmc_mfma_v.log; exit 1; }
python3 - <<'PY' | tee gpurun_out/${R:-r03}_pmc_mfma_busy.txt
import csv, glob, collections
for tag in ("g", "v"):
    files = glob.glob(f"gpurun_out/pmc_mfma_{tag}/**/*counter_collection.csv", recursive=True)
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for f in files:
        rows = list(csv.DictReader(open(f)))
        by = collections.defaultdict(dict)
        for r in rows:
            by[(r["Dispatch_Id"], r["Kernel_Name"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (d, k), c in by.items():
            a = agg[k]
            a[0] += 1
            a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
    print(f"== {'plain GEMM shapes (itts vs hipBLASLt)' if tag == 'g' else 'BigVGAN.forward [32,140,1280] fp16'}: kernel | dispatches | MFMA busy cycles (all SIMDs) / (GPU-active cycles x 1024 SIMDs)")
    for k, (n, m, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        if m > 0:
            print(f"{k[:110]:110s} {n:5d}  {m / (g * 1024) if g else 0:6.3f}")
PY
rm -rf gpurun_out/pmc_mfma_g/*/*.db gpurun_out/pmc_mfma_v/*/*.db
echo ALLDONE
