#!/usr/bin/env python3
"""Per-launch HBM traffic of gemm_skinny_kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB units).
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-B requests of a wide coalesced stream at 64 B, so
the read side is doubled.  Usage: pmc_summary.py <fetch_dir> <write_dir> [algorithmic bytes per launch] [out.json]"""
import csv
import glob
import json
import sys


def mean_counter(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f))
            if r["Counter_Name"] == counter and "gemm_skinny_kernel" in r["Kernel_Name"]]
    return sum(vals) / len(vals), len(vals)


fs, n1 = mean_counter(sys.argv[1], "FETCH_SIZE")
ws, n2 = mean_counter(sys.argv[2], "WRITE_SIZE")
read_b, write_b = 2.0 * fs * 1024.0, ws * 1024.0
print(f"gemm_skinny_kernel: {n1} / {n2} dispatches; FETCH_SIZE {fs:.1f} KiB (x2 gfx950 correction -> {read_b / 1e6:.3f} MB), "
      f"WRITE_SIZE {ws:.1f} KiB ({write_b / 1e6:.3f} MB); traffic per launch {(read_b + write_b) / 1e6:.3f} MB")
if len(sys.argv) > 4:
    out = {"kernel": "gemm_skinny_kernel", "launches": n1, "FETCH_SIZE_KiB_mean": round(fs, 1), "WRITE_SIZE_KiB_mean": round(ws, 1),
           "read_correction": "x2 (gfx950: 128-B requests counted at 64 B, MI355X_MICROARCH.md HBM section)",
           "traffic_bytes_per_launch": int(read_b + write_b), "algorithmic_bytes_per_launch": int(float(sys.argv[3])),
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (separate passes) --output-format csv -- python3 tools/pmc_decode_gemm.py"}
    json.dump(out, open(sys.argv[4], "w"), indent=1)
