#!/usr/bin/env python3
"""In-kernel timeline of gemm_conv_kernel on one vocoder shape (diagnostic build): per-workgroup segments, workgroup
lifetime, how many workgroups a CU runs at once, and how the grid drains.  Usage: timeline_conv.py C taps dil T [B]"""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("ITTS_HIP_LIB", os.path.join(ROOT, "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

C, taps, dil, T = (int(v) for v in sys.argv[1:5])
B = int(sys.argv[5]) if len(sys.argv) > 5 else 32
L = nat.lib()
L.itts_debug_stamps_conv.restype = ctypes.c_int
L.itts_debug_stamps_conv.argtypes = [ctypes.c_void_p]
dev, dt = "cuda", torch.float16
x = torch.randn(B, T, C, device=dev).to(dt)
w = (torch.randn(taps, C, C, device=dev) * 0.02).to(dt)
wp = nat.pack_weight(w)
y = torch.zeros(B, T, C, device=dev, dtype=dt)
bias = torch.zeros(C, device=dev)
pad = (taps * dil - dil) // 2
NW = 1 << 16
stamps = torch.zeros(NW, 16, dtype=torch.int64, device=dev)


if os.environ.get("ITTS_CONV_EXP"):
    nat.debug_set(5, int(os.environ["ITTS_CONV_EXP"]))
if os.environ.get("ITTS_CONV_CFG"):
    nat.debug_set(3, int(os.environ["ITTS_CONV_CFG"]))


def run():
    nat.gemm_conv(dt, B, T, T, C, C, wp, x, y, taps=taps, off0=-pad, dil=dil, bias=bias)


for _ in range(3):
    run()
L.itts_debug_stamps_conv(ctypes.c_void_p(stamps.data_ptr()))
run()
torch.cuda.synchronize()
L.itts_debug_stamps_conv(None)
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, 15] > 0]
n = s.shape[0]
clk = np.median((s[:, 5] - s[:, 0]) / np.maximum(s[:, 15] - s[:, 14], 1.0)) * 100.0
seg = ["issue first prefetch", "first chunk staged (HBM latency + LDS commit)", "first tile: chunk loop (LDS reads, weight loads, MFMA)",
       "first tile: epilogue issue", "remaining tiles + store drain"]
out = {"shape": dict(C=C, taps=taps, dil=dil, T=T, B=B), "workgroups": n, "clock_mhz": round(float(clk))}
out["segments_us"] = {name: round(float(np.median((s[:, i + 1] - s[:, i]) / clk)), 2) for i, name in enumerate(seg)}
def med(a, b):
    ok = (s[:, a] > 0) & (s[:, b] > 0)
    return round(float(np.median((s[ok, a] - s[ok, b]) / clk)), 2) if ok.any() else None


out["chunk1_us"] = {"barrier + LDS commit + barrier": med(8, 6), "prefetch issue + taps 0-1": med(9, 8), "taps 2-3": med(10, 9),
                    "taps 4-5": med(11, 10), "whole chunk": med(7, 6)}
out["wg_life_us"] = {"median": round(float(np.median((s[:, 5] - s[:, 0]) / clk)), 2),
                     "p90": round(float(np.percentile((s[:, 5] - s[:, 0]) / clk, 90)), 2)}
t0 = s[:, 14].min()
out["kernel_span_us"] = round(float((s[:, 15].max() - t0) / 100.0), 1)
# residency: workgroups alive at the kernel's midpoint, per CU (HW_ID: cu_id bits 8-11, sh 12, se 13-15 on gfx9; xcc separate)
mid = t0 + (s[:, 15].max() - t0) / 2
alive = s[(s[:, 14] <= mid) & (s[:, 15] >= mid)]
hw = alive[:, 12].astype(np.int64)
cu = ((alive[:, 13].astype(np.int64)) << 16) | (hw & 0xFF00)
out["alive_at_midpoint"] = int(alive.shape[0])
out["distinct_cus_at_midpoint"] = int(len(set(cu.tolist())))
out["start_times_us_percentiles"] = [round(float(np.percentile((s[:, 14] - t0) / 100.0, q)), 1) for q in (0, 10, 50, 90, 100)]
print(json.dumps(out, indent=1))
