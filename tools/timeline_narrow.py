#!/usr/bin/env python3
"""In-kernel timeline of conv_narrow_kernel (diagnostic build).  Usage: timeline_narrow.py C taps dil T [B]"""
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
y = torch.zeros(B, T, C, dtype=dt, device=dev)
bias = torch.zeros(C, device=dev)
pad = (taps * dil - dil) // 2
stamps = torch.zeros(1 << 14, 16, dtype=torch.int64, device=dev)


def run():
    nat.gemm_conv(dt, B, T, T, C, C, wp, x, y, taps=taps, off0=-pad, dil=dil, bias=bias)


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
L.itts_debug_stamps_conv(ctypes.c_void_p(stamps.data_ptr()))
run()
torch.cuda.synchronize()
L.itts_debug_stamps_conv(None)
s = stamps.cpu().numpy().astype(np.float64)
s = s[s[:, 14] > 0]
clk = 100.0 * 21.0   # s_memtime runs at the shader clock on this part (about 2.1 GHz under this load); order of magnitude only


def med(a, b):
    ok = (s[:, a] > 0) & (s[:, b] > 0)
    return round(float(np.median((s[ok, a] - s[ok, b]) / clk)), 2) if ok.any() else None


out = {"shape": dict(C=C, taps=taps, dil=dil, T=T, B=B), "us_per_launch": round(1e3 * e0.elapsed_time(e1) / 10, 1),
       "workgroups": int(s.shape[0]), "tiles_per_wave_median": float(np.median(s[:, 9])),
       "us": {"weights staged": med(1, 0), "tile 0: taps (first load latency + 7 taps)": med(3, 2), "tile 0: epilogue issue": med(4, 3),
              "tile 0 end -> tile 1 start": med(6, 4), "tile 1: taps": med(7, 6), "tile 1: epilogue issue": med(8, 7),
              "workgroup life": med(5, 0)}}
out["us"]["life / tiles"] = round(out["us"]["workgroup life"] / max(out["tiles_per_wave_median"], 1), 2)
print(json.dumps(out, indent=1))
