#!/usr/bin/env python3
"""In-kernel timeline of itts_sample (diagnostic build): where the per-token selection kernel spends its microseconds.
Stamps (include/indextts_hip_diag.h): 0 entry | 1 logits + penalty bitmap | 2 processed scores in LDS | 3 top-k threshold |
4 candidates compacted | 5 rank sort | 6 token drawn | 7 bookkeeping done.  Median over rows and launches, microseconds."""
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

L = nat.lib()
L.itts_debug_stamps_sample.restype = ctypes.c_int
L.itts_debug_stamps_sample.argtypes = [ctypes.c_void_p]
dev = "cuda"
B, V = 32, 8194
g = torch.Generator().manual_seed(0)
logits = (torch.randn(B, V, generator=g) * 3).to(dev)
tokens = torch.zeros(B, dtype=torch.int32, device=dev)
history = torch.randint(0, 8192, (B, 2048), generator=g).to(torch.int32).to(dev)
finished = torch.zeros(B, dtype=torch.int32, device=dev)
state = torch.zeros(8, dtype=torch.int32, device=dev)
state[0] = 70
extra = torch.tensor([1, 8192], dtype=torch.int32, device=dev)
res = {}
for lazy in (False, True):
    rows = []
    for it in range(20):
        stamps = torch.zeros(B, 16, dtype=torch.int64, device=dev)
        L.itts_debug_stamps_sample(ctypes.c_void_p(stamps.data_ptr()))
        state[0] = 70
        nat.sample(logits, tokens, history, finished, state, extra, None, 10.0, 1.0, 30, 0.8, True, 1234, 8193, no_advance=lazy)
        torch.cuda.synchronize()
        if it >= 5:
            rows.append(stamps.cpu().numpy().astype(np.float64))
    L.itts_debug_stamps_sample(None)
    s = np.concatenate(rows, 0)
    clk = np.median((s[:, 7] - s[:, 0]) / np.maximum(s[:, 15] - s[:, 14], 1.0)) * 100.0
    names = ["logits + bitmap", "processed scores -> LDS", "top-k threshold", "compaction", "rank sort", "softmax/top-p/draw", "bookkeeping"]
    seg = {n: round(float(np.median((s[:, i + 1] - s[:, i]) / clk)), 2) for i, n in enumerate(names)}
    seg["total_in_kernel"] = round(float(np.median((s[:, 7] - s[:, 0]) / clk)), 2)
    res["no_advance" if lazy else "advance_in_kernel"] = seg
print(json.dumps(res, indent=1))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write(json.dumps(res, indent=1) + "\n")
