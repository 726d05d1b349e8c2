#!/usr/bin/env python3
"""Decode attention: 4 vs 8 waves per (row, head) workgroup (itts_debug_set key 4); us per launch in a replayed graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import os as _os
_os.environ.setdefault("ITTS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))  # tuning knobs live in the diagnostic build
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev, T = "cuda", torch.bfloat16
B, D, H, L, R, smax = 32, 1280, 20, 24, 4, 320
q = torch.randn(B, D, device=dev).to(T)
a = torch.empty(B, D, device=dev, dtype=T)
kc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
vc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
pad = torch.zeros(B, dtype=torch.int32, device=dev)
state = torch.zeros(8, dtype=torch.int32, device=dev)
for ctx in (100, 165, 236):
    state[1] = ctx - 1
    for nw in (4, 8):
        nat.debug_set(4, nw)

        def fn():
            for _ in range(R):
                for i in range(L):
                    nat.attn_decode(q, kc[i], vc[i], a, pad, state[1:2], B, H, smax)
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / (20 * R * L)
        print(f"ctx {ctx:3d} waves {nw}: {us:6.2f} us  ({B * ctx * 2 * D * 2 / us / 1e6:5.2f} TB/s)", flush=True)
nat.debug_set(4, 4)
