#!/usr/bin/env python3
"""Sweep waves per workgroup (itts_debug_set key 2) for the four decode GEMMs; us per launch inside a replayed graph."""
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
B, D, H, L, R = 32, 1280, 20, 24, 4


def rand_w(K, N):
    return nat.pack_weight((torch.randn(K, N, device=dev) * 0.02).to(T))


def timed_graph(fn, n, replays=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(replays):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / (replays * n)


state = torch.zeros(8, dtype=torch.int32, device=dev)
state[1] = 150
pos = state[1:2]
xn = torch.randn(B, D, device=dev).to(T)
f = torch.randn(B, 4 * D, device=dev).to(T)
q = torch.randn(B, D, device=dev).to(T)
a = torch.randn(B, D, device=dev).to(T)
slab = torch.randn(4, B, D, device=dev)
b3, b4 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev)
smax = 320
kc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
vc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
w_qkv = [rand_w(D, 3 * D) for _ in range(L)]
w_o = [rand_w(D, D) for _ in range(L)]
w_fc = [rand_w(D, 4 * D) for _ in range(L)]
w_pr = [rand_w(4 * D, D) for _ in range(L)]
exps = {
    "QKV": lambda i: nat.gemm_skinny(T, B, 3 * D, D, w_qkv[i], b3, x=xn, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc[i], vcache=vc[i],
                                     pos=pos, heads=H, smax=smax),
    "proj ks3": lambda i: nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3),
    "FC": lambda i: nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], b4, x=xn, epi=nat.EPI_GELU_STORE, y=f),
    "FC2 ks3": lambda i: nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3),
}
for name, one in exps.items():
    for nw in (0, 4, 6, 8, 10, 12, 16):
        nat.debug_set(2, nw)

        def fn():
            for _ in range(R):
                for i in range(L):
                    one(i)
        print(f"{name:10s} waves={nw or 'auto':>4}: {timed_graph(fn, R * L):6.2f} us", flush=True)
nat.debug_set(2, 0)
