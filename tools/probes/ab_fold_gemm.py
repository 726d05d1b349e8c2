#!/usr/bin/env python3
"""Microseconds per launch of the decode GEMMs in both forms of the step, same box, same process: the 7-launch form's GEMMs
(plain QKV / FC, split-K out-projection / FC2) and the LayerNorm-folded form's (QKV' / FC' with the statistics on the matrix
pipe, out-projection' / FC2' with the residual epilogue, per launch geometry) -- inside replayed graphs over 24 rotating
weight sets (cold HBM reads, the token loop's case), 96 launches per graph."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev, T = "cuda", torch.bfloat16
B, D, H, L = int(os.environ.get("ITTS_ROWS", "32")), 1280, 20, 24


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
Bp = nat.packed_rows(B)
xn = torch.randn(Bp, D, device=dev).to(T)
hb = torch.randn(Bp, D, device=dev).to(T)
h = torch.randn(B, D, device=dev)
f = torch.randn(Bp, 4 * D, device=dev).to(T)
q = torch.randn(B, D, device=dev).to(T)
a = torch.randn(Bp, D, device=dev).to(T)
slab = torch.randn(4, B, D, device=dev)
b1, b3, b4 = torch.zeros(D, device=dev), torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev)
c3, c4 = torch.randn(3 * D, device=dev), torch.randn(4 * D, device=dev)
smax = 320
kc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
vc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
w_qkv = [rand_w(D, 3 * D) for _ in range(L)]
w_o = [rand_w(D, D) for _ in range(L)]
w_fc = [rand_w(D, 4 * D) for _ in range(L)]
w_pr = [rand_w(4 * D, D) for _ in range(L)]
exps = {
    "QKV": lambda i: nat.gemm_skinny(T, B, 3 * D, D, w_qkv[i], b3, x=xn, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc[i], vcache=vc[i],
                                     pos=pos, heads=H, smax=smax, x_packed=True),
    "QKV'": lambda i: nat.gemm_skinny(T, B, 3 * D, D, w_qkv[i], b3, x=hb, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc[i], vcache=vc[i],
                                      pos=pos, heads=H, smax=smax, x_packed=True, ln_c=c3),
    "FC": lambda i: nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], b4, x=xn, epi=nat.EPI_GELU_STORE, y=f, x_packed=True, y_packed=True),
    "FC'": lambda i: nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], b4, x=hb, epi=nat.EPI_GELU_STORE, y=f, x_packed=True, y_packed=True, ln_c=c4),
    "proj ks3": lambda i: nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True),
    "FC2 ks3": lambda i: nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True),
}
for rows in (0, 16):
    for wide in (False, True):
        exps[f"proj' r{rows}{'w' if wide else ''}"] = (lambda i, rows=rows, wide=wide: nat.gemm_skinny(
            T, B, D, D, w_o[i], b1, x=a, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=True, y_packed=True, rows_per_wg=rows, wide_wg=wide))
        exps[f"FC2' r{rows}{'w' if wide else ''}"] = (lambda i, rows=rows, wide=wide: nat.gemm_skinny(
            T, B, D, 4 * D, w_pr[i], b1, x=f, epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=True, y_packed=True, rows_per_wg=rows, wide_wg=wide))
N = 96
ABL = os.environ.get("ITTS_ABLATE")          # diagnostic build: "1" activations pinned to k-step 0, "2" weights pinned, "3" both
if ABL:
    nat.debug_set(6, int(ABL))
for rep in range(2):
    out = []
    for name, one in exps.items():
        out.append(f"{name} {timed_graph(lambda: [one(j % L) for j in range(N)], N):.2f}")
    print("us per launch (24 weight sets): " + " | ".join(out), flush=True)
