#!/usr/bin/env python3
"""How much of a decode GEMM launch is the HBM fetch?  Each of the four decode GEMMs is replayed over 1 / 2 / 4 / 24
rotating weight sets: 1 set stays in the XCDs' L2s (if the workgroup -> XCD placement repeats from launch to launch),
4 sets (13-52 MB) stay in the Infinity Cache, 24 sets (80-315 MB) are cold HBM reads -- the token loop's case.
A second pass puts an unrelated small kernel (ln_reduce) between the launches, as the token loop does.
Writes gpurun_out/probe_warm.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev, T = "cuda", torch.bfloat16
B, D, H, L = 32, 1280, 20, 24
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "probe_warm.txt"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


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
f = torch.randn(Bp, 4 * D, device=dev).to(T)
q = torch.randn(B, D, device=dev).to(T)
a = torch.randn(Bp, D, device=dev).to(T)
h = torch.randn(B, D, device=dev)
slab = torch.randn(4, B, D, device=dev)
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
b3, b4 = torch.zeros(3 * D, device=dev), torch.zeros(4 * D, device=dev)
smax = 320
kc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
vc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
w_qkv = [rand_w(D, 3 * D) for _ in range(L)]
w_o = [rand_w(D, D) for _ in range(L)]
w_fc = [rand_w(D, 4 * D) for _ in range(L)]
w_pr = [rand_w(4 * D, D) for _ in range(L)]
exps = {
    "QKV  9.8MB": lambda i: nat.gemm_skinny(T, B, 3 * D, D, w_qkv[i], b3, x=xn, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc[i],
                                            vcache=vc[i], pos=pos, heads=H, smax=smax, x_packed=True),
    "proj 3.3MB": lambda i: nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True),
    "FC  13.1MB": lambda i: nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], b4, x=xn, epi=nat.EPI_GELU_STORE, y=f, x_packed=True,
                                            y_packed=True),
    "FC2 13.1MB": lambda i: nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3,
                                            x_packed=True),
}
N = 96
log("==== probe_warm_weights: us per launch inside a replayed graph of 96 launches")
lnr = timed_graph(lambda: [nat.ln_reduce(h, lw, lb, xn, slab=slab[:3], nslab=3, bias=lw, y_packed=True) for _ in range(N)], N)
log(f"ln_reduce alone: {lnr:.2f} us")
for name, one in exps.items():
    for nsets in (1, 2, 4, 24):
        def fn():
            for j in range(N):
                one(j % nsets)
        t = timed_graph(fn, N)

        def fn2():
            for j in range(N):
                one(j % nsets)
                nat.ln_reduce(h, lw, lb, xn, slab=slab[:3], nslab=3, bias=lw, y_packed=True)
        t2 = timed_graph(fn2, N)
        log(f"{name} sets={nsets:2d}: {t:6.2f} us | with ln_reduce between: pair {t2:6.2f} us (gemm share {t2 - lnr:6.2f})")
