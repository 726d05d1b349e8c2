#!/usr/bin/env python3
"""Slot time of each decode-step kernel inside a replayed CUDA graph (MI355X; writes gpurun_out/microbench.txt).

Each experiment captures `reps` launches cycling through 24 layers' worth of distinct weights (so the weight stream is
real HBM traffic, not L2 hits), replays the graph a few times and reports microseconds per launch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import os as _os
_os.environ.setdefault("ITTS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))  # tuning knobs live in the diagnostic build
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev = "cuda"
T = torch.bfloat16
B, D, H, L = 32, 1280, 20, 24
out = open(os.path.join(ROOT, "gpurun_out", "microbench.txt"), "a")


def log(*a):
    s = " ".join(str(x) for x in a)
    print(s, flush=True)
    out.write(s + "\n")
    out.flush()


def timed_graph(fn, reps_in_graph, replays=20):
    fn()  # warm-up (loads code objects)
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
    return 1e3 * e0.elapsed_time(e1) / (replays * reps_in_graph)


def rand_w(K, N):
    return nat.pack_weight((torch.randn(K, N, device=dev) * 0.02).to(T))


log(f"==== microbench {time.strftime('%H:%M:%S')} B={B}")
tokens = torch.zeros(B, dtype=torch.int32, device=dev)
state = torch.zeros(8, dtype=torch.int32, device=dev)
state[1] = 150
table = torch.randn(8194, D, device=dev)
ptab = torch.randn(803, D, device=dev)
h = torch.randn(B, D, device=dev)
xn = torch.randn(B, D, device=dev).to(T)
f = torch.randn(B, 4 * D, device=dev).to(T)
q = torch.randn(B, D, device=dev).to(T)
a = torch.randn(B, D, device=dev).to(T)
slab = torch.randn(4, B, D, device=dev)
lw, lb = torch.ones(D, device=dev), torch.zeros(D, device=dev)
bias_d = torch.zeros(D, device=dev)
bias_3d = torch.zeros(3 * D, device=dev)
bias_4d = torch.zeros(4 * D, device=dev)
smax = 320
kc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
vc = torch.randn(L, B, H, smax, 64, device=dev).to(T)
pad = torch.zeros(B, dtype=torch.int32, device=dev)
pos = state[1:2]
w_qkv = [rand_w(D, 3 * D) for _ in range(L)]
w_o = [rand_w(D, D) for _ in range(L)]
w_fc = [rand_w(D, 4 * D) for _ in range(L)]
w_pr = [rand_w(4 * D, D) for _ in range(L)]

R = 4  # passes over the 24 layers per graph


def exp_embed():
    for _ in range(R * L):
        nat.embed_step(tokens, table, ptab, state[0:1], 1, h)


def exp_ln0():
    for _ in range(R * L):
        nat.ln_reduce(h, lw, lb, xn)


def exp_ln4():
    for _ in range(R * L):
        nat.ln_reduce(h, lw, lb, xn, slab=slab, nslab=4, bias=bias_d)


def exp_qkv():
    for _ in range(R):
        for i in range(L):
            nat.gemm_skinny(T, B, 3 * D, D, w_qkv[i], bias_3d, x=xn, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc[i], vcache=vc[i],
                            pos=pos, heads=H, smax=smax)


def exp_attn():
    for _ in range(R):
        for i in range(L):
            nat.attn_decode(q, kc[i], vc[i], a, pad, pos, B, H, smax)


def exp_proj(ks):
    def fn():
        for _ in range(R):
            for i in range(L):
                nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=ks)
    return fn


def exp_fc():
    for _ in range(R):
        for i in range(L):
            nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], bias_4d, x=xn, epi=nat.EPI_GELU_STORE, y=f)


def exp_fc2(ks):
    def fn():
        for _ in range(R):
            for i in range(L):
                nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=ks)
    return fn


MB = 1e-6
for name, fn, bytes_ in [
    ("embed_step (trivial floor)", exp_embed, 0),
    ("ln_reduce nslab=0", exp_ln0, 0),
    ("ln_reduce nslab=4", exp_ln4, 0),
    ("skinny QKV 1280x3840", exp_qkv, D * 3 * D * 2),
    ("attn_decode ctx=151", exp_attn, B * 151 * 2 * D * 2),
    ("skinny proj ksplit=4", exp_proj(4), D * D * 2),
    ("skinny proj ksplit=2", exp_proj(2), D * D * 2),
    ("skinny FC 1280x5120", exp_fc, D * 4 * D * 2),
    ("skinny FC2 ksplit=4", exp_fc2(4), D * 4 * D * 2),
    ("skinny FC2 ksplit=2", exp_fc2(2), D * 4 * D * 2),
]:
    us = timed_graph(fn, R * L)
    extra = f"  {bytes_ * MB:6.1f} MB -> {bytes_ / us / 1e6:6.2f} TB/s" if bytes_ else ""
    log(f"{name:32s} {us:7.2f} us/launch{extra}")

# ---- balance experiment: column tiles vs the 256 CUs
for N in (3840, 4096, 4112, 5120, 6144, 8192):
    ws = [rand_w(D, N) for _ in range(L)]
    yb = torch.zeros(B, N, device=dev, dtype=T)
    bb = torch.zeros(N, device=dev)

    def fn():
        for _ in range(R):
            for i in range(L):
                nat.gemm_skinny(T, B, N, D, ws[i], bb, x=xn, epi=nat.EPI_GELU_STORE, y=yb)
    us = timed_graph(fn, R * L)
    log(f"skinny K=1280 N={N:5d} ({N // 16:3d} tiles) {us:7.2f} us  {D * N * 2 / us / 1e6:5.2f} TB/s")
    del ws

# ---- sweep (ksplit, tiles per workgroup, waves) for the two N=1280 GEMMs and FC
slab8 = torch.randn(8, B, D, device=dev)
for name, K, ws, xin in (("proj", D, w_o, a), ("FC2", 4 * D, w_pr, f)):
    for ks in (2, 3, 4, 6, 8):
        for ntb in (1, 2):
            def fn(ks=ks, ws=ws, xin=xin, K=K):
                for _ in range(R):
                    for i in range(L):
                        nat.gemm_skinny(T, B, D, K, ws[i], None, x=xin, epi=nat.EPI_SLAB_F32, yf=slab8, ksplit=ks)
            nat.debug_set(1, ntb)
            us = timed_graph(fn, R * L)
            log(f"{name} ksplit={ks} ntb={ntb}: {us:6.2f} us  (blocks {((80 + ntb - 1) // ntb) * ks})")
nat.debug_set(1, 0)
for ntb in (1, 2, 3):
    for nw in (4, 8):
        nat.debug_set(1, ntb)
        nat.debug_set(2, nw)
        us = timed_graph(exp_fc, R * L)
        log(f"FC ntb={ntb} nw={nw}: {us:6.2f} us")
nat.debug_set(1, 0)
nat.debug_set(2, 0)
for ns in (3, 6, 8):
    def fn(ns=ns):
        for _ in range(R * L):
            nat.ln_reduce(h, lw, lb, xn, slab=slab8, nslab=ns, bias=bias_d)
    log(f"ln_reduce nslab={ns}: {timed_graph(fn, R * L):6.2f} us")
