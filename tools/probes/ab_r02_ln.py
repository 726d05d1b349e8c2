#!/usr/bin/env python3
"""Same-box A/B of the residual-reduce + LayerNorm launch inside its real neighbourhood: [split-K GEMM -> ln_reduce] pairs
over 24 rotating weight sets in a replayed graph (the slabs the reduce reads were just written by the GEMM, on all XCDs), the
round-2 library against the current one.  See ab_r02_gemm.py for the loading trick."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    bd = os.path.join(ROOT, "tools", "probes", "build")
    libs = [("r02", os.path.join(bd, "libindextts_hip_r02.so")), ("now", "")]
    for tag, lib in libs + libs:
        env = dict(os.environ)
        if lib:
            env["ITTS_HIP_LIB"] = lib
        print(f"== {tag}", flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=env, check=True)
    sys.exit(0)

for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import ctypes  # noqa: E402

import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

if os.environ.get("ITTS_HIP_LIB"):
    real = ctypes.CDLL

    class _Fn:
        restype = argtypes = None

        def __call__(self):
            return 6

    class Proxy:
        def __init__(self, path):
            self._l = real(path)

        def __getattr__(self, n):
            if n == "itts_abi_version" or not hasattr(self._l, n):
                return _Fn()
            return getattr(self._l, n)
    nat.C.CDLL = Proxy

dev, T = "cuda", torch.bfloat16
B, D, H, L = 32, 1280, 20, 24


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


Bp = nat.packed_rows(B)
xn = torch.randn(Bp, D, device=dev).to(T)
f = torch.randn(Bp, 4 * D, device=dev).to(T)
a = torch.randn(Bp, D, device=dev).to(T)
h = torch.randn(B, D, device=dev)
slab = torch.randn(3, B, D, device=dev)
w_o = [rand_w(D, D) for _ in range(L)]
w_pr = [rand_w(4 * D, D) for _ in range(L)]
lnw = [(torch.randn(D, device=dev), torch.randn(D, device=dev), torch.randn(D, device=dev)) for _ in range(2 * L)]


def pair_o(i):
    nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True)
    w, b, bias = lnw[2 * i]
    nat.ln_reduce(h, w, b, xn, slab=slab, nslab=3, bias=bias, y_packed=True)


def pair_p(i):
    nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True)
    w, b, bias = lnw[2 * i + 1]
    nat.ln_reduce(h, w, b, xn, slab=slab, nslab=3, bias=bias, y_packed=True)


N = 48
po = timed_graph(lambda: [pair_o(j % L) for j in range(N)], N)
pp = timed_graph(lambda: [pair_p(j % L) for j in range(N)], N)
go = timed_graph(lambda: [nat.gemm_skinny(T, B, D, D, w_o[j % L], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True) for j in range(N)], N)
gp = timed_graph(lambda: [nat.gemm_skinny(T, B, D, 4 * D, w_pr[j % L], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True) for j in range(N)], N)
print(f"   us: [proj + ln] {po:.2f} (proj alone {go:.2f} -> ln {po - go:.2f}) | [FC2 + ln] {pp:.2f} (FC2 alone {gp:.2f} -> ln {pp - gp:.2f})", flush=True)
