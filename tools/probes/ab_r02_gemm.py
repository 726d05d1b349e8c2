#!/usr/bin/env python3
"""Same-box A/B of the decode GEMMs: the round-2 library (tools/probes/build/libindextts_hip_r02.so, built from commit
58b4230) against the current one -- microseconds per launch inside a replayed graph over 24 rotating weight sets (cold HBM
reads, the token loop's case).  The old library is loaded through a proxy that answers the ABI version check (the argument
struct only GREW at its end since then)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) == 1:
    bd = os.path.join(ROOT, "tools", "probes", "build")
    libs = [("r02", os.path.join(bd, "libindextts_hip_r02.so")), ("now", "")]
    libs += [(os.path.basename(f)[len("libindextts_hip_"):-3], os.path.join(bd, f)) for f in sorted(os.listdir(bd))
             if f.startswith("libindextts_hip_") and f.endswith(".so") and "r02" not in f]
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
            if n == "itts_abi_version" or not hasattr(self._l, n):   # entry points the old library does not have yet
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


state = torch.zeros(8, dtype=torch.int32, device=dev)
state[1] = 150
pos = state[1:2]
Bp = nat.packed_rows(B)
xn = torch.randn(Bp, D, device=dev).to(T)
f = torch.randn(Bp, 4 * D, device=dev).to(T)
q = torch.randn(B, D, device=dev).to(T)
a = torch.randn(Bp, D, device=dev).to(T)
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
                                     pos=pos, heads=H, smax=smax, x_packed=True),
    "proj": lambda i: nat.gemm_skinny(T, B, D, D, w_o[i], None, x=a, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True),
    "FC": lambda i: nat.gemm_skinny(T, B, 4 * D, D, w_fc[i], b4, x=xn, epi=nat.EPI_GELU_STORE, y=f, x_packed=True, y_packed=True),
    "FC2": lambda i: nat.gemm_skinny(T, B, D, 4 * D, w_pr[i], None, x=f, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=3, x_packed=True),
}
N = 96
out = []
for name, one in exps.items():
    out.append(f"{name} {timed_graph(lambda: [one(j % L) for j in range(N)], N):.2f}")
allf = lambda: [fn(j % L) for j in range(24) for fn in exps.values()]   # noqa: E731
out.append(f"block-order {timed_graph(allf, 96):.2f}")
print("   us per launch (24 weight sets): " + " | ".join(out), flush=True)
