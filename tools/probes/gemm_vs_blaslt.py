#!/usr/bin/env python3
"""The prefill / latent-pass GEMM shapes: itts_gemm_conv (taps = 1) against torch.matmul (hipBLASLt) on the same box.
A yardstick only -- the product never calls the library GEMM."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev, bf = "cuda", torch.bfloat16
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def timed(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


for M in (3008, 4480):
    for name, N, K in (("QKV", 3840, 1280), ("proj", 1280, 1280), ("FC", 5120, 1280), ("FC2", 1280, 5120)):
        x = torch.randn(1, M, K, device=dev).to(bf)
        w = (torch.randn(1, K, N, device=dev) * 0.02).to(bf)
        wp = nat.pack_weight(w)
        y = torch.zeros(1, M, N, device=dev, dtype=bf)
        bias = torch.zeros(N, device=dev)
        us_i = timed(lambda: nat.gemm_conv(bf, 1, M, M, K, N, wp, x, y, taps=1, off0=0, dil=1, bias=bias))
        x2, w2, b2 = x[0], w[0].contiguous(), bias.to(bf)
        us_t = timed(lambda: torch.addmm(b2, x2, w2))
        wt = w[0].t().contiguous()
        us_tt = timed(lambda: torch.nn.functional.linear(x2, wt, b2))
        fl = 2.0 * M * N * K
        print(f"M={M} {name:5s} N={N} K={K}: itts {us_i:7.1f} us {fl / us_i / 1e6:7.1f} TF/s | addmm {us_t:7.1f} us {fl / us_t / 1e6:7.1f} | "
              f"linear(NT) {us_tt:7.1f} us {fl / us_tt / 1e6:7.1f}", flush=True)
