#!/usr/bin/env python3
"""Diagnostic build: the 256 x 128 plain-GEMM kernel (itts_debug_set(3, 9)) against the product's 128 x 128 one: same bits
(same k order per output), and the time of both on the prefill / latent shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("ITTS_HIP_LIB", os.path.join(ROOT, "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev = "cuda"


def run(cfg, dtype, M, K, N, y_f32, resid, act, reps=0):
    nat.debug_set(3, cfg)
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    x = (torch.randn(1, M, K, generator=g)).to(dtype).to(dev)
    w = (torch.randn(1, K, N, generator=g) * 0.02).to(dtype).to(dev)
    wp = nat.pack_weight(w)
    y0 = torch.randn(1, M, N, generator=g).to(torch.float32 if y_f32 else dtype).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    y = y0.clone()
    call = lambda: nat.gemm_conv(dtype, 1, M, M, K, N, wp, x, y, taps=1, off0=0, dil=1, bias=bias, y_f32=y_f32,  # noqa: E731
                                 resid=y if resid else None, act=act)
    call()
    out = y.clone()
    us = 0.0
    if reps:
        y.copy_(y0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / reps
    return out, us


bf = torch.bfloat16
if os.environ.get("ITTS_BIG_EXP"):
    # ablations of the 256 x 128 kernel (results are wrong by construction; only the time is of interest)
    for ex in [int(v) for v in os.environ["ITTS_BIG_EXP"].split(",")]:
        nat.debug_set(5, ex)
        for cfg in (9, 11):
            _, us = run(cfg, bf, 4544, 1280, 5120, False, False, 1, reps=20)
            print(f"exp {ex:5d} cfg {cfg}: {us:7.1f} us", flush=True)
    nat.debug_set(5, 0)
    sys.exit(0)
for dtype, M, K, N, y_f32, resid, act in [(bf, 2016, 1280, 3840, False, False, 0), (bf, 2016, 1280, 1280, True, True, 0),
                                           (bf, 2016, 1280, 5120, False, False, 1), (bf, 2016, 5120, 1280, True, True, 0),
                                           (bf, 4544, 1280, 3840, False, False, 0), (bf, 4544, 1280, 1280, True, True, 0),
                                           (bf, 4544, 1280, 5120, False, False, 1), (bf, 4544, 5120, 1280, True, True, 0),
                                           (bf, 333, 1280, 1280, False, False, 0), (torch.float32, 700, 1280, 3840, False, False, 0),
                                           (torch.float16, 513, 96, 128, False, False, 0)]:
    a, ua = run(0, dtype, M, K, N, y_f32, resid, act, reps=20)
    fl = 2.0 * M * N * K
    line = f"{str(dtype)[6:]:9s} M={M:5d} K={K:5d} N={N:5d} f32out={int(y_f32)} resid={int(resid)} act={act}: 128x128 {ua:6.1f} us {fl / ua / 1e6:6.1f} TF/s"
    for cfg in (9, 11):
        b, ub = run(cfg, dtype, M, K, N, y_f32, resid, act, reps=20)
        line += f" | cfg{cfg} {'==' if torch.equal(a, b) else '!='} {ub:6.1f} us {fl / ub / 1e6:6.1f}"
    print(line, flush=True)
