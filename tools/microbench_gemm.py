#!/usr/bin/env python3
"""Time gemm_conv on the shapes of the latent pass / vocoder (MI355X).  Usage: microbench_gemm.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import os as _os
_os.environ.setdefault("ITTS_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))  # tuning knobs live in the diagnostic build
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402

dev = "cuda"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only = sys.argv[2] if len(sys.argv) > 2 else ""


def bench(name, dtype, B, T, Cin, N, taps, dil, y_f32=False, resid=False, act=0):
    if only and only not in name:
        return
    x = (torch.randn(B, T, Cin, device=dev)).to(dtype)
    w = (torch.randn(taps, Cin, N, device=dev) * 0.02).to(dtype)
    wp = nat.pack_weight(w)
    y = torch.zeros(B, T, N, device=dev, dtype=torch.float32 if y_f32 else dtype)
    bias = torch.zeros(N, device=dev)
    pad = (taps * dil - dil) // 2

    def run():
        nat.gemm_conv(dtype, B, T, T, Cin, N, wp, x, y, taps=taps, off0=-pad, dil=dil, bias=bias, y_f32=y_f32,
                      resid=y if resid else None, act=act)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    fl = 2.0 * B * T * N * Cin * taps
    print(f"{name:34s} {us:9.1f} us  {fl / us / 1e6:8.1f} TFLOP/s", flush=True)


bf, fh = torch.bfloat16, torch.float16
if os.environ.get("ITTS_CONV_EXP"):
    nat.debug_set(5, int(os.environ["ITTS_CONV_EXP"]))
if os.environ.get("ITTS_CONV_CFG"):
    nat.debug_set(3, int(os.environ["ITTS_CONV_CFG"]))
    print("plain-GEMM tile cfg", os.environ["ITTS_CONV_CFG"])
bench("prefill QKV  3008x3840x1280", bf, 1, 3008, 1280, 3840, 1, 1)
bench("prefill proj 3008x1280x1280 +res", bf, 1, 3008, 1280, 1280, 1, 1, y_f32=True, resid=True)
bench("prefill FC   3008x5120x1280 gelu", bf, 1, 3008, 1280, 5120, 1, 1, act=1)
bench("prefill FC2  3008x1280x5120 +res", bf, 1, 3008, 5120, 1280, 1, 1, y_f32=True, resid=True)
bench("latent QKV  7488x3840x1280", bf, 1, 7488, 1280, 3840, 1, 1)
bench("latent proj 7488x1280x1280 +res", bf, 1, 7488, 1280, 1280, 1, 1, y_f32=True, resid=True)
bench("latent FC   7488x5120x1280 gelu", bf, 1, 7488, 1280, 5120, 1, 1, act=1)
bench("latent FC2  7488x1280x5120 +res", bf, 1, 7488, 5120, 1280, 1, 1, y_f32=True, resid=True)
bench("voc C768 k11 d1 T560 B32", fh, 32, 560, 768, 768, 11, 1)
bench("voc C768 k3 d5 T560 B32", fh, 32, 560, 768, 768, 3, 5)
bench("voc C384 k7 d3 T2240 B32", fh, 32, 2240, 384, 384, 7, 3)
bench("voc C192 k7 d1 T8960 B32", fh, 32, 8960, 192, 192, 7, 1)
bench("voc C96 k7 d1 T35840 B32", fh, 32, 35840, 96, 96, 7, 1)
bench("voc C48 k7 d1 T71680 B32", fh, 32, 71680, 48, 48, 7, 1)
bench("voc C24 k7 d1 T143360 B32", fh, 32, 143360, 24, 24, 7, 1)
for d in (1, 3, 5):                           # stage 3's k = 3 layers (candidates for the LDS-staged narrow kernel, ITTS_NARROW_C96)
    bench(f"voc C96 k3 d{d} T35840 B32 +res", fh, 32, 35840, 96, 96, 3, d, resid=True)
for C, T in ((48, 71680), (24, 143360)):     # the narrow layers of the last two stages, every (taps, dilation) class, with the residual epilogue
    for k, d in ((3, 1), (3, 5), (7, 1), (7, 5), (11, 1), (11, 5)):
        bench(f"narrow C{C} k{k} d{d} +res", fh, 32, T, C, C, k, d, resid=True)
if os.environ.get("MB_VOC_ALL"):              # every (taps, dilation) class of the four wide stages, with the residual epilogue
    for C, T in ((96, 35840), (192, 8960), (384, 2240), (768, 560)):
        for k, d in ((3, 1), (3, 5), (7, 1), (7, 5), (11, 1), (11, 5)):
            bench(f"stage C{C} k{k} d{d} +res", fh, 32, T, C, C, k, d, resid=True)
