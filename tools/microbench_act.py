#!/usr/bin/env python3
"""Time aa_snake (channels-last) on the vocoder's stage shapes (MI355X).  Usage: microbench_act.py [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from indextts import _native as nat  # noqa: E402
from indextts.BigVGAN.models import kaiser_sinc_filter  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
f = kaiser_sinc_filter()
SHAPES = ((560, 768), (2240, 384), (8960, 192), (35840, 96), (71680, 48), (143360, 24))
if os.environ.get("ACT_C"):      # one shape only (profiler runs)
    SHAPES = tuple(sh for sh in SHAPES if sh[1] == int(os.environ["ACT_C"]))
for T, C in SHAPES:
    x = torch.randn(32, T, C, device="cuda").half()
    y = torch.empty_like(x)
    al = torch.zeros(C, device="cuda")
    be = torch.zeros(C, device="cuda")
    nat.aa_snake(x, al, be, f, f, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        nat.aa_snake(x, al, be, f, f, out=y)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    gb = 2 * x.numel() * 2 / 1e9
    print(f"aa_snake fp16 B32 T{T} C{C}: {us:8.1f} us  {gb / (us * 1e-6):7.0f} GB/s", flush=True)
