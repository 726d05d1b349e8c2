#!/usr/bin/env python3
"""BigVGAN.forward with the round-2 kernel library against the current one, same box, separate processes.
usage: ab_r02_vocoder.py            (parent: runs itself twice per shape)
The old library is loaded through a ctypes proxy that answers the ABI-version query (its vocoder entry points have the
signatures of today's)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OLD = os.path.join(ROOT, "tools", "probes", "build", "libindextts_hip_r02.so")

if len(sys.argv) == 1:
    for B, T in ((32, 140), (64, 1024)):
        for lib in (OLD, ""):
            env = dict(os.environ, VOC_B=str(B), VOC_T=str(T))
            env.pop("ITTS_HIP_LIB", None)
            if lib:
                env["ITTS_HIP_LIB"] = lib
            print(f"== B={B} T={T} {'round-2 library' if lib else 'current library'}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=True)
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

import weights  # noqa: E402
from indextts.BigVGAN.models import BigVGAN  # noqa: E402
from indextts.utils.config import Config  # noqa: E402

torch.set_grad_enabled(False)
B, T = int(os.environ["VOC_B"]), int(os.environ["VOC_T"])
v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
v.load_state_dict(weights.bigvgan_state_dict())
v.to("cuda").to(torch.float16).remove_weight_norm()
g = torch.Generator().manual_seed(3)
lat = (torch.randn(B, T, 1280, generator=g) * 0.5).cuda().half()
spk = torch.randn(1, 1, 512, generator=g).cuda()
n = 5 if T <= 200 else 2
for rep in range(2):
    w, _ = v(lat, speaker_embedding=spk)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        w, _ = v(lat, speaker_embedding=spk)
    e1.record()
    torch.cuda.synchronize()
    print(f"   vocoder forward: {e0.elapsed_time(e1) / n:8.2f} ms  checksum {float(w.float().abs().sum()):.3f}", flush=True)
