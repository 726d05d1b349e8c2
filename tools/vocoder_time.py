#!/usr/bin/env python3
"""Milliseconds per BigVGAN.forward (fp16, [32, 140, 1280] latents = BASELINE config 3's vocoder phase).
usage: vocoder_time.py   (VOC_B / VOC_T override the batch and the frame count)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import weights  # noqa: E402
from indextts.BigVGAN.models import BigVGAN  # noqa: E402
from indextts.utils.config import Config  # noqa: E402

torch.set_grad_enabled(False)
settings = [1]
B, T = int(os.environ.get("VOC_B", "32")), int(os.environ.get("VOC_T", "140"))
v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
v.load_state_dict(weights.bigvgan_state_dict())
v.to("cuda").to(torch.float16).remove_weight_norm()
g = torch.Generator().manual_seed(3)
lat = (torch.randn(B, T, 1280, generator=g) * 0.5).cuda().half()
spk = torch.randn(1, 1, 512, generator=g).cuda()
ref = None
for rep in range(2):
    for st in settings:
        w, _ = v(lat, speaker_embedding=spk)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            w, _ = v(lat, speaker_embedding=spk)
        e1.record()
        torch.cuda.synchronize()
        if ref is None:
            ref = w.clone()
        print(f"vocoder forward: {e0.elapsed_time(e1) / 5:7.2f} ms  same_samples={bool(torch.equal(w, ref))}", flush=True)
