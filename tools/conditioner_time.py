#!/usr/bin/env python3
"""Milliseconds per prompt of the Conformer + Perceiver conditioner (one 300-frame prompt, BASELINE config 3's): the HIP engine
(gpt/conditioner.py) eagerly and as a graph replay, and the functional PyTorch form as a graph replay.  Under rocprofv3
--kernel-trace --stats the eager loop gives the per-kernel times.   usage: conditioner_time.py [frames] [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.gpt.model import UnifiedVoice  # noqa: E402

torch.set_grad_enabled(False)
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
m = UnifiedVoice(**dict(weights.reference_config()["gpt"], layers=2))
m.load_state_dict(weights.gpt_state_dict(2))
m.to("cuda").to(torch.bfloat16)
mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, frames), -6.0, 2.0)).to("cuda")
row = mel[0].t().contiguous()
eng = m.conditioner()


def bench(fn, n):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / n


def graphed(fn):
    fn()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


print(f"conditioner, frames {frames}: HIP engine")
print(f"  eager        {bench(lambda: eng(row), reps):8.3f} ms  ({eng.launches} launches)")
g, out = graphed(lambda: eng(row))
print(f"  graph replay {bench(g.replay, reps):8.3f} ms")
os.environ["ITTS_NATIVE_CONDITIONER"] = "0"
g2, out2 = graphed(lambda: m.get_conditioning(mel, None))
print(f"functional PyTorch form (fp32), graph replay {bench(g2.replay, reps):8.3f} ms")
print(f"max |HIP - functional| = {(out - out2[0]).abs().max().item():.2e}")

# ---- the speaker encoder (ECAPA-TDNN), same prompt
from indextts.BigVGAN.models import BigVGAN  # noqa: E402
from indextts.utils.config import Config  # noqa: E402

del os.environ["ITTS_NATIVE_CONDITIONER"]
v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
v.load_state_dict(weights.bigvgan_state_dict())
v.to("cuda").to(torch.float16).remove_weight_norm()
mel_btf = mel.transpose(1, 2).contiguous()
se = v.speaker_engine()
print(f"speaker encoder, frames {frames}: HIP engine")
print(f"  eager        {bench(lambda: se(row), reps):8.3f} ms  ({se.launches} launches)")
g3, out3 = graphed(lambda: se(row))
print(f"  graph replay {bench(g3.replay, reps):8.3f} ms")
os.environ["ITTS_NATIVE_SPEAKER"] = "0"
g4, out4 = graphed(lambda: v.speaker_embedding(mel_btf))
print(f"functional PyTorch form (fp32), graph replay {bench(g4.replay, reps):8.3f} ms")
print(f"max |HIP - functional| = {(out3 - out4[0, 0]).abs().max().item():.2e} (embedding max {out4.abs().max().item():.2f})")
