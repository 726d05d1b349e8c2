#!/usr/bin/env python3
"""Side measurements for BASELINE configs 2 and 5 (profiles/README.md); the headline (config 3) is bench.py.

  config 2: batch 1, greedy, 128 acoustic tokens (whole pipeline), plus the same with the reference's default
            generation settings (beam-sample, 3 beams)
  config 5: BigVGAN only, latent fp16 [64, 1024, 1280] -> 64 x 1 048 576 samples
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import contextlib  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.infer import IndexTTS  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
with contextlib.redirect_stdout(sys.stderr):
    tts = IndexTTS.from_weights(weights.reference_config(), weights.gpt_state_dict(24), weights.bigvgan_state_dict(), device=dev,
                                precision_config={"gpt": "bf16", "vocoder": "fp16"})
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
rng = np.random.default_rng(1)
text = [torch.from_numpy(rng.integers(2, 12000, size=12)).to(torch.int32)]
out = {}


def timed(fn, reps):
    fn()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


T = 128
audio = T * 1024 / 24000.0
dt = timed(lambda: tts.infer_batch(cond_mel, text, max_mel_tokens=T + 1, force_stop=[T], seed=1, do_sample=False, num_beams=1,
                                   repetition_penalty=10.0), 5)
out["config2_b1_greedy_128tok"] = {"ms": round(1e3 * dt, 2), "audio_s_per_s": round(audio / dt, 1)}
# default generation settings of infer(): beam-sample with 3 beams; random weights never stop -> runs to max_mel_tokens
dt = timed(lambda: tts.infer_batch(cond_mel, text, max_mel_tokens=T, seed=1, do_sample=True, num_beams=3, top_k=30, top_p=0.8,
                                   temperature=1.0, repetition_penalty=10.0, length_penalty=0.0), 5)
out["config2_b1_beam_sample3_128tok"] = {"ms": round(1e3 * dt, 2), "audio_s_per_s": round((T - 1) * 1024 / 24000.0 / dt, 1)}

lat = (torch.randn(64, 1024, 1280, device=dev) * 0.5).half()
spk = torch.randn(64, 1, 512, device=dev) * 0.1
dt = timed(lambda: tts.bigvgan(lat, speaker_embedding=spk), 2)
samples = 64 * 1024 * 1024
out["config5_vocoder_b64_1024frames"] = {"ms": round(1e3 * dt, 1), "audio_s_per_s": round(samples / 24000.0 / dt, 1),
                                         "GFLOP_per_frame": 3.01, "TFLOPs": round(3.01e9 * 64 * 1024 / dt / 1e12, 1)}
print(json.dumps(out))
