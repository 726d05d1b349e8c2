#!/usr/bin/env python3
"""cProfile of the HOST side of one bench step (IndexTTS.infer_batch, BASELINE config 3) after warm-up: which Python / torch calls the
host spends its time in while the GPU waits (prefill set-up, the hand-over from the token loop to the latent pass)."""
import contextlib
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.infer import IndexTTS  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
dev = "cuda:0"
with contextlib.redirect_stdout(sys.stderr):
    tts = IndexTTS.from_weights(weights.reference_config(), weights.gpt_state_dict(24), weights.bigvgan_state_dict(), device=dev,
                                precision_config={"gpt": "bf16", "vocoder": "fp16"})
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
texts, stops = bench.make_workload(3, 1)
kw = dict(max_mel_tokens=int(max(stops)) + 1, force_stop=stops, seed=1, do_sample=True, top_p=0.8, top_k=30, temperature=1.0,
          repetition_penalty=10.0, num_beams=1)
for _ in range(3):
    tts.infer_batch(cond_mel, texts, **kw)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    tts.infer_batch(cond_mel, texts, **kw)
    torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(45)
