#!/usr/bin/env python3
"""Phase split (HIP events) of several consecutive bench steps -- BASELINE config 3 through IndexTTS.infer_batch.
usage: phase_steps.py [steps]   (engine knobs such as ITTS_KSPLIT come from the environment).  Appends to gpurun_out/phase_steps.txt"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.infer import IndexTTS  # noqa: E402
from indextts.utils import dist as idist  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
gsd, bsd = weights.gpt_state_dict(24), weights.bigvgan_state_dict()
gsd_c = idist.compact_gpt_state_dict(gsd, torch.bfloat16)
bsd_c = idist.compact_bigvgan_state_dict(bsd, torch.float16)
with contextlib.redirect_stdout(sys.stderr):
    tts = IndexTTS.from_weights(weights.reference_config(), gsd_c, bsd_c, device=dev, precision_config={"gpt": "bf16", "vocoder": "fp16"})
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
texts, stops = bench.make_workload(3, 1)
gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
names = ["start", "conditioned", "prefilled", "decoded", "latents", "vocoded"]
out = open(os.path.join(ROOT, "gpurun_out", "phase_steps.txt"), "a")
for k in range(n + 2):
    pe = {}
    tts.infer_batch(cond_mel, texts, max_mel_tokens=max(stops) + 1, force_stop=stops, seed=100 + k, phase_events=pe, **gen)
    torch.cuda.synchronize()
    ph = {f"{a}->{b}": round(pe[a].elapsed_time(pe[b]), 2) for a, b in zip(names[:-1], names[1:])}
    line = f"step {k}: {ph}"
    print(line, flush=True)
    out.write(line + "\n")
