#!/usr/bin/env python3
"""Where the first token's milliseconds go (BASELINE config 3's batch: 32 rows, text U{20..60}, one 3.2 s prompt): conditioner graph
replay | prefix embeddings | prefill | first sample, each bracketed by a device synchronisation (so the parts add up to slightly more
than the p50 the bench reports)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.infer import IndexTTS  # noqa: E402

torch.set_grad_enabled(False)
tts = IndexTTS.from_weights(weights.reference_config(), weights.gpt_state_dict(24), weights.bigvgan_state_dict(), device="cuda:0",
                            precision_config={"gpt": "bf16", "vocoder": "fp16"})
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to("cuda:0")
g = torch.Generator().manual_seed(2)
lens = torch.randint(20, 61, (32,), generator=g)
tok = torch.full((32, int(lens.max())), 1, dtype=torch.int32)
for i, n in enumerate(lens):
    tok[i, : int(n)] = torch.randint(2, 12000, (int(n),), generator=g).to(torch.int32)
# (host ids, as IndexTTS.infer_batch has them: the padding is then known without a device round trip)
sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=1)
parts = {"conditioner": [], "prefix": [], "prefill": [], "sample": [], "speaker embedding (not on the first token's path)": []}


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    parts[name].append(1e3 * (time.perf_counter() - t))
    return out


for it in range(24):
    conds = timed("conditioner", lambda: tts._prompt_conds(cond_mel))
    emb, pad = timed("prefix", lambda: tts.gpt.prefix_rows(conds, tok))
    timed("prefill", lambda: tts.gpt.engine.prefill(emb, pad, 4, shared_rows=int(conds.shape[1])))
    timed("sample", lambda: tts.gpt.engine._sample(32, sp))
    timed("speaker embedding (not on the first token's path)", lambda: tts._prompt_spk(cond_mel))
for k, v in parts.items():
    print(f"{k:52s} {np.median(v[4:]):7.3f} ms")
