#!/usr/bin/env python3
"""Mixed-length utterances on one MI355X (BASELINE config 4's distribution: text lengths U{8..100}, stop steps U{40..400},
top-k/top-p sampling) with a queue DEEPER than the 32 decode slots: static batches of 32 (longest texts first, each batch
decoded to its longest row, infer_batch) against continuous batching (infer_queue: finished slots are refilled).
usage: bench_refill.py [batches_of_32 = 4]   -> one JSON line, also appended to gpurun_out/bench_refill.jsonl"""
import contextlib
import json
import os
import sys
import time

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
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
gsd, bsd = weights.gpt_state_dict(24), weights.bigvgan_state_dict()
gsd_c = idist.compact_gpt_state_dict(gsd, torch.bfloat16)
bsd_c = idist.compact_bigvgan_state_dict(bsd, torch.float16)
with contextlib.redirect_stdout(sys.stderr):
    tts = IndexTTS.from_weights(weights.reference_config(), gsd_c, bsd_c, device=dev, precision_config={"gpt": "bf16", "vocoder": "fp16"})
del gsd, bsd, gsd_c, bsd_c
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
texts, stops = bench.make_workload(4, depth)          # 32 * depth utterances
N = len(texts)
gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
max_new = max(stops) + 1
audio_s = sum(stops) * 1024 / 24000.0


def static(seed):
    order = sorted(range(N), key=lambda i: -int(texts[i].numel()))
    outs = [None] * N
    for k in range(0, N, 32):
        ids = order[k:k + 32]
        for i, w in zip(ids, tts.infer_batch(cond_mel, [texts[i] for i in ids], max_mel_tokens=max_new,
                                             force_stop=[stops[i] for i in ids], seed=seed, **gen)):
            outs[i] = w
    return outs


def refill(seed, check_every=16, staged=True):
    return tts.infer_queue(cond_mel, texts, slots=32, max_mel_tokens=max_new, force_stop=stops, seed=seed,
                           cache_positions=4096, check_every=check_every, staged=staged, **gen)


res = {"workload": f"{N} utterances, text U{{8..100}}, stop steps U{{40..400}} (BASELINE config 4's distribution), 32 decode slots, 1 GPU",
       "audio_seconds": round(audio_s, 1), "decode_steps_static": None}
order = sorted(range(N), key=lambda i: -int(texts[i].numel()))
res["decode_steps_static"] = sum(max(stops[i] for i in order[k:k + 32]) + 1 for k in range(0, N, 32))
res["decode_steps_ideal_refill"] = round(sum(s + 1 for s in stops) / 32, 1)
import functools  # noqa: E402
variants = [("static", static), ("refill", refill)]
if os.environ.get("ITTS_REFILL_SWEEP") == "1":
    variants += [(f"refill_every{ce}_{'staged' if stg else 'immediate'}", functools.partial(refill, check_every=ce, staged=stg))
                 for ce, stg in ((16, False), (8, True), (8, False), (32, True))]
for name, fn in variants:
    fn(1)                                             # warm-up, graph capture
    torch.cuda.synchronize()
    pe = {}
    t0 = time.perf_counter()
    outs = fn(2)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert all(int(o.numel()) == stops[i] * 1024 for i, o in enumerate(outs)), name
    res[name] = {"seconds": round(dt, 3), "audio_s_per_s": round(audio_s / dt, 1)}
    if name.startswith("refill"):
        res[name]["loop"] = dict(tts.gpt.engine.refill_stats)
    print(f"{name}: {dt:.3f} s -> {audio_s / dt:.1f} audio-s/s", file=sys.stderr, flush=True)
res["speedup"] = round(res["static"]["seconds"] / res["refill"]["seconds"], 3)
line = json.dumps(res)
print(line)
with open(os.path.join(ROOT, "gpurun_out", "bench_refill.jsonl"), "a") as f:
    f.write(line + "\n")
