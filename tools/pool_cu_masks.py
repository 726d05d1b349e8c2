#!/usr/bin/env python3
"""Two (or three) batch-32 requests in flight on one MI355X (BASELINE config 3 through indextts.infer.RequestPool): ordinary
streams against streams restricted to disjoint CU subsets (hipExtStreamCreateWithCUMask).
usage: pool_cu_masks.py [schedule ...]   schedules: serial plainN (N requests in flight) halves interleaved xcd44 split192_64 pipe pipe_bN (stage B on N CUs)
Appends to gpurun_out/pool_cu_masks.txt"""
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import synth  # noqa: E402
import weights  # noqa: E402
from indextts.infer import BatchPipeline, IndexTTS, RequestPool  # noqa: E402
from indextts.utils import dist as idist  # noqa: E402

sys.path.insert(0, ROOT)
import bench  # noqa: E402

torch.set_grad_enabled(False)
scheds = sys.argv[1:] or ["serial", "plain2", "halves", "interleaved", "xcd44", "plain3"]
dev = "cuda:0"
NCU = torch.cuda.get_device_properties(0).multi_processor_count
gsd, bsd = weights.gpt_state_dict(24), weights.bigvgan_state_dict()
gsd_c = idist.compact_gpt_state_dict(gsd, torch.bfloat16)
bsd_c = idist.compact_bigvgan_state_dict(bsd, torch.float16)
with contextlib.redirect_stdout(sys.stderr):
    tts = IndexTTS.from_weights(weights.reference_config(), gsd_c, bsd_c, device=dev, precision_config={"gpt": "bf16", "vocoder": "fp16"})
del gsd, bsd, gsd_c, bsd_c
cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
texts, stops = bench.make_workload(3, 1)
gen = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, num_beams=1)
kw = dict(max_mel_tokens=max(stops) + 1, force_stop=stops, **gen)
audio_s = sum(stops) * 1024 / 24000.0


def words(bits):
    w = [0] * ((NCU + 31) // 32)
    for b in bits:
        w[b // 32] |= 1 << (b % 32)
    return w


def masks(name):
    if name == "halves":            # CU bits 0..127 | 128..255
        return [words(range(0, NCU // 2)), words(range(NCU // 2, NCU))]
    if name == "interleaved":       # even | odd bits
        return [words(range(0, NCU, 2)), words(range(1, NCU, 2))]
    if name == "xcd44":             # bits taken 8 at a time round-robin: bit i -> group (i % 8) < 4
        return [words([i for i in range(NCU) if i % 8 < 4]), words([i for i in range(NCU) if i % 8 >= 4])]
    if name == "split192_64":
        return [words(range(0, 192)), words(range(192, NCU))]
    return None


out = open(os.path.join(ROOT, "gpurun_out", "pool_cu_masks.txt"), "a")
replicas = [tts.replica() for _ in range(max([int(n[5:]) for n in scheds if n.startswith("plain")] + [2]) - 1)]
for name in scheds:
    if name.startswith("pipe"):
        # BatchPipeline: stage B (latent pass + vocoder) of batch k on a second stream beside the token loop of batch k + 1;
        # pipe_bN: that stream restricted to the first N compute units
        nb = int(name[6:]) if name.startswith("pipe_b") else 0
        pipe = BatchPipeline(tts, cu_mask_b=words(range(nb)) if nb else None)
        for seed0, n in ((7000, 2), (8000, 8)):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            tickets = [pipe.submit(cond_mel, texts, seed=seed0 + k, **kw) for k in range(n)]
            for t in tickets:
                t.result()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        pipe.close()
    elif name == "serial":
        for k in range(2):
            tts.infer_batch(cond_mel, texts, seed=10 + k, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(4):
            tts.infer_batch(cond_mel, texts, seed=20 + k, **kw)
        torch.cuda.synchronize()
        dt, n = time.perf_counter() - t0, 4
    else:
        ninst = int(name[5:]) if name.startswith("plain") else 2
        pl = RequestPool([tts] + replicas[:ninst - 1], cu_masks=masks(name))
        pl.warm_up(cond_mel, texts, seed=900, **kw)
        n = 4 * ninst
        for seed0 in (5000, 6000):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            jobs = [pl.submit(cond_mel, texts, seed=seed0 + k, **kw) for k in range(n)]
            for j in jobs:
                j.result()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        pl.close()
        # graphs captured on a pool stream stay valid for the next pool (they are replayed on whatever stream is current)
    line = f"{name:12s}: {audio_s * n / dt:8.1f} audio-s/s  {1e3 * dt / n:7.2f} ms per batch-32 step ({n} steps)"
    print(line, flush=True)
    out.write(line + "\n")
    out.flush()
