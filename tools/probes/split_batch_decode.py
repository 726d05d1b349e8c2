#!/usr/bin/env python3
"""Probe: does decoding ONE batch of 32 rows as two concurrent half-batches (two engines over the same weights, two streams, two
captured step graphs replayed alternately by one host thread) beat the one 32-row chain?  The step is latency-bound (20 % of HBM), so
the second chain's launches could fill the first one's boundaries; the price is a second pass over the weights per token.
Prints microseconds per token-of-32-rows for: one engine x 32 rows | two engines x 16 rows | four engines x 8 rows."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import weights  # noqa: E402
from indextts.gpt.engine import GPTEngine  # noqa: E402

torch.set_grad_enabled(False)
B, P, NEW = 32, 72, 140
gsd = weights.gpt_state_dict(24)
base = GPTEngine(gsd, 24, 1280, 20, dtype=torch.bfloat16, device="cuda")
g = torch.Generator().manual_seed(1)
prefix = torch.randn(B, P, 1280, generator=g) * 0.1
sp = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, seed=7)


def run(parts):
    rows = B // parts
    engs = [base] + [base.fork() for _ in range(parts - 1)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    graphs = []
    for i, (e, s) in enumerate(zip(engs, streams)):
        with torch.cuda.stream(s):
            e.prefill(prefix[i * rows:(i + 1) * rows].cuda(), torch.zeros(rows, dtype=torch.int32), NEW + 2)
            sps = e._seed_to_state(sp)
            e._step_kernels(rows, sps)          # warm-up
            s.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=s):
                e._step_kernels(rows, sps)
            graphs.append(gr)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t = time.perf_counter()
        for _ in range(NEW - 4):
            for gr, s in zip(graphs, streams):
                with torch.cuda.stream(s):
                    gr.replay()
        torch.cuda.synchronize()
        best = min(best, 1e6 * (time.perf_counter() - t) / (NEW - 4))
    return best


for parts in (1, 2, 4, 1, 2):
    print(f"{parts} engine(s) x {B // parts} rows: {run(parts):8.1f} us per token of {B} rows", flush=True)
