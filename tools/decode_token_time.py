#!/usr/bin/env python3
"""Microseconds per decode token of the GPT engine alone (BASELINE config 3 shape: 24 layers, bf16, 32 rows, ~72-position
prompt, 140 tokens, graph replay).  Appends to gpurun_out/token_time.txt.
usage: decode_token_time.py [setting ...]   A setting is a label, or engine attributes to set for that arm in the SAME process,
same box:  mode=launch   mode=fold,rows=16:16,wide=1   mode=fold,rows=0:16,wide=0   (decode_mode, fold_rows = out-projection :
FC2 rows per workgroup, fold_wide).  Other knobs come from the environment (ITTS_KSPLIT ...); ITTS_TOKENS shortens the run."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import weights  # noqa: E402
from indextts.gpt.engine import GPTEngine  # noqa: E402

torch.set_grad_enabled(False)
settings = sys.argv[1:] or ["default"]
B, P, NEW = 32, 72, int(os.environ.get("ITTS_TOKENS", "140"))
NB = int(os.environ.get("ITTS_BEAMS", "1"))      # > 1: beam-sample over B x NB rows (prompt cached once per element, row table)
gsd = weights.gpt_state_dict(24)
eng = GPTEngine(gsd, 24, 1280, 20, dtype=torch.bfloat16, device="cuda")
g = torch.Generator().manual_seed(1)
prefix = torch.randn(B, P, 1280, generator=g) * 0.1
pad = torch.zeros(B, dtype=torch.int32)
sp = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, seed=7)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", "token_time.txt"), "a")
ref = None
for rep in range(2):
    for st in settings:
        for kv in st.split(","):
            if "=" in kv:
                k, v = kv.split("=", 1)
                if k == "mode":
                    eng.decode_mode = v
                elif k == "rows":
                    eng.fold_rows = [int(x) for x in v.split(":")]
                elif k == "rc":
                    eng.fold_rows_consumers = int(v)
                elif k == "wide":
                    eng.fold_wide = v == "1"
        eng._graphs.clear()

        def run():
            if NB > 1:
                eng.prefill(prefix, pad, NEW + 2, beams=NB)
                torch.cuda.synchronize()
                t = time.perf_counter()
                c = eng.decode_beam(NEW, dict(sp, length_penalty=0.0), NB)
            else:
                eng.prefill(prefix, pad, NEW + 2)
                torch.cuda.synchronize()
                t = time.perf_counter()
                c = eng.decode(NEW, sp, force_stop=[NEW - 1] * B)
            torch.cuda.synchronize()
            return c, t
        run()                                                   # warm-up + capture
        codes, t0 = run()
        us = 1e6 * (time.perf_counter() - t0) / NEW
        if ref is None:
            ref = codes.clone()
        same = bool((codes == ref).all())
        line = f"{st:32s} {us:8.1f} us/token  codes_equal_to_first={same}"
        print(line, flush=True)
        out.write(line + "\n")
        out.flush()
