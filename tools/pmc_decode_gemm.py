#!/usr/bin/env python3
"""Driver for the PMC passes of the dominant kernel (profiles/README.md): the 97 skinny-GEMM launches of one decode
step at batch 32 (24 layers, bf16, synthetic weights), launched eagerly `reps` times after a prefill.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -o g -- python tools/pmc_decode_gemm.py
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -o g -- python tools/pmc_decode_gemm.py
Summarise with tools/pmc_summary.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import weights  # noqa: E402
from indextts.gpt.engine import GPTEngine  # noqa: E402

torch.set_grad_enabled(False)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = 32
sd = weights.gpt_state_dict(24, with_conditioner=False)
eng = GPTEngine(sd, 24, 1280, 20, dtype=torch.bfloat16, device="cuda:0")
emb = torch.randn(B, 70, 1280, device="cuda:0") * 0.02
eng.prefill(emb, torch.zeros(B, dtype=torch.int32, device="cuda:0"), 160)
torch.cuda.synchronize()
for _ in range(reps):
    n, nbytes = eng.gemm_launches_of_step(B)
torch.cuda.synchronize()
print(f"launches per step {n}, algorithmic bytes per step {nbytes}, per launch {nbytes / n:.0f}")
