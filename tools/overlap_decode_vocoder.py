#!/usr/bin/env python3
"""What one request's token loop pays while ANOTHER stream runs the vocoder (MI355X): decode microseconds per token (engine
alone, config 3 shape, graph replay) with a background thread that vocodes [32, 140, 1280] latents in a loop on a second
stream -- an ordinary stream, or one restricted to the first N compute units (hipExtStreamCreateWithCUMask).
usage: overlap_decode_vocoder.py [none plain prio b192 b128 b64 ...]   Appends to gpurun_out/overlap_decode_vocoder.txt"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import weights  # noqa: E402
from indextts.BigVGAN.models import BigVGAN  # noqa: E402
from indextts.gpt.engine import GPTEngine  # noqa: E402
from indextts.infer import stream_with_cu_mask  # noqa: E402
from indextts.utils.config import Config  # noqa: E402

torch.set_grad_enabled(False)
modes = sys.argv[1:] or ["none", "plain", "b192", "b128", "b64"]
B, P, NEW = 32, 72, 140
eng = GPTEngine(weights.gpt_state_dict(24), 24, 1280, 20, dtype=torch.bfloat16, device="cuda")
g = torch.Generator().manual_seed(1)
prefix = torch.randn(B, P, 1280, generator=g) * 0.1
pad = torch.zeros(B, dtype=torch.int32)
sp = dict(do_sample=True, top_k=30, top_p=0.8, temperature=1.0, repetition_penalty=10.0, seed=7)
v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
v.load_state_dict(weights.bigvgan_state_dict())
v.to("cuda").to(torch.float16).remove_weight_norm()
lat = (torch.randn(32, 140, 1280, generator=g) * 0.5).cuda().half()
spk = torch.randn(1, 1, 512, generator=g).cuda()
NCU = torch.cuda.get_device_properties(0).multi_processor_count
eng.prefill(prefix, pad, NEW + 2)
eng.decode(NEW, sp, force_stop=[NEW - 1] * B)            # warm-up + graph capture
v(lat, speaker_embedding=spk)
torch.cuda.synchronize()
out = open(os.path.join(ROOT, "gpurun_out", "overlap_decode_vocoder.txt"), "a")
for mode in modes:
    stop, count = threading.Event(), [0, 0.0]
    th = None
    if mode != "none":
        if mode == "plain":
            sb = torch.cuda.Stream()
        elif mode == "prio":                       # vocoder on a low-priority stream, the token loop on a high-priority one
            sb = torch.cuda.Stream(priority=0)
        else:
            n = int(mode[1:])
            w = [0] * ((NCU + 31) // 32)
            for b in range(n):
                w[b // 32] |= 1 << (b % 32)
            sb = stream_with_cu_mask("cuda:0", w)

        def bg():
            torch.cuda.set_device(0)
            with torch.no_grad(), torch.cuda.stream(sb):
                while not stop.is_set():
                    t0 = time.perf_counter()
                    v(lat, speaker_embedding=spk)
                    sb.synchronize()
                    count[0] += 1
                    count[1] += time.perf_counter() - t0
        th = threading.Thread(target=bg, daemon=True)
        th.start()
        time.sleep(0.3)
        count[0], count[1] = 0, 0.0
    sa = torch.cuda.Stream(priority=-1) if mode == "prio" else torch.cuda.current_stream()
    with torch.cuda.stream(sa):
        eng.prefill(prefix, pad, NEW + 2)
        sa.synchronize()
        t0 = time.perf_counter()
        eng.decode(NEW, sp, force_stop=[NEW - 1] * B)
        sa.synchronize()
    us = 1e6 * (time.perf_counter() - t0) / NEW
    nv, tv = count[0], count[1]
    stop.set()
    if th is not None:
        th.join()
    line = f"{mode:6s}: decode {us:8.1f} us/token | vocoder beside it: {nv} passes, {1e3 * tv / max(nv, 1):7.2f} ms each"
    print(line, flush=True)
    out.write(line + "\n")
    out.flush()
