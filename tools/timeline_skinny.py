#!/usr/bin/env python3
"""Where does a decode-step skinny GEMM spend its microseconds?  In-kernel time stamps (diagnostic build only).

Needs libindextts_hip_diag.so (make -C index-tts-lora_amd/csrc diag).  A full-size bf16 engine decodes a few tokens,
then ONE decode step is captured into a graph with a distinct stamp area per skinny-GEMM launch and replayed; the stamps of
the last replay are read back.  Per launch, every workgroup recorded (include/indextts_hip_diag.h):
  s_memtime at  0 entry | 1 all loads issued | 2 operands landed (vmcnt(0), diagnostic wait) | 3 MFMAs done |
                4 cross-wave barrier passed | 5 epilogue stores issued | 10 exit;  s_memrealtime (100 MHz) at entry / exit;  XCC id.
Printed per GEMM kind (median over the step's launches of that kind, us): dispatch skew (first to last workgroup entry),
segment medians over workgroups, kernel span (first entry to last exit), and the gap to the NEXT kernel's first entry.
Read the SHARES, not the lengths: the diagnostic waits forbid overlaps the product kernel has.

    ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_hip_diag.so python tools/timeline_skinny.py [--mode fold|launch]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "index-tts-lora_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
os.environ.setdefault("ITTS_HIP_LIB", os.path.join(ROOT, "index-tts-lora_amd", "indextts", "_lib", "libindextts_hip_diag.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="fold")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    os.environ["ITTS_DECODE_MODE"] = args.mode
    import synth
    import weights
    from indextts import _native as nat
    from indextts.gpt.model import UnifiedVoice
    L = nat.lib()
    L.itts_debug_stamps.restype = ctypes.c_int
    L.itts_debug_stamps.argtypes = [ctypes.c_void_p]
    torch.set_grad_enabled(False)
    dev = "cuda"
    m = UnifiedVoice(**dict(weights.reference_config()["gpt"], layers=args.layers))
    m.load_state_dict(weights.gpt_state_dict(args.layers))
    m.to(dev).to(torch.bfloat16).post_init_gpt2_config(kv_cache=True)
    eng = m.engine
    B = args.batch
    g = torch.Generator().manual_seed(2)
    lens = torch.randint(20, 61, (B,), generator=g)
    text = torch.full((B, int(lens.max())), 1, dtype=torch.long)
    for i, n in enumerate(lens):
        text[i, : int(n)] = torch.randint(2, 12000, (int(n),), generator=g)
    cond_mel = torch.from_numpy(synth.uniform("bench.cond_mel", (1, 100, 300), -6.0, 2.0)).to(dev)
    conds = m.get_conditioning(cond_mel, None)
    _, emb, mask = m.prepare_gpt_inputs(conds, text.to(dev))
    eng.prefill(emb, (mask == 0).sum(1).to(torch.int32), 200)
    sp = dict(do_sample=True, top_p=0.8, top_k=30, temperature=1.0, repetition_penalty=10.0, seed=1)
    eng.decode(70, sp)   # reach a mid-utterance context (and warm everything up)

    # one decode step captured with a stamp area per skinny launch
    WG, SLOT = 256, 16
    nlaunch = 4 * args.layers + 1
    stamps = torch.zeros(nlaunch, WG, SLOT, dtype=torch.int64, device=dev)
    kinds, geo = [], []
    orig = nat.gemm_skinny

    def wrapped(dtype, M, N, K, *a, **k):
        i = len(kinds)
        L.itts_debug_stamps(ctypes.c_void_p(stamps[i].data_ptr()))
        kinds.append({(3840, 1280): "qkv", (1280, 1280): "out_proj", (5120, 1280): "fc", (1280, 5120): "fc2"}.get((N, K), "head"))
        pl = nat.skinny_plan(dtype, M, N, K, k.get("ksplit", 1), k.get("rows_per_wg", 0), k.get("wide_wg", False),
                             k.get("ln_c") is not None)
        geo.append(pl["grid"][0] * pl["grid"][1] * pl["grid"][2])
        return orig(dtype, M, N, K, *a, **k)

    sps = eng._seed_to_state(sp)
    nat.gemm_skinny = wrapped
    try:
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            eng._step_kernels(B, sps)
    finally:
        nat.gemm_skinny = orig
        L.itts_debug_stamps(None)
    assert len(kinds) == nlaunch, (len(kinds), nlaunch)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().astype(np.float64)

    seg_names = ["issue loads", "operands land", "mfma", "lds exchange + barrier", "reduce + epilogue stores", "drain + barrier",
                 "ticket", "wait for all tickets", "row reduce (loads + LN + store)"]
    per_kind = {}
    starts, ends = [], []
    for i in range(nlaunch):
        n = geo[i]
        s = st[i, :n]
        rt0, rt1 = s[:, 11], s[:, 12]                         # 100 MHz ticks
        clk = np.median((s[:, 10] - s[:, 0]) / np.maximum(rt1 - rt0, 1.0)) * 100.0   # shader MHz
        seg = {}
        prev = 0
        for j, name in enumerate(seg_names, start=1):
            if j in (7, 8, 9) and not (s[:, j] > 0).any():
                continue
            if j in (8, 9):
                red = s[:, j] > 0                                 # reducers only
                seg[name] = float(np.median((s[red, j] - s[red, j - 1]) / clk))
                seg.setdefault("reducers", int(red.sum()))
                continue
            ok = s[:, j] > 0
            seg[name] = float(np.median((s[ok, j] - s[ok, prev]) / clk)) if ok.any() else 0.0
            prev = j
        skew = float((rt0.max() - rt0.min()) / 100.0)
        span = float((rt1.max() - rt0.min()) / 100.0)
        wg_life = float(np.median((s[:, 10] - s[:, 0]) / clk))
        starts.append(rt0.min())
        ends.append(rt1.max())
        d = per_kind.setdefault(kinds[i], [])
        d.append(dict(seg=seg, dispatch_skew_us=skew, kernel_span_us=span, wg_median_life_us=wg_life, clock_mhz=float(clk),
                      xcds=int(len(set(s[:, 13].astype(int))))))
    out = {"mode": args.mode, "batch": B, "layers": args.layers, "kinds": {}}
    for kind, lst in per_kind.items():
        keys = sorted({k for d in lst for k in d["seg"]})
        out["kinds"][kind] = {
            "launches": len(lst),
            "dispatch_skew_us": round(float(np.median([d["dispatch_skew_us"] for d in lst])), 2),
            "kernel_span_us": round(float(np.median([d["kernel_span_us"] for d in lst])), 2),
            "wg_median_life_us": round(float(np.median([d["wg_median_life_us"] for d in lst])), 2),
            "clock_mhz": round(float(np.median([d["clock_mhz"] for d in lst])), 0),
            "segments_us": {k: round(float(np.median([d["seg"].get(k, 0.0) for d in lst])), 2) for k in keys},
        }
    # time from one skinny launch's last exit to the next skinny launch's first entry (everything in between: boundaries,
    # attention, sampling ...), per predecessor kind
    gaps = {}
    for i in range(nlaunch - 1):
        gaps.setdefault(f"{kinds[i]}->{kinds[i + 1]}", []).append((starts[i + 1] - ends[i]) / 100.0)
    out["between_skinny_launches_us"] = {k: round(float(np.median(v)), 2) for k, v in gaps.items()}
    out["step_span_us"] = round(float((max(ends) - min(starts)) / 100.0), 1)
    txt = json.dumps(out, indent=1)
    print(txt)
    if args.out:
        with open(args.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()
