#!/usr/bin/env python3
"""Generates tests/golden/host_logic.json: inputs / outputs of the reference's HOST-side helpers on the inference path,
obtained by importing the reference in this container (never at test time on the GPU box):

  indextts/infer.py      IndexTTS.remove_long_silence (:446-497), .bucket_sentences (:499-550), .pad_tokens_cat (:552-580)
  indextts/utils/front.py   TextTokenizer.split_sentences_by_token (:341-424)
  indextts/utils/common.py  tokenize_by_CJK_char / de_tokenized_by_CJK_char (:39-87)

indextts/infer.py imports `soundfile` and `omegaconf` at module level and uses them only inside functions that are not
called here (sf.write, OmegaConf.load); two more inert stand-ins next to the three of _ref_harness.py let it import.
The methods are called unbound on a namespace carrying the few attributes they read.  Only data is written."""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _ref_harness as H  # noqa: E402

H.install()
for name in ("soundfile", "omegaconf"):
    if name not in sys.modules:
        mod = types.ModuleType(name)
        if name == "omegaconf":
            mod.OmegaConf = type("OmegaConf", (), {})
        sys.modules[name] = mod
import indextts.infer as ref_infer  # noqa: E402
from indextts.utils.common import de_tokenized_by_CJK_char, tokenize_by_CJK_char  # noqa: E402
from indextts.utils.front import TextTokenizer  # noqa: E402

STOP, SIL = 8193, 52
out = {}

# ---- remove_long_silence
rng = np.random.default_rng(7)
me = types.SimpleNamespace(stop_mel_token=STOP)
cases = []


def row(parts, length):
    r = np.concatenate([np.asarray(p, dtype=np.int64) for p in parts])
    return np.concatenate([r, np.full(length - r.size, STOP, dtype=np.int64)])[:length]


batches = [
    [row([rng.integers(100, 8000, 20)], 24), row([rng.integers(100, 8000, 9)], 24)],                       # nothing to fix
    [row([rng.integers(100, 8000, 5), [SIL] * 40, rng.integers(100, 8000, 6), [SIL] * 4], 60)],          # one long run, B = 1
    [row([rng.integers(100, 8000, 5), [SIL] * 35, rng.integers(100, 8000, 6)], 50),
     row([rng.integers(100, 8000, 30)], 50), row([[SIL] * 12, rng.integers(100, 8000, 3), [SIL] * 25], 50)],
    [row([rng.integers(100, 8000, 16)], 16)],                                                            # no stop token at all
    [row([[SIL] * 31], 31), row([[SIL] * 30, [77]], 31)],                                                   # threshold 30 vs 31
]
for b in batches:
    codes = torch.from_numpy(np.stack(b))
    c, lens = ref_infer.IndexTTS.remove_long_silence(me, codes, silent_token=SIL, max_consecutive=30)
    cases.append({"codes": codes.tolist(), "out": c.tolist(), "lens": lens.tolist()})
out["remove_long_silence"] = cases

# ---- bucket_sentences / pad_tokens_cat
sent_sets = [
    [["a"] * n for n in (3, 50, 4, 5, 48, 20, 21, 6, 7, 100)],
    [["a"] * n for n in (10, 10, 10, 10, 10)],
    [["a"] * n for n in (1, 2, 3, 5, 8, 13, 21, 34, 55, 89)],
    [["a"] * 7],
]
bs = []
for sset in sent_sets:
    for mx in (4, 2, 1):
        buckets = ref_infer.IndexTTS.bucket_sentences(types.SimpleNamespace(), sset, bucket_max_size=mx)
        bs.append({"lens": [len(s) for s in sset], "max": mx,
                   "buckets": [[{"idx": it["idx"], "len": it["len"]} for it in b] for b in buckets]})
out["bucket_sentences"] = bs
cfgns = types.SimpleNamespace(gpt=types.SimpleNamespace(stop_text_token=1, start_text_token=0))
pt = []
for ver in (1.5, 1.0, None):
    toks = [torch.arange(2, 2 + n, dtype=torch.int32)[None] for n in (5, 9, 3, 20)]
    r = ref_infer.IndexTTS.pad_tokens_cat(types.SimpleNamespace(model_version=ver, cfg=cfgns), toks)
    pt.append({"version": ver, "lens": [5, 9, 3, 20], "out": r.tolist()})
out["pad_tokens_cat"] = pt

# ---- sentence splitting on token lists / CJK tokenisation
punct = [".", "!", "?", "▁.", "▁?", "▁...", "。", "？", "！"]
tok_cases = []
base = ["▁HE", "LLO", ",", "▁WOR", "LD", ".", "▁THIS", "▁IS", "▁A", "▁TEST", "?", "你", "好", "，", "世", "界", "。", "▁END"]
long_run = ["▁W%d" % i for i in range(37)] + ["."] + ["▁X", ",", "▁Y", "-", "▁Z"] * 9 + ["!"]
for toks in (base, long_run, ["▁A"] * 50, []):
    for mx in (120, 10, 6, 3):
        try:
            r = TextTokenizer.split_sentences_by_token(list(toks), punct, mx)
            tok_cases.append({"tokens": toks, "max": mx, "out": r})
        except Exception as e:  # noqa: BLE001
            tok_cases.append({"tokens": toks, "max": mx, "error": type(e).__name__})
out["split_sentences_by_token"] = tok_cases
texts = ["你好世界 hello World", "IndexTTS 是一个 zero-shot 语音合成系统！", "abc", "", "今天2024年，气温-3.5℃。", "ＡＢ全角　空格"]
out["tokenize_by_CJK_char"] = [{"text": t, "upper": tokenize_by_CJK_char(t), "keep": tokenize_by_CJK_char(t, do_upper_case=False)}
                               for t in texts]
out["de_tokenized_by_CJK_char"] = [{"text": t, "out": de_tokenized_by_CJK_char(t), "lower": de_tokenized_by_CJK_char(t, do_lower_case=True)}
                                   for t in ["你 好 世 界 HELLO WORLD", "ZERO-SHOT 语 音", "", "A B 中 C"]]
with open(os.path.join(HERE, "host_logic.json"), "w", encoding="utf-8") as f:
    json.dump(out, f, ensure_ascii=False, indent=0)
print("wrote host_logic.json:", {k: len(v) for k, v in out.items()})
