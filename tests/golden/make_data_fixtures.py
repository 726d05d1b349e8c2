#!/usr/bin/env python3
"""Copy the DATA files the reference holds for this path into tests/golden/ (this container only; no source is copied):
    /root/reference/tests/sample_prompt.wav -> sample_prompt.wav     (the prompt BASELINE config 1 names: stereo, 44.1 kHz, 5.4 s)
    /root/reference/vocab.txt ("id: piece" dump of bpe.model, 12000 pieces) -> vocab_pieces.json (pieces in id order)
    /root/reference/tests/cases.jsonl (zh / en / mixed / pinyin texts + infer_mode) -> cases.json
The SentencePiece model itself (checkpoints/bpe.model) is not in the reference tree (a dangling link); tests/vocab_model.py
rebuilds a tokenizer with the SAME id map from vocab_pieces.json."""
import json
import os
import re
import shutil

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

pieces = {}
for line in open(f"{REF}/vocab.txt", encoding="utf-8"):
    m = re.match(r"^(\d+): (.*)$", line.rstrip("\n"))
    if m:
        pieces[int(m.group(1))] = m.group(2)
n = max(pieces) + 1
assert sorted(pieces) == list(range(n))
json.dump([pieces[i] for i in range(n)], open(os.path.join(HERE, "vocab_pieces.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=0)
cases = [json.loads(l) for l in open(f"{REF}/tests/cases.jsonl", encoding="utf-8") if l.strip()]
json.dump(cases, open(os.path.join(HERE, "cases.json"), "w", encoding="utf-8"), ensure_ascii=False, indent=1)
shutil.copyfile(f"{REF}/tests/sample_prompt.wav", os.path.join(HERE, "sample_prompt.wav"))
print(n, "pieces,", len(cases), "cases")
