"""Text front-end glue (the step before the hot path).

`TextTokenizer` mirrors the reference surface (indextts/utils/front.py:227-424): tokenize / encode /
convert_tokens_to_ids / split_sentences on a SentencePiece model after CJK-character splitting + upper-casing
(utils/common.py:39-63).  `TextNormalizer` is a light stand-in: the reference's WeTextProcessing-based normaliser is an
optional dependency that is not available offline; text is passed through with full-width punctuation folded.
When no `bpe.model` exists (offline benchmarking with synthetic weights) `TextTokenizer(..., allow_synthetic=True)`
falls back to a deterministic one-piece-per-character vocabulary so that the pipeline can still be exercised."""
import os
import re
import warnings
import zlib
from typing import List

_CJK = re.compile("([\u1100-\u11ff\u2e80-\ua4cf\ua840-\uD7AF\uF900-\uFAFF\uFE30-\uFE4F\uFF65-\uFFDC\U00020000-\U0002FFFF])")


def tokenize_by_CJK_char(line: str, do_upper_case=True) -> str:
    parts = [w.strip() for w in _CJK.split(line.strip())]
    return " ".join((w.upper() if do_upper_case else w) for w in parts if w)


class TextNormalizer:
    _PUNCT = {"，": ",", "。": ".", "！": "!", "？": "?", "；": ",", "：": ",", "、": ",", "（": "'", "）": "'", "“": "'",
              "”": "'", "‘": "'", "’": "'", "《": "'", "》": "'", "…": "...", "—": "-", "～": "-", "~": "-", "\n": " "}

    def load(self):
        return self

    def normalize(self, text: str) -> str:
        return "".join(self._PUNCT.get(ch, ch) for ch in text).strip()


class TextTokenizer:
    punctuation_marks_tokens = [".", "!", "?", "▁.", "▁?", "▁..."]

    def __init__(self, vocab_file: str, normalizer: TextNormalizer = None, allow_synthetic: bool = False):
        self.vocab_file, self.normalizer = vocab_file, normalizer
        self.sp_model = None
        if vocab_file is not None and os.path.exists(vocab_file):
            from sentencepiece import SentencePieceProcessor
            self.sp_model = SentencePieceProcessor(model_file=vocab_file)
        elif not allow_synthetic:
            raise ValueError(f"vocab file {vocab_file} does not exist")
        if self.normalizer:
            self.normalizer.load()

    bos_token_id, eos_token_id, pad_token_id = 0, 1, -1
    unk_token = "<unk>"

    @property
    def vocab_size(self):
        return self.sp_model.GetPieceSize() if self.sp_model else 12000

    def _prep(self, text):
        if self.normalizer:
            text = self.normalizer.normalize(text)
        return tokenize_by_CJK_char(text)

    def tokenize(self, text: str) -> List[str]:
        return self.encode(text, out_type=str)

    def encode(self, text: str, out_type=int, **kw):
        if len(text) == 0:
            return []
        if self.sp_model is not None:
            if len(text.strip()) == 1:
                return self.sp_model.Encode(text, out_type=out_type, **kw)
            return self.sp_model.Encode(self._prep(text), out_type=out_type, **kw)
        pieces = []
        for word in self._prep(text).split():
            if _CJK.fullmatch(word):
                pieces.append(word)
            else:
                pieces.extend(["▁" + word[0]] + list(word[1:]) if word[0].isalnum() else list(word))
        return pieces if out_type is str else self.convert_tokens_to_ids(pieces)

    def convert_tokens_to_ids(self, tokens) -> List[int]:
        if isinstance(tokens, str):
            tokens = [tokens]
        if self.sp_model is not None:
            return [self.sp_model.PieceToId(t) for t in tokens]
        return [7 + zlib.crc32(t.encode("utf-8")) % (12000 - 7) for t in tokens]  # synthetic ids, never 0/1 (start/stop)

    def convert_ids_to_tokens(self, ids):
        if self.sp_model is None:
            raise ValueError("synthetic vocabulary has no inverse")
        return self.sp_model.IdToPiece(ids)

    @staticmethod
    def split_sentences_by_token(tokens: List[str], split_tokens: List[str], max_tokens_per_sentence: int) -> List[List[str]]:
        """front.py:341-412: cut after a split token (once the running sentence has > 2 tokens); sentences that outgrow
        the limit are re-split on commas, then hyphens, then by length; short neighbours are merged back."""
        if not tokens:
            return []
        out, cur = [], []
        for i, tok in enumerate(tokens):
            cur.append(tok)
            if len(cur) <= max_tokens_per_sentence:
                if tok in split_tokens and len(cur) > 2:
                    if i < len(tokens) - 1 and tokens[i + 1] in ("'", "▁'"):
                        cur.append(tokens[i + 1])  # the quote closes this sentence (and, as in the reference, reopens the next)
                    out.append(cur)
                    cur = []
                continue
            if not ("," in split_tokens or "▁," in split_tokens) and ("," in cur or "▁," in cur):
                sub = TextTokenizer.split_sentences_by_token(cur, [",", "▁,"], max_tokens_per_sentence)
            elif "-" not in split_tokens and "-" in cur:
                sub = TextTokenizer.split_sentences_by_token(cur, ["-"], max_tokens_per_sentence)
            else:
                sub = [cur[j:j + max_tokens_per_sentence] for j in range(0, len(cur), max_tokens_per_sentence)]
                warnings.warn(f"sentence longer than {max_tokens_per_sentence} tokens was cut by length", RuntimeWarning)
            out.extend(sub)
            cur = []
        if cur:
            out.append(cur)
        merged: List[List[str]] = []
        for s in out:
            if not s:
                continue
            if merged and len(merged[-1]) + len(s) <= max_tokens_per_sentence:
                merged[-1] = merged[-1] + s
            else:
                merged.append(s)
        return merged

    def split_sentences(self, tokenized: List[str], max_tokens_per_sentence=120) -> List[List[str]]:
        return self.split_sentences_by_token(tokenized, self.punctuation_marks_tokens, max_tokens_per_sentence)
