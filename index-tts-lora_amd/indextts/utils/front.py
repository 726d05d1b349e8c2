"""Text front-end glue (the step before the hot path).

`TextTokenizer` mirrors the reference surface (indextts/utils/front.py:227-424): tokenize / encode /
convert_tokens_to_ids / split_sentences on a SentencePiece model after CJK-character splitting + upper-casing
(utils/common.py:39-63).  `TextNormalizer` restates the reference's regex pre/post-processing (language routing, contraction rewrite, pinyin and
name protection, punctuation maps; pinned by tests/golden/text_normalizer.json) around the two WeTextProcessing
normalisers, which are used when that optional package is importable and passed through otherwise.
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
    """Text normaliser with the reference's surface and control flow (indextts/utils/front.py:11-219).

    The reference wraps two WeTextProcessing normalisers (zh / en number-date-unit verbalisation, loaded in `load()`) in
    regex pre/post-processing.  WeTextProcessing is an optional dependency that cannot be fetched offline: `load()` uses it
    when importable and otherwise keeps pass-through normalisers (a warning says so once).  Everything AROUND those two
    calls is restated here and pinned by reference-run fixtures (tests/golden/text_normalizer.json):
      * language routing: Chinese path when the text has a CJK character, no latin letter, is an e-mail address, or
        carries tone-numbered pinyin (`xuan4`), else English path (front.py:92-101);
      * "what's / it's / that's ..." -> "... is" (front.py:88, 132, 150);
      * pinyin protection: tone-numbered syllables are swapped for placeholders before normalisation and restored after,
        upper-cased, with j/q/x + u written as v (ju4 -> JV4) (front.py:148-160, 196-226);
      * name protection: 中文·中文(-中文) names survive the punctuation map (front.py:162-194);
      * the punctuation maps (front.py:19-59)."""

    # full-width / bracket / quote folding applied after normalisation (interface data of front.py:19-55)
    _FOLD = (("：", ","), ("；", ","), (";", ","), ("，", ","), ("。", "."), ("！", "!"), ("？", "?"), ("\n", " "), ("·", "-"),
             ("、", ","), ("...", "…"), (",,,", "…"), ("，，，", "…"), ("……", "…"), ("“", "'"), ("”", "'"), ('"', "'"),
             ("‘", "'"), ("’", "'"), ("（", "'"), ("）", "'"), ("(", "'"), (")", "'"), ("《", "'"), ("》", "'"), ("【", "'"),
             ("】", "'"), ("[", "'"), ("]", "'"), ("—", "-"), ("～", "-"), ("~", "-"), ("「", "'"), ("」", "'"), (":", ","))

    PINYIN_TONE_PATTERN = (r"(?<![a-z])((?:[bpmfdtnlgkhjqxzcsryw]|[zcs]h)?(?:[aeiouüv]|[ae]i|u[aio]|ao|ou|i[aue]|[uüv]e|"
                           r"[uvü]ang?|uai|[aeiuv]n|[aeio]ng|ia[no]|i[ao]ng)|ng|er)([1-5])")
    NAME_PATTERN = r"[\u4e00-\u9fff]+(?:[-·—][\u4e00-\u9fff]+){1,2}"
    ENGLISH_CONTRACTION_PATTERN = r"(what|where|who|which|how|t?here|it|s?he|that|this)'s"

    class _PassThrough:
        def normalize(self, text):
            return text

    def __init__(self):
        self.zh_normalizer = None
        self.en_normalizer = None
        self.char_rep_map = dict(self._FOLD)
        self.zh_char_rep_map = {"$": ".", **self.char_rep_map}   # the Chinese path also folds "$" (front.py:56-59)
        # one alternation per map, keys in table order: the leftmost-longest behaviour of the reference's pattern
        self._fold_en = re.compile("|".join(re.escape(k) for k in self.char_rep_map))
        self._fold_zh = re.compile("|".join(re.escape(k) for k in self.zh_char_rep_map))
        self._pinyin = re.compile(self.PINYIN_TONE_PATTERN, re.IGNORECASE)
        self._name = re.compile(self.NAME_PATTERN, re.IGNORECASE)

    # ---- reference surface ------------------------------------------------------------------------------------
    def load(self):
        if self.zh_normalizer is not None and self.en_normalizer is not None:
            return self
        try:  # the reference's optional dependency (front.py:103-127); absent offline
            from tn.chinese.normalizer import Normalizer as NormalizerZh
            from tn.english.normalizer import Normalizer as NormalizerEn
            self.zh_normalizer = NormalizerZh(remove_interjections=False, remove_erhua=False, overwrite_cache=False)
            self.en_normalizer = NormalizerEn(overwrite_cache=False)
        except Exception:  # noqa: BLE001
            if not getattr(TextNormalizer, "_warned", False):
                warnings.warn("WeTextProcessing is not installed: numbers, dates and units are passed to the tokenizer "
                              "un-verbalised (punctuation, pinyin and name handling are active)", RuntimeWarning)
                TextNormalizer._warned = True
            self.zh_normalizer = self.en_normalizer = TextNormalizer._PassThrough()
        return self

    def match_email(self, email: str) -> bool:
        return re.match(r"^[a-zA-Z0-9]+@[a-zA-Z0-9]+\.[a-zA-Z]+$", email) is not None

    def use_chinese(self, s: str) -> bool:
        if re.search(r"[\u4e00-\u9fff]", s) or not re.search(r"[a-zA-Z]", s) or self.match_email(s):
            return True
        return self._pinyin.search(s) is not None

    def normalize(self, text: str) -> str:
        if not self.zh_normalizer or not self.en_normalizer:
            print("[error] TextNormalizer.load() has not been called")
            return ""
        text = re.sub(self.ENGLISH_CONTRACTION_PATTERN, r"\1 is", text, flags=re.IGNORECASE)
        if not self.use_chinese(text):
            try:
                result = self.en_normalizer.normalize(text)
            except Exception:  # noqa: BLE001
                result = text
            return self._fold_en.sub(lambda m: self.char_rep_map[m.group()], result)
        held, pinyins = self.save_pinyin_tones(text.rstrip())
        held, names = self.save_names(held)
        try:
            result = self.zh_normalizer.normalize(held)
        except Exception:  # noqa: BLE001
            result = ""
        result = self.restore_pinyin_tones(self.restore_names(result, names), pinyins)
        return self._fold_zh.sub(lambda m: self.zh_char_rep_map[m.group()], result)

    def correct_pinyin(self, pinyin: str) -> str:
        """j / q / x followed by u or ü spell the vowel ü: written v, whole syllable upper-cased (ju4 -> JV4);
        other initials are returned untouched."""
        if pinyin[0] not in "jqxJQX":
            return pinyin
        return re.sub(r"([jqx])[uü](n|e|an)*(\d)", r"\g<1>v\g<2>\g<3>", pinyin, flags=re.IGNORECASE).upper()

    @staticmethod
    def _hold(text, items, tag):
        for i, item in enumerate(items):
            text = text.replace(item, f"<{tag}_{chr(ord('a') + i)}>")
        return text

    def save_pinyin_tones(self, original_text: str):
        found = ["".join(m) for m in self._pinyin.findall(original_text)]
        if not found:
            return original_text, None
        items = list(dict.fromkeys(found))   # first-occurrence order (the reference's set() order is arbitrary; the
        return self._hold(original_text, items, "pinyin"), items   # normalised text does not depend on it)

    def restore_pinyin_tones(self, normalized_text: str, original_pinyin_list):
        for i, pinyin in enumerate(original_pinyin_list or ()):
            normalized_text = normalized_text.replace(f"<pinyin_{chr(ord('a') + i)}>", self.correct_pinyin(pinyin))
        return normalized_text

    def save_names(self, original_text: str):
        found = self._name.findall(original_text)
        if not found:
            return original_text, None
        items = list(dict.fromkeys(found))
        return self._hold(original_text, items, "n"), items

    def restore_names(self, normalized_text: str, original_name_list):
        for i, name in enumerate(original_name_list or ()):
            normalized_text = normalized_text.replace(f"<n_{chr(ord('a') + i)}>", name)
        return normalized_text


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
