"""IndexTTS inference orchestrator -- same Python surface as the reference's indextts/infer.py (class IndexTTS,
`infer`, `infer_fast`, `set_seed`), running the GPT decoder and the BigVGAN vocoder on hand-written gfx950 kernels.

Control flow follows the reference (infer.py:185-439 __init__, :446-497 remove_long_silence, :499-580 bucketing and
padding, :595-777 infer_fast, :779-917 infer); the arithmetic is in indextts.gpt.engine / indextts.BigVGAN.models.
Differences that are deliberate and visible: no CPU/MPS execution path (the HIP library is mandatory), bitsandbytes
quantisation and DeepSpeed are accepted in the config but not used, and the conditioning latents / speaker embedding of
a prompt are computed once per call instead of once per sentence (same values).
"""
from __future__ import annotations

import json
import os
import queue
import random
import threading
import time
import warnings
from typing import Dict, List

import numpy as np
import torch

from indextts import _native as nat
from indextts.BigVGAN.models import BigVGAN as Generator
from indextts.gpt.model import UnifiedVoice
from indextts.utils.audio import read_audio, write_pcm16
from indextts.utils.checkpoint import load_checkpoint
from indextts.utils.config import Config, load_config
from indextts.utils.feature_extractors import MelSpectrogramFeatures, resample
from indextts.utils.front import TextNormalizer, TextTokenizer


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def _resolve_dtype(name):
    if name in ("bf16", "bfloat16"):
        return torch.bfloat16
    if name in ("fp16", "float16"):
        return torch.float16
    if name == "fp8":
        return torch.bfloat16  # no fp8 weights path; served in bf16
    return torch.float32


class IndexTTS:
    def __init__(self, cfg_path="checkpoints/config.yaml", model_dir="checkpoints", is_fp16=True, device=None,
                 use_cuda_kernel=None, speaker_info_path=None, precision_config=None, gpt_path=None,
                 _weights=None, _cfg=None):
        if device is None:
            device = "cuda:0" if torch.cuda.is_available() else "cpu"
        if not str(device).startswith("cuda") or not torch.cuda.is_available():
            raise RuntimeError("this build of IndexTTS runs on an AMD GPU through libindextts_hip.so; "
                               f"device={device!r} is not supported (there is no CPU fallback)")
        self.device = device
        self.is_fp16 = is_fp16
        self.use_cuda_kernel = True  # the fused HIP activation kernel is always used
        self.cfg = _cfg if _cfg is not None else load_config(cfg_path)
        self.model_dir = model_dir

        # precision: precision_config > config_inference.yaml > cfg.inference > is_fp16   (infer.py:213-304)
        source = "Runtime Args" if precision_config is not None else None
        if precision_config is None:
            p = os.path.join(model_dir, "config_inference.yaml") if model_dir else None
            if p and os.path.exists(p):
                icfg = load_config(p)
                if "inference" in icfg:
                    precision_config, source = icfg["inference"], "config_inference.yaml"
            elif "inference" in self.cfg:
                precision_config, source = self.cfg["inference"], "config.yaml [inference]"
        self.use_quantization = self.load_in_8bit = self.load_in_4bit = False
        if precision_config and isinstance(precision_config, dict):
            gpt_p = precision_config.get("gpt", "bf16")
            quant = precision_config.get("quantization", {}) or {}
            if quant.get("enabled", False) or gpt_p in ("int8", "int4"):
                print(">> [warning] bitsandbytes quantisation is not available in this build; using BF16 weights")
                self.gpt_dtype = torch.bfloat16
            else:
                self.gpt_dtype = _resolve_dtype(gpt_p)
            self.vocoder_dtype = _resolve_dtype(precision_config.get("vocoder", "bf16"))
            print(f">> [config] mixed precision ({source}): GPT={self.gpt_dtype} vocoder={self.vocoder_dtype}")
        elif self.is_fp16:
            self.gpt_dtype, self.vocoder_dtype = torch.bfloat16, torch.float32
            print(">> [config] BF16 GPT / FP32 vocoder (legacy is_fp16)")
        else:
            self.gpt_dtype = self.vocoder_dtype = torch.float32
            print(">> [config] FP32 (legacy)")
        if self.gpt_dtype == torch.float16:
            self.gpt_dtype = torch.bfloat16  # GPT kernels: fp32 | bf16
        self.dvae_dtype = self.gpt_dtype
        self.dtype = self.gpt_dtype if self.gpt_dtype != torch.float32 else None
        self.stop_mel_token = self.cfg.gpt.stop_mel_token

        if _weights is None:
            if gpt_path is not None:
                self.gpt_path = gpt_path if os.path.isabs(gpt_path) else os.path.join(model_dir, gpt_path)
            else:
                self.gpt_path = os.path.join(model_dir, self.cfg.gpt_checkpoint)
        else:
            self.gpt_path = None

        self._cache_conds = None
        self._feat_graphs = {}
        self._batch_feat = None   # (prompt tensor, its version counter, conds, spk) of the last infer_batch prompt
        # the latent pass takes the prompt's keys / values from the KV cache the decode loop leaves behind (same bits)
        self.reuse_prompt_kv = os.environ.get("ITTS_REUSE_PROMPT_KV", "1") == "1"
        self.gpt = UnifiedVoice(**self.cfg.gpt)
        if _weights is None:
            load_checkpoint(self.gpt, self.gpt_path)
        else:
            self.gpt.load_state_dict(_weights["gpt"])
        self.gpt = self.gpt.to(self.device).to(self.gpt_dtype).eval()
        self.gpt.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=self.gpt_dtype != torch.float32)
        print(f">> [system] GPT loaded ({self.gpt_dtype})")

        self.bigvgan = Generator(self.cfg.bigvgan, use_cuda_kernel=True)
        if _weights is None:
            self.bigvgan_path = os.path.join(model_dir, self.cfg.bigvgan_checkpoint)
            self.bigvgan.load_state_dict(torch.load(self.bigvgan_path, map_location="cpu")["generator"])
        else:
            self.bigvgan_path = None
            self.bigvgan.load_state_dict(_weights["bigvgan"])
        self.bigvgan = self.bigvgan.to(self.device).to(self.vocoder_dtype)
        self.bigvgan.remove_weight_norm()
        self.bigvgan.eval()
        print(f">> [system] BigVGAN loaded ({self.vocoder_dtype})")

        self.normalizer = TextNormalizer()
        self.normalizer.load()
        bpe = self.cfg.dataset["bpe_model"] if "dataset" in self.cfg else None
        self.bpe_path = os.path.join(model_dir, bpe) if (bpe and model_dir) else None
        self.tokenizer = TextTokenizer(self.bpe_path, self.normalizer, allow_synthetic=_weights is not None)

        self.cache_audio_prompt = None
        self.cache_cond_mel = None
        self._cache_spk = None
        self.gr_progress = None
        self.model_version = self.cfg.version if "version" in self.cfg else None
        self.speaker_list = []
        if speaker_info_path and os.path.exists(speaker_info_path):
            try:
                with open(speaker_info_path, "r", encoding="utf-8") as f:
                    self.speaker_list = [it["speaker"] for it in json.load(f) if "speaker" in it]
                print(f">> [system] multi-speaker mode ({len(self.speaker_list)} speakers)")
            except Exception as e:  # noqa: BLE001
                print(f">> [error] could not read speaker info: {e}")
        self.mel_extractor = MelSpectrogramFeatures()

    # `tts.gpt` is replaceable, as the reference's callers do when they hot-swap a fine-tuned checkpoint (api.py:118-175,
    # webui.py:107-160: build UnifiedVoice(**tts.cfg.gpt), load_checkpoint, .to / .eval / .half, post_init_gpt2_config, then
    # `tts.gpt = new_gpt; tts.gpt_path = path`).  Everything derived from the old weights goes with it: cached conditioning
    # latents and the captured conditioner graph.
    @property
    def gpt(self):
        return self._gpt

    @gpt.setter
    def gpt(self, module):
        self._gpt = module
        self._cache_conds = None
        self._feat_graphs = {}
        self._batch_feat = None

    @gpt.deleter
    def gpt(self):   # `del tts.gpt` before the swap (api.py:160)
        self._gpt = None

    def reload_gpt(self, model_path: str):
        """The reference's model hot-swap (api.py:137-169) as one call: a new UnifiedVoice from `model_path` in this
        instance's precision replaces `self.gpt`.  Returns the config dict load_checkpoint found next to the file."""
        new_gpt = UnifiedVoice(**self.cfg.gpt)
        info = load_checkpoint(new_gpt, model_path)
        new_gpt = new_gpt.to(self.device).to(self.gpt_dtype).eval()
        new_gpt.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=self.gpt_dtype != torch.float32)
        del self.gpt
        self.torch_empty_cache()
        self.gpt = new_gpt
        self.gpt_path = model_path
        return info

    @classmethod
    def from_weights(cls, cfg, gpt_state_dict, bigvgan_state_dict, device="cuda:0", is_fp16=True, precision_config=None):
        """Build from in-memory state dicts in the reference checkpoint key format (offline / synthetic weights)."""
        cfg = cfg if isinstance(cfg, Config) else Config(cfg)
        return cls(model_dir="", is_fp16=is_fp16, device=device, precision_config=precision_config, _cfg=cfg,
                   _weights={"gpt": gpt_state_dict, "bigvgan": bigvgan_state_dict})

    def replica(self) -> "IndexTTS":
        """Another instance over the same weight tensors (GPT packed weights, conditioner, vocoder) with its own KV cache,
        decode state, prompt caches and captured graphs -- for RequestPool."""
        import copy
        r = copy.copy(self)
        r.gpt = self.gpt.replica()
        r.cache_audio_prompt = r.cache_cond_mel = r._cache_conds = r._cache_spk = r._batch_feat = None
        r._feat_graphs = {}       # captured graphs replay into their own static buffers: one set per instance
        return r

    # ------------------------------------------------------------------------------------------------ helpers
    def remove_long_silence(self, codes: torch.Tensor, silent_token=52, max_consecutive=30):
        """infer.py:446-497: cut at the first stop token; when a row holds more than `max_consecutive` silent tokens,
        keep at most 10 per run."""
        c, lens = IndexTTS._squeeze_silence_host(self, codes, silent_token, max_consecutive)   # (self: anything with stop_mel_token)
        return torch.from_numpy(c).to(codes.device), torch.tensor(lens, dtype=torch.long, device=codes.device)

    def _squeeze_silence_host(self, codes: torch.Tensor, silent_token=52, max_consecutive=30):
        """remove_long_silence on the host: (codes as a contiguous numpy array, lengths as a list) -- ONE device-to-host copy; the
        batch path keeps working from these host copies instead of uploading them and reading them back."""
        c = codes.detach().cpu().numpy()
        if c.shape[0] and not ((c == silent_token).sum(axis=1) > max_consecutive).any():
            # no row has a long silence (the usual case): only the cut at the first stop token, all rows at once
            is_stop = c == self.stop_mel_token
            lens = np.where(is_stop.any(axis=1), is_stop.argmax(axis=1), c.shape[1]).tolist()
            return np.ascontiguousarray(c[:, : max(lens)]), lens
        rows, lens, fixed = [], [], False
        for code in c:
            stops = np.nonzero(code == self.stop_mel_token)[0]
            n = int(stops[0]) if stops.size else code.shape[0]
            if int((code == silent_token).sum()) > max_consecutive:
                keep, run = [], 0
                for k in range(n):
                    if code[k] != silent_token:
                        keep.append(k)
                        run = 0
                    elif run < 10:
                        keep.append(k)
                        run += 1
                rows.append(code[keep])
                lens.append(len(keep))
                fixed = True
            else:
                rows.append(code[:n])
                lens.append(n)
        if fixed:
            if len(rows) > 1:
                m = max(len(r) for r in rows)
                out = np.full((len(rows), m), self.stop_mel_token, dtype=c.dtype)
                for i, r in enumerate(rows):
                    out[i, :len(r)] = r
                c = out
            else:
                c = rows[0][None]
        mx = max(lens)
        if mx < c.shape[1]:
            c = c[:, :mx]
        return np.ascontiguousarray(c), lens

    def bucket_sentences(self, sentences, bucket_max_size=4) -> List[List[Dict]]:
        """infer.py:499-550: sort by length, open a new bucket when a sentence is >= 1.5x the bucket median or the
        bucket is full, then fold singleton buckets into buckets with room."""
        items = [{"idx": i, "sent": s, "len": len(s)} for i, s in enumerate(sentences)]
        if len(items) <= bucket_max_size:
            return [items]
        buckets, last, median = [], None, 0
        for it in sorted(items, key=lambda x: x["len"]):
            if it["len"] == 0:
                continue
            if last is None or it["len"] >= int(median * 1.5) or len(last) >= bucket_max_size:
                last = [it]
                buckets.append(last)
                median = it["len"]
            else:
                last.append(it)
                median = last[len(last) // 2]["len"]
        out = [b for b in buckets if len(b) > 1]
        ones = [b[0] for b in buckets if len(b) == 1]
        if ones:
            for b in out:
                if len(b) < bucket_max_size:
                    b.append(ones.pop(0))
                    if not ones:
                        break
            if ones:
                out.extend([ones[i:i + bucket_max_size] for i in range(0, len(ones), bucket_max_size)])
        return out

    def pad_tokens_cat(self, tokens: List[torch.Tensor]) -> torch.Tensor:
        """infer.py:552-580: right-pad with the stop text token (v>=1.5), or 8 stops then start tokens (older)."""
        stop, start = self.cfg.gpt.stop_text_token, self.cfg.gpt.start_text_token
        if self.model_version and self.model_version >= 1.5:
            ts = [t.squeeze(0) for t in tokens]
            m = max(t.size(0) for t in ts)
            return torch.stack([torch.cat([t, torch.full((m - t.size(0),), stop, dtype=t.dtype, device=t.device)]) for t in ts])
        m = max(t.size(1) for t in tokens)
        outs = []
        for t in tokens:
            padn = m - t.size(1)
            if padn > 0:
                n = min(8, padn)
                t = torch.nn.functional.pad(t, (0, n), value=stop)
                t = torch.nn.functional.pad(t, (0, padn - n), value=start)
            outs.append(t[:, :m])
        return torch.cat(outs, dim=0)

    def torch_empty_cache(self):
        try:
            torch.cuda.empty_cache()
        except Exception:  # noqa: BLE001
            pass

    def _set_gr_progress(self, value, desc):
        if self.gr_progress is not None:
            self.gr_progress(value, desc=desc)

    def _prompt(self, audio_prompt):
        """infer.py:789-800: mono mean -> 24 kHz -> log-mel, cached by path; plus (new) cached conditioner outputs."""
        if self.cache_cond_mel is None or self.cache_audio_prompt != audio_prompt:
            audio, sr = read_audio(audio_prompt)
            audio = torch.from_numpy(audio.T if audio.ndim > 1 else audio.reshape(1, -1)).float()
            audio = torch.mean(audio, dim=0, keepdim=True)
            audio = resample(audio, sr, 24000)
            self.cache_cond_mel = self.mel_extractor(audio).to(self.device)
            self.cache_audio_prompt = audio_prompt
            self._cache_conds = self._cache_spk = None
        return self.cache_cond_mel

    MAX_FEATURE_GRAPHS = 64     # (network, prompt shape) pairs kept captured: ~30 MB of activations + a graph each

    def _graphed(self, name, fn, cond_mel):
        """fn(prompt mel) through a CUDA graph per (network, prompt shape): the host-side PyTorch networks are a few hundred
        small launches each; the first call runs eagerly as warm-up, the second captures, later ones replay (any capture
        failure falls back to eager execution)."""
        key = (name,) + tuple(cond_mel.shape)
        ent = self._feat_graphs.get(key)
        if ent is None:
            if len(self._feat_graphs) >= self.MAX_FEATURE_GRAPHS:
                # a server that has seen this many prompt shapes starts over: the graphs go, and with them the per-length
                # activation buffers of the two front-end engines they replay into (this thread's)
                self._feat_graphs.clear()
                for eng in (self.gpt.conditioner(), self.bigvgan.speaker_engine()):
                    if eng is not None:
                        eng.forget()
            self._feat_graphs[key] = "warm"
            return fn(cond_mel)
        if ent == "warm":
            try:
                static_mel = cond_mel.clone()
                g = torch.cuda.CUDAGraph()
                with nat.CAPTURE_LOCK, torch.cuda.graph(g):
                    out = fn(static_mel)
                ent = (g, static_mel, out)
            except Exception as e:  # noqa: BLE001
                warnings.warn(f"prompt-feature graph capture failed ({e}); running eagerly", RuntimeWarning)
                ent = "eager"
            self._feat_graphs[key] = ent
        if ent == "eager":
            return fn(cond_mel)
        g, static_mel, out = ent
        static_mel.copy_(cond_mel)
        g.replay()
        return out.clone()

    def _prompt_conds(self, cond_mel):
        """Conditioning latents [1, 32, D] of a prompt mel (Conformer + Perceiver): all the first token needs."""
        return self._graphed("conds", lambda m: self.gpt.get_conditioning(m, None), cond_mel)   # None: rows are T frames long

    def _prompt_spk(self, cond_mel):
        """Speaker embedding [1, 1, 512] of a prompt mel (ECAPA-TDNN): only the vocoder needs it."""
        return self._graphed("spk", lambda m: self.bigvgan.speaker_embedding(m.transpose(1, 2)), cond_mel)

    def _prompt_features(self, cond_mel):
        """(conditioning latents, speaker embedding) of a prompt mel."""
        return self._prompt_conds(cond_mel), self._prompt_spk(cond_mel)

    def _conds(self, cond_mel, speaker_id=None):
        """Conditioning latents of the call.  The reference passes BOTH the prompt mel and speaker_ids=[speaker_id] to
        gpt.inference_speech / gpt.forward (infer.py:833-847, :864-874), and get_conditioning uses a stored
        mean_condition_{id} only when there is NO prompt mel (model.py:488-509): with an audio prompt -- which infer()
        always has -- the encoder path wins and speaker_id only has to be a known id.  Same here."""
        if cond_mel is None:
            return self.gpt.get_conditioning(None, None, speaker_ids=[speaker_id] if speaker_id else None)
        if self._cache_conds is None:
            self._cache_conds = self.gpt.get_conditioning(cond_mel, torch.tensor([cond_mel.shape[-1]], device=self.device))
        return self._cache_conds

    def _spk(self, cond_mel):
        if self._cache_spk is None:
            self._cache_spk = self.bigvgan.speaker_embedding(cond_mel.transpose(1, 2))
        return self._cache_spk

    @staticmethod
    def _gen_kwargs(kw):
        return dict(do_sample=kw.pop("do_sample", True), top_p=kw.pop("top_p", 0.8), top_k=kw.pop("top_k", 30),
                    temperature=kw.pop("temperature", 1.0), length_penalty=kw.pop("length_penalty", 0.0),
                    num_beams=kw.pop("num_beams", 3), repetition_penalty=kw.pop("repetition_penalty", 10.0)), \
            kw.pop("max_mel_tokens", 600)

    def _generate(self, conds, text_tokens, gen, max_mel_tokens, **extra):
        """gpt.inference_speech with precomputed conditioning latents (same values as recomputing them per call)."""
        g = self.gpt
        emb, pad = g.prefix_rows(conds, text_tokens)
        shared = int(conds.shape[1]) if conds.shape[0] == 1 else 0   # one prompt: every row starts with the same latents
        sp = dict(do_sample=bool(gen["do_sample"]), top_p=float(gen["top_p"]), top_k=int(gen["top_k"]),
                  temperature=float(gen["temperature"]), repetition_penalty=float(gen["repetition_penalty"]),
                  seed=int(extra.pop("seed", torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)))
        if not sp["do_sample"]:
            sp["top_p"], sp["top_k"], sp["temperature"] = 1.0, 0, 1.0
        nb = int(gen.get("num_beams", 1))
        if nb > 1:  # beam search / beam-sample: every row becomes num_beams rows (HF generate semantics)
            sp["length_penalty"] = float(gen.get("length_penalty", 0.0))
            g.engine.prefill(emb.repeat_interleave(nb, dim=0), pad.repeat_interleave(nb), max_mel_tokens, shared_rows=shared, paged=False)
            return g.engine.decode_beam(max_mel_tokens, sp, nb)
        g.engine.prefill(emb, pad, max_mel_tokens, shared_rows=shared)
        return g.engine.decode(max_mel_tokens, sp, force_stop=extra.pop("force_stop", None))

    def _latents(self, conds, text_rows: List[torch.Tensor], code_rows: List[torch.Tensor], reuse_prefix=False, cache_rows=None):
        """Teacher-forced pass for several utterances at once (right-padded; causal attention makes padding inert).
        Each row reproduces gpt(cond, text, [L], codes, code_len*1024, return_latent=True) of infer.py:864-874.
        The [cond | text | mel] embedding batch is assembled with two gathers from index arrays built on the host (one
        upload) instead of a dozen small launches per utterance.
        reuse_prefix: the rows are, in order, the batch the engine's LAST prefill() cached (same conds, same texts) and nothing
        has touched its KV cache since -- then only the mel rows are recomputed and the prompt's keys / values come from the
        cache (GPTEngine.latent_mel_rows: same bits, ~40 % fewer GEMM rows)."""
        g, eng, dev = self.gpt, self.gpt.engine, self.device
        if reuse_prefix and self.reuse_prompt_kv:
            cl = [int(c.numel()) for c in code_rows]
            flat = torch.cat([c.reshape(-1).long() for c in code_rows]).cpu().numpy() if code_rows else np.zeros(0, np.int64)
            # start | codes | stop of every row and the position of each token in its row, for all rows at once
            cln = np.asarray(cl, dtype=np.int64)
            m = cln + 2
            offs = np.concatenate([[0], np.cumsum(m)])
            pos = np.arange(int(offs[-1])) - np.repeat(offs[:-1], m)
            ids = np.full(int(offs[-1]), g.stop_mel_token, dtype=np.int64)
            ids[pos == 0] = g.start_mel_token
            ids[(pos > 0) & (pos <= np.repeat(cln, m))] = flat
            idx = torch.from_numpy(np.stack([ids, pos])).to(dev)
            enc = eng.latent_mel_rows(eng.mel_emb[idx[0]] + eng.mel_pos[idx[1]], [c + 2 for c in cl], cache_rows)
            return [enc[int(offs[i]): int(offs[i]) + cl[i]] for i in range(len(cl))]

        def flat_host(rows):
            flat = torch.cat([r.reshape(-1).long() for r in rows]) if rows else torch.zeros(0, dtype=torch.long)
            return flat.cpu().numpy()

        tl = [int(t.numel()) for t in text_rows]
        cl = [int(c.numel()) for c in code_rows]
        tflat, cflat = flat_host(text_rows), flat_host(code_rows)
        nc = int(conds.shape[1])
        S = max(nc + t + 2 + c + 2 for t, c in zip(tl, cl))
        ri, ci, tok, pos, spans = ([], []), ([], []), ([], []), ([], []), []
        to = co = 0
        for i, (t, c) in enumerate(zip(tl, cl)):
            ti = np.concatenate([[g.start_text_token], tflat[to: to + t], [g.stop_text_token]])
            mi = np.concatenate([[g.start_mel_token], cflat[co: co + c], [g.stop_mel_token]])
            to, co = to + t, co + c
            for k, (ids, c0) in enumerate(((ti, nc), (mi, nc + ti.size))):
                ri[k].append(np.full(ids.size, i))
                ci[k].append(c0 + np.arange(ids.size))
                tok[k].append(ids)
                pos[k].append(np.arange(ids.size))
            spans.append((nc + ti.size, c))
        n_t = sum(a.size for a in tok[0])
        packed = np.stack([np.concatenate(ri[0] + ri[1]), np.concatenate(ci[0] + ci[1]), np.concatenate(tok[0] + tok[1]),
                           np.concatenate(pos[0] + pos[1])]).astype(np.int64)
        idx = torch.from_numpy(packed).to(dev)
        batch = torch.zeros(len(tl), S, conds.shape[2], dtype=torch.float32, device=dev)
        batch[:, :nc] = conds[0].to(dev, torch.float32)
        batch[idx[0, :n_t], idx[1, :n_t]] = eng.text_emb[idx[2, :n_t]] + eng.text_pos[idx[3, :n_t]]
        batch[idx[0, n_t:], idx[1, n_t:]] = eng.mel_emb[idx[2, n_t:]] + eng.mel_pos[idx[3, n_t:]]
        enc = eng.latent(batch, lengths=[s0 + n + 2 for s0, n in spans])   # real rows only (cond | text | mel incl. start/stop)
        return [enc[i, s0: s0 + n] for i, (s0, n) in enumerate(spans)]

    def _finish(self, wavs, output_path, start_time, gpt_gen_time, gpt_forward_time, bigvgan_time, sampling_rate=24000):
        end_time = time.perf_counter()
        wav = torch.cat(wavs, dim=1)
        wav_length = wav.shape[-1] / sampling_rate
        print(f">> [stats] total: {end_time - start_time:.2f}s (RTF: {(end_time - start_time) / max(wav_length, 1e-9):.4f})")
        print(f"   - GPT generation: {gpt_gen_time:.2f}s")
        print(f"   - GPT forward: {gpt_forward_time:.2f}s")
        print(f"   - vocoder: {bigvgan_time:.2f}s")
        wav = wav.cpu()
        if output_path:
            if os.path.isfile(output_path):
                os.remove(output_path)
            if os.path.dirname(output_path) != "":
                os.makedirs(os.path.dirname(output_path), exist_ok=True)
            write_pcm16(output_path, wav.squeeze(0).to(torch.float32).numpy().astype("int16"), sampling_rate)
            print(f">> [output] saved to: {output_path}")
            return output_path
        return (sampling_rate, wav.type(torch.int16).numpy().T)

    def _vocode(self, latent, spk, lens=None):
        wav, _ = self.bigvgan(latent, speaker_embedding=spk, lens=lens)
        return torch.clamp(32767 * wav.squeeze(1), -32767.0, 32767.0)

    # utterances whose lengths are within this ratio share one (ragged) vocoder batch; the padding rows of the shorter ones
    # cost nothing (tiles past a row's end are not computed) but the batch allocates for the longest
    VOCODER_BUCKET_RATIO = 2.0

    def _vocode_ragged(self, lat, spk):
        """lat: list of [T_i, D] latents of one speaker.  Returns the list of waveforms [T_i * hop], each bit-identical to
        vocoding that utterance alone (BigVGAN.forward(lens=...)): utterances are sorted by length and vocoded in
        buckets of similar length -- one launch sequence per bucket instead of one per distinct length."""
        outs = [None] * len(lat)
        order = sorted(range(len(lat)), key=lambda i: int(lat[i].shape[0]))
        k = 0
        while k < len(order):
            t_min = int(lat[order[k]].shape[0])
            if t_min == 0:
                outs[order[k]] = torch.zeros(0, device=self.device)
                k += 1
                continue
            e = k
            while e < len(order) and int(lat[order[e]].shape[0]) <= self.VOCODER_BUCKET_RATIO * t_min:
                e += 1
            idx = order[k:e]
            lens = [int(lat[i].shape[0]) for i in idx]
            t_max = lens[-1]
            if lens[0] == t_max:
                wav = self._vocode(torch.stack([lat[i] for i in idx], 0), spk)
            else:
                x = torch.zeros(len(idx), t_max, lat[idx[0]].shape[1], dtype=lat[idx[0]].dtype, device=lat[idx[0]].device)
                for j, i in enumerate(idx):
                    x[j, : lens[j]] = lat[i]
                wav = self._vocode(x, spk, lens=lens)
            hop = wav.shape[1] // t_max
            for j, i in enumerate(idx):
                outs[i] = wav[j, : lens[j] * hop]
            k = e
        return outs

    # ------------------------------------------------------------------------------------------------ public API
    def infer(self, audio_prompt, text, output_path, verbose=False, max_text_tokens_per_sentence=120, speaker_id=None,
              **generation_kwargs):
        """Sentence-by-sentence synthesis (infer.py:779-917)."""
        if speaker_id is not None:
            if not self.speaker_list:
                raise ValueError("multi-speaker mode is not enabled; load speaker_info_path first")
            if speaker_id not in self.speaker_list:
                raise ValueError(f"invalid speaker_id: {speaker_id}")
        start_time = time.perf_counter()
        cond_mel = self._prompt(audio_prompt)
        self._set_gr_progress(0.1, "text processing...")
        sentences = self.tokenizer.split_sentences(self.tokenizer.tokenize(text), max_text_tokens_per_sentence)
        gen, max_mel_tokens = self._gen_kwargs(generation_kwargs)
        conds = self._conds(cond_mel, speaker_id)
        spk = self._spk(cond_mel)
        wavs, gpt_gen_time, gpt_forward_time, bigvgan_time, has_warned = [], 0.0, 0.0, 0.0, False
        for n, sent in enumerate(sentences, 1):
            text_tokens = torch.tensor(self.tokenizer.convert_tokens_to_ids(sent), dtype=torch.int32, device=self.device)[None]
            self._set_gr_progress(0.2 + 0.4 * (n - 1) / len(sentences), f"generating... {n}/{len(sentences)}")
            t0 = time.perf_counter()
            codes = self._generate(conds, text_tokens, gen, max_mel_tokens)
            torch.cuda.synchronize()
            gpt_gen_time += time.perf_counter() - t0
            if not has_warned and (codes[:, -1] != self.stop_mel_token).any():
                warnings.warn(f"generation stopped at max_mel_tokens ({max_mel_tokens}); consider shorter sentences",
                              category=RuntimeWarning)
                has_warned = True
            codes, code_lens = self.remove_long_silence(codes)
            if verbose:
                print(f">> codes {tuple(codes.shape)} lens {code_lens.tolist()}")
            self._set_gr_progress(0.2 + 0.4 * n / len(sentences), f"synthesising... {n}/{len(sentences)}")
            t0 = time.perf_counter()
            latent = self._latents(conds, [text_tokens[0]], [codes[0, : int(code_lens[0])]], reuse_prefix=True)[0][None]
            torch.cuda.synchronize()
            gpt_forward_time += time.perf_counter() - t0
            t0 = time.perf_counter()
            wav = self._vocode(latent, spk)
            torch.cuda.synchronize()
            bigvgan_time += time.perf_counter() - t0
            wavs.append(wav.cpu())
        self._set_gr_progress(0.9, "saving audio...")
        return self._finish(wavs, output_path, start_time, gpt_gen_time, gpt_forward_time, bigvgan_time)

    def infer_fast(self, audio_prompt, text, output_path, verbose=False, max_text_tokens_per_sentence=100,
                   sentences_bucket_max_size=4, **generation_kwargs):
        """Bucketed batch synthesis (infer.py:595-777): sentences of similar length are decoded as one left-padded batch;
        latents are computed in one batched pass and vocoded in time-concatenated pairs (chunk_size 2)."""
        print(">> [infer] fast mode")
        self._set_gr_progress(0, "initialising...")
        start_time = time.perf_counter()
        cond_mel = self._prompt(audio_prompt)
        sentences = self.tokenizer.split_sentences(self.tokenizer.tokenize(text), max_tokens_per_sentence=max_text_tokens_per_sentence)
        gen, max_mel_tokens = self._gen_kwargs(generation_kwargs)
        conds, spk = self._conds(cond_mel), self._spk(cond_mel)
        self._set_gr_progress(0.1, "text processing...")
        buckets = self.bucket_sentences(sentences, bucket_max_size=sentences_bucket_max_size)
        all_tokens = [[torch.tensor(self.tokenizer.convert_tokens_to_ids(it["sent"]), dtype=torch.int32, device=self.device)[None]
                       for it in b] for b in buckets]
        gpt_gen_time = gpt_forward_time = bigvgan_time = 0.0
        total = sum(len(b) for b in buckets)
        done, all_codes = 0, []
        for toks in all_tokens:
            batch = self.pad_tokens_cat(toks) if len(toks) > 1 else toks[0]
            done += len(toks)
            self._set_gr_progress(0.2 + 0.3 * done / total, f"GPT generating... {done}/{total}")
            t0 = time.perf_counter()
            all_codes.append(self._generate(conds, batch, gen, max_mel_tokens))
            torch.cuda.synchronize()
            gpt_gen_time += time.perf_counter() - t0
        self._set_gr_progress(0.5, "computing latents...")
        idxs, text_rows, code_rows, has_warned = [], [], [], False
        for codes_b, toks, b in zip(all_codes, all_tokens, buckets):
            for i in range(codes_b.shape[0]):
                codes = codes_b[i]
                if not has_warned and codes[-1] != self.stop_mel_token:
                    warnings.warn(f"generation stopped at max_mel_tokens ({max_mel_tokens})", category=RuntimeWarning)
                    has_warned = True
                codes, lens = self.remove_long_silence(codes[None])
                idxs.append(b[i]["idx"])
                text_rows.append(toks[i][0])
                code_rows.append(codes[0, : int(lens[0])])
        t0 = time.perf_counter()
        lat = self._latents(conds, text_rows, code_rows)
        torch.cuda.synchronize()
        gpt_forward_time += time.perf_counter() - t0
        lat = [lat[idxs.index(i)] for i in range(len(lat))]
        self._set_gr_progress(0.7, "vocoding...")
        wavs = []
        for i in range(0, len(lat), 2):
            t0 = time.perf_counter()
            wav = self._vocode(torch.cat(lat[i:i + 2], dim=0)[None], spk)
            torch.cuda.synchronize()
            bigvgan_time += time.perf_counter() - t0
            wavs.append(wav.cpu())
        self.torch_empty_cache()
        self._set_gr_progress(0.9, "saving audio...")
        return self._finish(wavs, output_path, start_time, gpt_gen_time, gpt_forward_time, bigvgan_time)

    def infer_batch(self, cond_mel: torch.Tensor, text_token_rows: List[torch.Tensor], max_mel_tokens=600, force_stop=None,
                    seed=1234, return_codes=False, phase_events: dict | None = None, **generation_kwargs):
        """Utterance-batch data path used by bench.py / the multi-GPU sharder (not in the reference API): one shared
        prompt, N independent texts decoded as ONE left-padded batch, one batched latent pass, and one vocoder call per
        group of equal-length utterances (batching unequal lengths would change the tail of the shorter waveforms).
        Returns a list of fp32 waveforms already scaled to the int16 range, like infer.py:892.
        phase_events, if given, receives torch.cuda.Event marks at the phase boundaries."""
        st = self._batch_tokens(cond_mel, text_token_rows, max_mel_tokens, force_stop, seed, phase_events, lazy_spk=True,
                                **generation_kwargs)
        outs = self._batch_waveforms(st, phase_events, reuse_prefix=True)   # serial: the KV cache still holds this batch's prompt
        return (outs, st["rows"]) if return_codes else outs

    def infer_queue(self, cond_mel: torch.Tensor, text_token_rows: List[torch.Tensor], slots=32, max_mel_tokens=600,
                    force_stop=None, seed=1234, return_codes=False, cache_positions=4096, check_every=16, staged=True,
                    phase_events: dict | None = None, **generation_kwargs):
        """Continuous batching (not in the reference API; SURVEY.md section 8e's mitigation for mixed output lengths): any
        number of utterances of one prompt through `slots` decode slots.  The longest texts start; whenever a row emits its
        stop token its codes are taken and its slot is refilled with the next utterance (GPTEngine.decode_refill: the new
        prompt is prefilled into that slot's KV rows right in front of the loop's current cache position, the row gets its
        own clock), so the token loop runs sum(lengths) / slots steps instead of, batch after batch, to each batch's longest
        row.  Then the latent pass and the vocoder run over groups of `slots` finished utterances of similar length.
        num_beams = 1 only.  cache_positions bounds the KV cache (one loop runs at most that many steps past its prompt; the
        queue continues in a fresh loop after that); check_every / staged: how often the loop looks for finished rows and whether
        a refill's prefill runs on a second stream under the loop's next steps (GPTEngine.decode_refill).  Returns the waveforms
        in the order of text_token_rows, as infer_batch."""
        gen, _ = self._gen_kwargs(generation_kwargs)
        if int(gen.get("num_beams", 1)) != 1:
            raise NotImplementedError("infer_queue: num_beams = 1 only (beam rows cannot be refilled one at a time)")
        if not text_token_rows:
            return ([], []) if return_codes else []
        if int(slots) < 1:
            raise ValueError("infer_queue: slots must be >= 1")
        self._mark(phase_events, "start")
        bf = self._batch_feat
        if bf is not None and bf[0] is cond_mel and bf[1] == cond_mel._version:
            conds, spk = bf[2], bf[3]
        else:
            conds, spk = self._prompt_features(cond_mel)
            self._batch_feat = [cond_mel, cond_mel._version, conds, spk]
        if spk is None:
            spk = self._batch_feat[3] = self._prompt_spk(cond_mel)
        g, eng = self.gpt, self.gpt.engine
        N = len(text_token_rows)
        texts = [t.reshape(-1).to(torch.int32).cpu() for t in text_token_rows]
        stops = [-1] * N if force_stop is None else [int(v) for v in force_stop]
        sp = dict(do_sample=bool(gen["do_sample"]), top_p=float(gen["top_p"]), top_k=int(gen["top_k"]),
                  temperature=float(gen["temperature"]), repetition_penalty=float(gen["repetition_penalty"]), seed=int(seed))
        if not sp["do_sample"]:
            sp["top_p"], sp["top_k"], sp["temperature"] = 1.0, 0, 1.0
        stop_text = self.cfg.gpt.stop_text_token

        def prefixes(ids):
            """(left-padded prefix batch, pad) of the utterances `ids` -- one itts_prefix_rows launch"""
            L = max(int(texts[i].numel()) for i in ids)
            bh = torch.full((len(ids), L), stop_text, dtype=torch.int32)
            for j, i in enumerate(ids):
                bh[j, : texts[i].numel()] = texts[i]
            return g.prefix_rows(conds, bh)

        queue = sorted(range(N), key=lambda i: -int(texts[i].numel()))     # longest first: every later prompt fits
        rows: List[torch.Tensor | None] = [None] * N
        self._mark(phase_events, "conditioned")
        while queue:
            first, queue = queue[:slots], queue[slots:]
            emb, pad = prefixes(first)
            ce = int(check_every)
            if eng.paged:
                # paged cache: one loop serves the whole queue -- a slot's blocks go back to the pool when its row stops.  The pool
                # holds `slots` windows of (longest prompt + start token + max_mel_tokens + the steps up to the second next poll)
                eng.prefill(emb, pad, max_mel_tokens + ce + 1, shared_rows=int(conds.shape[1]) if conds.shape[0] == 1 else 0,
                            slots_window=emb.shape[1] + 1 + max_mel_tokens + 2 * ce + 2)
            else:
                # contiguous strips: the loop runs whole blocks of check_every steps, the cache has to hold that many positions
                # past max_mel_tokens; cache positions only grow, so a loop ends when `cache_positions` are used up
                eng.prefill(emb, pad, max(max_mel_tokens + ce + 1, int(cache_positions) - emb.shape[1] - 2),
                            shared_rows=int(conds.shape[1]) if conds.shape[0] == 1 else 0, paged=False)
            entered = list(first)

            def feed(k):
                nonlocal queue
                take, queue = queue[:k], queue[k:]
                if not take:
                    return []
                e, p = prefixes(take)
                p = p.tolist()
                entered.extend(take)
                return [(e[j, p[j]:], stops[i]) for j, i in enumerate(take)]

            codes, leftover = eng.decode_refill(max_mel_tokens, sp, feed, force_stop=[stops[i] for i in first],
                                                positions=None if eng.kv is not None else
                                                max(int(cache_positions), eng._S + max_mel_tokens + int(check_every) + 1),
                                                check_every=int(check_every), staged=bool(staged))
            if leftover:                                               # fed but not placed: back to the head of the queue
                back = entered[len(entered) - len(leftover):]
                entered = entered[: len(entered) - len(leftover)]
                queue = back + queue
            for i, c in zip(entered, codes):
                rows[i] = c
        self._mark(phase_events, "decoded")
        squeezed = []
        for c in rows:
            cc, ln = self.remove_long_silence(c[None])
            squeezed.append(cc[0, : int(ln[0])].cpu())
        outs: List[torch.Tensor | None] = [None] * N
        order = sorted(range(N), key=lambda i: int(squeezed[i].numel()))
        for k in range(0, N, slots):
            ids = order[k: k + slots]
            lat = self._latents(conds, [texts[i] for i in ids], [squeezed[i] for i in ids])
            for i, w in zip(ids, self._vocode_ragged(lat, spk)):
                outs[i] = w
        self._mark(phase_events, "vocoded")
        return (outs, squeezed) if return_codes else outs

    @staticmethod
    def _mark(phase_events, name):
        if phase_events is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            phase_events[name] = e

    def _batch_tokens(self, cond_mel, text_token_rows, max_mel_tokens=600, force_stop=None, seed=1234, phase_events=None,
                      lazy_spk=False, **generation_kwargs):
        """Stage A of infer_batch: prompt conditioning -> prefill -> sampling loop -> silence squeeze (host).  Everything
        here is latency-bound small launches; it ends with the codes on the host, as infer.py:848-861 does."""
        gen, _ = self._gen_kwargs(generation_kwargs)
        self._mark(phase_events, "start")
        # One prompt serves many batches (one speaker, a stream of texts): its conditioning latents and speaker embedding are
        # kept per prompt TENSOR -- the same object with an unchanged version counter is the same prompt, as infer() keeps
        # them per prompt path (the cache holds a reference, so the storage cannot be recycled under it).
        bf = self._batch_feat
        if bf is not None and bf[0] is cond_mel and bf[1] == cond_mel._version:
            conds, spk = bf[2], bf[3]
        else:
            # the token loop needs the conditioning latents only; the speaker embedding (ECAPA, vocoder input) is computed
            # when stage B asks for it (lazy_spk) -- it is not on the first token's critical path
            conds, spk = self._prompt_conds(cond_mel), (None if lazy_spk else self._prompt_spk(cond_mel))
            self._batch_feat = [cond_mel, cond_mel._version, conds, spk]
        L = max(int(t.numel()) for t in text_token_rows)
        stop = self.cfg.gpt.stop_text_token
        batch_h = torch.full((len(text_token_rows), L), stop, dtype=torch.int32)
        if text_token_rows and text_token_rows[0].is_cuda:
            text_token_rows = [t.cpu() for t in text_token_rows]
        for i, t in enumerate(text_token_rows):
            batch_h[i, : t.numel()] = t.reshape(-1).to(torch.int32)
        g = self.gpt
        emb, pad = g.prefix_rows(conds, batch_h)   # one upload for the whole batch; the padding stays a host tensor
        sp = dict(do_sample=bool(gen["do_sample"]), top_p=float(gen["top_p"]), top_k=int(gen["top_k"]),
                  temperature=float(gen["temperature"]), repetition_penalty=float(gen["repetition_penalty"]), seed=int(seed))
        if not sp["do_sample"]:
            sp["top_p"], sp["top_k"], sp["temperature"] = 1.0, 0, 1.0
        self._mark(phase_events, "conditioned")
        nb = int(gen.get("num_beams", 1))
        shared = int(conds.shape[1]) if conds.shape[0] == 1 else 0   # one prompt: every row starts with the same latents
        if nb > 1:
            if force_stop is not None:
                raise NotImplementedError("force_stop is a measurement aid of the num_beams=1 loop")
            sp["length_penalty"] = float(gen.get("length_penalty", 0.0))
            if g.engine.beam_kv == "table":    # the prompt is computed and cached once per batch element (row table)
                g.engine.prefill(emb, pad, max_mel_tokens, beams=nb, shared_rows=shared)
            else:                              # generate() expands every row to num_beams copies before the first forward
                g.engine.prefill(emb.repeat_interleave(nb, dim=0), pad.repeat_interleave(nb), max_mel_tokens, shared_rows=shared, paged=False)
            self._mark(phase_events, "prefilled")
            codes = g.engine.decode_beam(max_mel_tokens, sp, nb)
        else:
            g.engine.prefill(emb, pad, max_mel_tokens, shared_rows=shared)
            self._mark(phase_events, "prefilled")
            codes = g.engine.decode(max_mel_tokens, sp, force_stop=force_stop)
        self._mark(phase_events, "decoded")
        codes_np, lens_h = self._squeeze_silence_host(codes)   # infer.py:848-861 on the host copy: one device-to-host transfer in all
        codes_h = torch.from_numpy(codes_np)
        rows = [codes_h[i, : lens_h[i]] for i in range(codes_h.shape[0])]
        # where the latent pass finds each element's prompt in the KV cache: row b, or row b * num_beams when the beam prefill
        # expanded the rows (beam_kv = "copy")
        crows = [b * nb for b in range(len(rows))] if nb > 1 and g.engine.beam_kv != "table" else None
        return dict(conds=conds, spk=spk, rows=rows, texts=[t.reshape(-1) for t in text_token_rows], cache_rows=crows,
                    cond_mel=cond_mel)

    def _batch_waveforms(self, st, phase_events=None, reuse_prefix=False):
        """Stage B of infer_batch: batched teacher-forced latent pass + vocoder (large MFMA-bound launches, no host sync).
        With reuse_prefix = False it touches no decode-loop state, so it may run on another stream beside the next batch's
        stage A (BatchPipeline, whose prefill overwrites the KV cache: it must not reuse the prompt's keys / values)."""
        conds, spk = st["conds"], st["spk"]
        if spk is None:                                   # lazy_spk: first batch of this prompt
            spk = self._prompt_spk(st["cond_mel"])
            bf = self._batch_feat
            if bf is not None and bf[0] is st["cond_mel"] and bf[2] is conds:
                bf[3] = spk
        lat = self._latents(conds, st["texts"], st["rows"], reuse_prefix=reuse_prefix, cache_rows=st.get("cache_rows"))
        self._mark(phase_events, "latents")
        outs = self._vocode_ragged(lat, spk)
        self._mark(phase_events, "vocoded")
        return outs


class BatchTicket:
    """Handle of one batch in flight in a BatchPipeline: result() waits for its vocoder stage and returns the waveforms."""

    def __init__(self, rows, keep):
        self.rows, self._keep = rows, keep
        self._ready = threading.Event()
        self._outs = self._done = self._err = None

    def _set(self, outs, done, err=None):
        self._outs, self._done, self._err = outs, done, err
        self._ready.set()

    def result(self):
        self._ready.wait()
        if self._err is not None:
            raise self._err
        self._done.synchronize()
        self._keep = None
        return self._outs


class BatchPipeline:
    """Two-stage software pipeline over utterance batches (serving schedule; not in the reference API).

    Stage A of a batch (conditioning, prefill, the token loop: ~7 small launches per block per token, latency-bound, the
    matrix cores idle) runs on the caller's thread; its stage B (latent pass + vocoder: large MFMA-bound launches) is
    enqueued by a worker thread on a second HIP stream and overlaps stage A of the NEXT batch.  Results are identical
    to infer_batch(): the two stages share only read-only weights (the latent pass and the vocoder allocate their
    activations per call from the stream-aware allocator, the decode loop owns the KV cache and its state)."""

    def __init__(self, tts: "IndexTTS"):
        self.tts = tts
        lo, hi = 0, -1
        self.stream_a = torch.cuda.Stream(device=tts.device, priority=hi)   # latency-critical token loop
        self.stream_b = torch.cuda.Stream(device=tts.device, priority=lo)   # throughput work
        self._jobs: "queue.Queue" = queue.Queue()
        self._inflight: List[BatchTicket] = []
        self._thread = threading.Thread(target=self._worker, name="itts-stage-b", daemon=True)
        self._thread.start()

    def _worker(self):
        torch.cuda.set_device(self.tts.device)
        while True:
            job = self._jobs.get()
            if job is None:
                return
            st, ticket, a_done = job
            try:
                # graph captures (global error mode) must not see another thread allocating or synchronising
                with nat.CAPTURE_LOCK, torch.no_grad(), torch.cuda.stream(self.stream_b):
                    self.stream_b.wait_event(a_done)
                    outs = self.tts._batch_waveforms(st)
                    done = torch.cuda.Event()
                    done.record()
                ticket._set(outs, done)
            except BaseException as e:  # noqa: BLE001  (surfaced by result())
                ticket._set(None, None, e)

    def submit(self, cond_mel, text_token_rows, **kw) -> BatchTicket:
        """Runs stage A (returns once the codes are on the host) and hands stage B to the worker."""
        self.stream_a.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream_a):
            st = self.tts._batch_tokens(cond_mel, text_token_rows, **kw)
            for t in (st["conds"], st["spk"]):
                t.record_stream(self.stream_b)
            a_done = torch.cuda.Event()
            a_done.record()
        ticket = BatchTicket(st["rows"], st)
        self._inflight = [t for t in self._inflight if not t._ready.is_set()] + [ticket]
        self._jobs.put((st, ticket, a_done))
        return ticket

    def drain(self):
        for t in self._inflight:
            t._ready.wait()
        self._inflight.clear()
        self.stream_a.synchronize()
        self.stream_b.synchronize()

    def close(self):
        self._jobs.put(None)
        self._thread.join()


class RequestPool:
    """Serving concurrency (not in the reference API): several independent utterance batches in flight on one GPU, each on
    its own IndexTTS instance, host thread and HIP stream.  A single request's token loop is a chain of ~170 dependent
    microsecond-scale launches per token and leaves most CUs idle; the loops of different requests interleave on the
    chip (measured on MI355X, batch 32: 1.4x the audio-seconds/s of one request at a time with two in flight).
    Instances come from IndexTTS.replica() (shared read-only weights, private KV cache / state / graphs) or are
    independent objects.  Results are those of infer_batch() on one instance."""

    class _Job:
        def __init__(self):
            self.ready, self.out, self.err = threading.Event(), None, None

        def result(self):
            self.ready.wait()
            if self.err is not None:
                raise self.err
            return self.out

    @classmethod
    def of(cls, tts: "IndexTTS", inflight: int = 2) -> "RequestPool":
        return cls([tts] + [tts.replica() for _ in range(max(1, inflight) - 1)])

    def __init__(self, instances: List["IndexTTS"]):
        """Every instance gets an ordinary HIP stream of its own (streams restricted to disjoint CU subsets were measured in
        round 3 and never beat letting the dispatcher interleave the requests: profiles/r03_pool_cu_masks.txt).  The HIP runtime
        multiplexes a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4, read when the runtime initialises):
        with more requests in flight than queues two of them share one and run back to back -- export GPU_MAX_HW_QUEUES >=
        len(instances) before the process first touches the GPU (bench.py does for its 4-deep leg: 1 740 instead of 1 400
        audio-s/s)."""
        assert instances, "need at least one instance"
        self.instances = list(instances)
        hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4") or 4)
        if len(self.instances) > hwq:
            warnings.warn(f"RequestPool: {len(self.instances)} requests in flight over GPU_MAX_HW_QUEUES={hwq} hardware queues: "
                          f"some request streams will share a queue (export GPU_MAX_HW_QUEUES={len(self.instances)} before the "
                          f"process initialises the GPU)", RuntimeWarning)
        self._queues = [queue.Queue() for _ in self.instances]
        self._threads = [threading.Thread(target=self._run, args=(i,), name=f"itts-request-{i}", daemon=True)
                         for i in range(len(self.instances))]
        self._next = 0
        for t in self._threads:
            t.start()

    def _run(self, i):
        inst, q = self.instances[i], self._queues[i]
        torch.cuda.set_device(inst.device)
        stream = torch.cuda.Stream(device=inst.device)
        while True:
            item = q.get()
            if item is None:
                return
            job, args, kw = item
            try:
                with torch.no_grad(), torch.cuda.stream(stream):
                    job.out = inst.infer_batch(*args, **kw)
                    stream.synchronize()
            except BaseException as e:  # noqa: BLE001  (surfaced by result())
                job.err = e
            job.ready.set()

    def warm_up(self, *args, rounds=2, **kw):
        """Run `rounds` requests on every instance, ONE INSTANCE AT A TIME: the first calls capture CUDA graphs, and a
        capture (global error mode) must not see another thread allocating or synchronising.  Call it once for every
        (batch size, generation settings, prompt length) combination that will be served concurrently; a capture that
        happens later, while other requests are running, fails that request loudly (result() raises)."""
        for i in range(len(self.instances)):
            for _ in range(rounds):
                self.submit(*args, _instance=i, **kw).result()

    def submit(self, cond_mel, text_token_rows, _instance=None, **kw):
        """Queue one infer_batch(cond_mel, text_token_rows, **kw) on the next instance (round robin); returns a handle
        whose result() blocks until the waveforms are complete."""
        i = self._next % len(self.instances) if _instance is None else _instance
        if _instance is None:
            self._next += 1
        job = RequestPool._Job()
        self._queues[i].put((job, (cond_mel, text_token_rows), kw))
        return job

    def close(self):
        for q in self._queues:
            q.put(None)
        for t in self._threads:
            t.join()
