"""UnifiedVoice: the IndexTTS GPT wrapper (drop-in surface of indextts/gpt/model.py:312-720, inference only).

Kept from the reference: constructor keywords (`UnifiedVoice(**cfg.gpt)`), `load_state_dict/state_dict` with the
reference key names, `.eval()/.half()/.bfloat16()/.to()`, `post_init_gpt2_config(use_deepspeed, kv_cache, half)`,
`get_conditioning`, `prepare_gpt_inputs`, `inference_speech(...)` (HF-generate keyword arguments) and
`forward(..., return_latent=True)`.  The transformer itself runs in GPTEngine (HIP kernels); the Conformer+Perceiver
conditioner is host-side PyTorch-ROCm.  Training paths (losses, LoRA) are out of scope.
"""
from __future__ import annotations

import os
import warnings

import torch
import torch.nn.functional as F

from .. import _native as nat
from .conditioner import ConditionerEngine
from .conformer_encoder import conformer_encode
from .engine import GPTEngine
from .perceiver import perceiver_resample


class UnifiedVoice:
    def __init__(self, layers=8, model_dim=512, heads=8, max_text_tokens=120, max_mel_tokens=250,
                 max_conditioning_inputs=1, mel_length_compression=1024, number_text_tokens=256, start_text_token=0,
                 stop_text_token=1, number_mel_codes=8194, start_mel_token=8192, stop_mel_token=8193,
                 train_solo_embeddings=False, use_mel_codes_as_input=True, checkpointing=True, types=1,
                 activation_function=None, condition_num_latent=32, condition_type="perceiver", condition_module=None):
        if condition_type != "conformer_perceiver":
            raise NotImplementedError("only condition_type='conformer_perceiver' is on the IndexTTS inference path")
        if activation_function not in (None, "gelu_new"):
            raise NotImplementedError("only gelu_new is supported")
        self.layers, self.model_dim, self.heads = layers, model_dim, heads
        self.max_text_tokens, self.max_mel_tokens = max_text_tokens, max_mel_tokens
        self.max_conditioning_inputs = max_conditioning_inputs
        self.mel_length_compression = mel_length_compression
        self.number_text_tokens, self.number_mel_codes = number_text_tokens, number_mel_codes
        self.start_text_token, self.stop_text_token = start_text_token, stop_text_token
        self.start_mel_token, self.stop_mel_token = start_mel_token, stop_mel_token
        self.condition_type, self.cond_num = condition_type, condition_num_latent
        self.condition_module = dict(condition_module or {})
        self.mean_condition = None
        self.device = torch.device("cpu")
        self.dtype = torch.float32
        self._sd = {}
        self._cond_w = None
        self._cond_engine = None
        self.engine: GPTEngine | None = None
        self.inference_model = None

    # ---- nn.Module-like surface -------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=False):
        self._sd = {k: v.detach() for k, v in sd.items() if not k.startswith("inference_model.")}
        self._cond_w, self._cond_engine, self.engine = None, None, None
        return self

    def state_dict(self):
        return dict(self._sd)

    def parameters(self):
        return iter(self._sd.values())

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                self.dtype = a
            elif isinstance(a, (str, torch.device)):
                self.device = torch.device(a)
        self._cond_w, self._cond_engine, self.engine = None, None, None
        return self

    def half(self):
        # the HIP GPT kernels implement fp32 and bf16; a request for fp16 is served in bf16 (same storage width,
        # fp32 accumulation) -- the reference makes the same substitution when bf16 is available (infer.py:287-293)
        return self.to(torch.bfloat16)

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def float(self):
        return self.to(torch.float32)

    def eval(self):
        return self

    def post_init_gpt2_config(self, use_deepspeed=False, kv_cache=False, half=False):
        """Reference: builds GPT2InferenceModel (model.py:395-432).  Here: packs the weights into the HIP engine."""
        if self.device.type != "cuda":
            raise RuntimeError("UnifiedVoice runs on the HIP kernels only: move it to a cuda device (no CPU fallback)")
        need = ["gpt.ln_f.weight", "mel_head.weight", "mel_embedding.weight", "text_embedding.weight"]
        missing = [k for k in need if k not in self._sd]
        if missing:
            raise RuntimeError(f"checkpoint is missing {missing}")
        self.engine = GPTEngine(self._sd, self.layers, self.model_dim, self.heads, dtype=self.dtype, device=self.device,
                                start_mel_token=self.start_mel_token, stop_mel_token=self.stop_mel_token)
        self.inference_model = self.engine
        return self

    def _cond_weights(self):
        if self._cond_w is None:
            self._cond_w = {k: v.to(self.device, torch.float32) for k, v in self._sd.items()
                            if k.startswith(("conditioning_encoder.", "perceiver_encoder.")) and not k.endswith("pos_enc.pe")}
        return self._cond_w

    def conditioner(self):
        """The Conformer + Perceiver conditioner on the HIP kernels (gpt/conditioner.py), or None in fp32 mode: a 16-bit model
        (the benched and the reference's GPU default precision) runs it in fp16 -- normalised activations, fp32 residual stream --
        whatever the transformer's 16-bit type is; fp32 keeps the functional PyTorch form, the parity mode."""
        if self.dtype == torch.float32 or self.device.type != "cuda" or os.environ.get("ITTS_NATIVE_CONDITIONER", "1") == "0":
            return None
        if self._cond_engine is None:
            self._cond_engine = ConditionerEngine(self._cond_weights(), heads=int(self.condition_module.get("attention_heads", 8)),
                                                  dtype=torch.float16, device=self.device)
        return self._cond_engine

    # ---- conditioning / prefix --------------------------------------------------------------------------------
    def get_conditioning(self, speech_conditioning_input, cond_mel_lengths=None, speaker_ids=None):
        """model.py:487-546 (conformer_perceiver branch, plus the stored mean_condition_{id} shortcut)."""
        if speaker_ids is not None and speech_conditioning_input is None:
            out = []
            for sid in speaker_ids:
                c = getattr(self, f"mean_condition_{sid}", None)
                if c is None:
                    raise ValueError(f"no stored condition for speaker {sid}")
                c = c.to(self.device, torch.float32)
                out.append(c[None] if c.ndim == 2 else (c[0] if c.ndim == 4 else c))
            return torch.cat(out, dim=0)
        if self.mean_condition is not None and speech_conditioning_input is None:
            return self.mean_condition.to(self.device).expand(1, -1, -1)
        mel = speech_conditioning_input.to(self.device, torch.float32)
        if mel.ndim == 2:
            mel = mel[None]
        eng = self.conditioner() if cond_mel_lengths is None else None
        if eng is not None:                  # unpadded prompts: the HIP conditioner, one prompt per pass
            rows = mel.transpose(1, 2).contiguous()
            return torch.stack([eng(rows[i]).clone() for i in range(rows.shape[0])], 0)
        W = self._cond_weights()
        # cond_mel_lengths = None: every row is as long as the tensor (model.py:491 builds exactly that) -- no padding, the
        # mask operations of the two networks are identities and are skipped
        x, mask = conformer_encode(W, mel.transpose(1, 2), None if cond_mel_lengths is None else cond_mel_lengths.to(self.device),
                                   heads=int(self.condition_module.get("attention_heads", 8)))
        cmask = None if mask is None else F.pad(mask.squeeze(1), (self.cond_num, 0), value=True)
        return perceiver_resample(W, x, cmask, heads=int(self.condition_module.get("attention_heads", 8)))

    def prepare_gpt_inputs(self, conditional_latents, text_inputs):
        """model.py:606-667 -> (fake_inputs [B,P+1], prefix_emb [B,P,D] fp32, attention_mask [B,P+1])."""
        eng = self.engine
        dev = self.device
        t = text_inputs.to(dev).long().contiguous()
        B, L = t.shape
        C = conditional_latents.shape[1]
        P = C + L + 2
        c = conditional_latents.to(dev, torch.float32).contiguous()
        # strip start / stop ids, then start | tokens | stop, embedded behind the latents and left-padded to P positions
        # (model.py:630-649): one launch, no host round trip (itts_prefix_rows)
        emb, mask, _ = nat.prefix_rows(t, c, eng.text_emb, eng.text_pos, self.start_text_token, self.stop_text_token)
        fake = torch.ones(B, P + 1, dtype=torch.long, device=dev)
        fake[:, -1] = self.start_mel_token
        return fake, emb, mask

    def prefix_rows(self, conditional_latents, text_inputs):
        """What the engine needs of prepare_gpt_inputs: (prefix_emb [B,P,D] fp32, left padding per row int32 [B]) -- the same
        launch, without the fake ids and without re-deriving the padding from the mask.  The padding comes back as a HOST tensor
        when the ids were given on the host (no device round trip in front of the prefill), else as the kernel's device tensor."""
        pad_host = None
        if not text_inputs.is_cuda:
            # ids still on the host: the padding (L - ids kept) is known here, and GPTEngine.prefill() -- which sizes its launches
            # from it -- need not wait for the device to hand it back
            th = text_inputs.long()
            keep = (th != self.stop_text_token) & (th != self.start_text_token)
            pad_host = (th.shape[1] - keep.sum(dim=1)).to(torch.int32)
        t = text_inputs.to(self.device).long().contiguous()
        c = conditional_latents.to(self.device, torch.float32).contiguous()
        emb, _, pad = nat.prefix_rows(t, c, self.engine.text_emb, self.engine.text_pos, self.start_text_token, self.stop_text_token)
        return emb, (pad if pad_host is None else pad_host)

    # ---- generation -------------------------------------------------------------------------------------------
    def inference_speech(self, speech_conditioning_mel, text_inputs, cond_mel_lengths=None, input_tokens=None,
                         num_return_sequences=1, max_generate_length=None, typical_sampling=False, typical_mass=.9,
                         speaker_ids=None, force_stop=None, seed=None, return_logits=False, **hf):
        """model.py:669-720.  Accepted generate() keywords: do_sample, top_p, top_k, temperature, repetition_penalty,
        num_beams, length_penalty.  Returns codes [B * num_return_sequences, n] (stop-token padded), like
        `output[:, trunc_index:]`: with beams the num_return_sequences best hypotheses of each element, best first; with
        sampling that many independent draws per element."""
        if self.engine is None:
            raise RuntimeError("call post_init_gpt2_config() first")
        if input_tokens is not None:
            raise NotImplementedError("input_tokens (continuing a given code prefix) is off the infer.py path")
        num_beams = int(hf.pop("num_beams", 1))
        if typical_sampling:
            # model.py:704-708; never enabled by infer.py / cli.py / api.py (out of scope)
            raise NotImplementedError("typical_sampling is off the infer.py path and not built (oracle/sampling_ref.typical restates it)")
        nrs = int(num_return_sequences)
        if nrs < 1 or (num_beams > 1 and nrs > num_beams):
            raise ValueError("num_return_sequences has to be in [1, num_beams]")
        if nrs > 1 and num_beams == 1 and not bool(hf.get("do_sample", False)):
            raise ValueError("greedy decoding returns one sequence: num_return_sequences > 1 needs do_sample=True or num_beams > 1")
        length_penalty = float(hf.pop("length_penalty", 1.0))  # HF default 1.0; infer.py passes 0.0
        sp = dict(do_sample=bool(hf.pop("do_sample", False)), top_p=float(hf.pop("top_p", 1.0)),
                  top_k=int(hf.pop("top_k", 50)), temperature=float(hf.pop("temperature", 1.0)),
                  repetition_penalty=float(hf.pop("repetition_penalty", 1.0)),
                  seed=int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF) if seed is None else int(seed))
        if not sp["do_sample"]:
            sp["top_p"], sp["top_k"], sp["temperature"] = 1.0, 0, 1.0
        elif sp["top_k"] <= 0:
            raise ValueError("do_sample=True needs top_k >= 1: the device sampler keeps at most 1024 candidates per row "
                             "(128 per beam) and refuses to truncate an unrestricted distribution silently")
        if hf:
            raise TypeError(f"unsupported generate() arguments: {sorted(hf)}")
        conds = self.get_conditioning(speech_conditioning_mel, cond_mel_lengths, speaker_ids=speaker_ids)
        emb, pad = self.prefix_rows(conds, text_inputs)
        shared = int(conds.shape[1]) if conds.shape[0] == 1 else 0   # one prompt: every row starts with the same latents
        max_new = (self.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
        if num_beams > 1:
            # generate() expands every row to num_beams identical rows before the first forward (beam search / beam-sample)
            if force_stop is not None or return_logits:
                raise NotImplementedError("force_stop / return_logits are measurement aids of the num_beams=1 loop")
            sp["length_penalty"] = length_penalty
            if self.engine.beam_kv == "table":   # prompt computed and cached once per batch element (row table)
                self.engine.prefill(emb, pad, max_new, beams=num_beams, shared_rows=shared)
            else:
                self.engine.prefill(emb.repeat_interleave(num_beams, dim=0), pad.repeat_interleave(num_beams), max_new, shared_rows=shared,
                                    paged=False)
            return self.engine.decode_beam(max_new, sp, num_beams, num_return_sequences=nrs)
        if nrs > 1:
            # sampling: generate() expands every row to num_return_sequences copies before the first forward
            # (_expand_inputs_for_generation); row b * nrs + j is the j-th independent draw of element b
            emb, pad = emb.repeat_interleave(nrs, dim=0), pad.repeat_interleave(nrs)
            if force_stop is not None:
                force_stop = [v for v in force_stop for _ in range(nrs)]
        self.engine.prefill(emb, pad, max_new, shared_rows=shared)
        out = self.engine.decode(max_new, sp, force_stop=force_stop, return_logits=return_logits)
        return out

    def attach_lora(self, adapters: dict | None, scaling: float = 1.0):
        """Unmerged LoRA adapters at run time (None detaches; peft tensors of the Conv1D targets attn.c_attn / attn.c_proj / mlp.c_fc /
        mlp.c_proj, train.py:555-563): see GPTEngine.attach_lora.  The reference itself only ever loads merged weights."""
        if self.engine is None:
            raise RuntimeError("call post_init_gpt2_config() first")
        self.engine.attach_lora(adapters, scaling)
        return self

    def replica(self) -> "UnifiedVoice":
        """Same weights (shared tensors), separate decode state: see GPTEngine.fork()."""
        import copy
        r = copy.copy(self)
        r.engine = self.engine.fork()
        r.inference_model = r.engine
        return r

    # ---- teacher-forced latent pass -----------------------------------------------------------------------------
    def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, wav_lengths,
                cond_mel_lengths=None, types=None, text_first=True, raw_mels=None, return_attentions=False,
                return_latent=False, clip_inputs=False, speaker_ids=None, conds=None):
        """model.py:548-597 with return_latent=True (the only mode infer.py uses): -> latent [B, T, D] fp32."""
        if not return_latent:
            raise NotImplementedError("training losses are out of scope; call with return_latent=True")
        eng = self.engine
        dev = self.device
        if conds is None:
            conds = self.get_conditioning(speech_conditioning_latent, cond_mel_lengths, speaker_ids)
        text_inputs = text_inputs.to(dev).long()
        mel_codes = mel_codes.to(dev).long()
        B = text_inputs.shape[0]
        text_lengths = torch.as_tensor(text_lengths).to(dev).long().reshape(-1)
        wav_lengths = torch.as_tensor(wav_lengths).to(dev).reshape(-1)
        mel_len = torch.ceil(wav_lengths.float() / self.mel_length_compression).long() + 1
        T = mel_codes.shape[1]
        # set_mel_padding / set_text_padding (:439-457), stop pads (:576-577), aligned inputs (:580-588)
        ar_t = torch.arange(text_inputs.shape[1], device=dev)[None]
        text_inputs = torch.where(ar_t < text_lengths[:, None], text_inputs, torch.full_like(text_inputs, self.stop_text_token))
        ar_m = torch.arange(T, device=dev)[None]
        mel_codes = torch.where(ar_m < mel_len[:, None], mel_codes, torch.full_like(mel_codes, self.stop_mel_token))
        ti = F.pad(F.pad(text_inputs, (0, 1), value=self.stop_text_token), (1, 0), value=self.start_text_token)
        mi = F.pad(F.pad(mel_codes, (0, 1), value=self.stop_mel_token), (1, 0), value=self.start_mel_token)
        te = eng.text_emb[ti] + eng.text_pos[: ti.shape[1]]
        me = eng.mel_emb[mi] + eng.mel_pos[: mi.shape[1]]
        c = conds.to(dev, torch.float32)
        if c.shape[0] == 1 and B > 1:
            c = c.expand(B, -1, -1)
        emb = torch.cat([c, te, me], dim=1)
        enc = eng.latent(emb)[:, c.shape[1]:]
        mel_part = enc[:, -mi.shape[1]:]
        return mel_part[:, :-2]

    __call__ = forward
