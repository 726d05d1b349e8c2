"""Conditioned BigVGAN generator on the HIP kernels (drop-in for indextts/BigVGAN/models.py:130-262, inference only).

API kept from the reference: `BigVGAN(h, use_cuda_kernel=False)`, `.load_state_dict(sd)`, `.to()/.half()/.float()`,
`.remove_weight_norm()`, `.eval()`, `forward(x[B,T,gpt_dim], mel_ref[B,Tref,num_mels], lens=None) -> (wav[B,1,T*1024], None)`.

Design (MI355X-first, not a translation): activations are channels-last [B][T][C] so that every convolution is an
implicit GEMM over (tap, channel) with time on the MFMA M dimension; each transposed-conv upsampler becomes a 1- or
2-tap convolution over the INPUT grid producing u*Cout columns that land contiguously in the upsampled tensor
(t' + pad = q*u + s  =>  y[t'] = x[q] W[..,s] + x[q-1] W[..,s+u]); the residual add of each AMP block and the 1/3 mean
over the three blocks are conv epilogues; the anti-aliased SnakeBeta activation is one fused kernel.  The ECAPA-TDNN
speaker encoder and the 1x1 conditioning projections (a few MFLOP, once per call) stay in host PyTorch.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn.functional as F

from .. import _native as nat
from .ECAPA_TDNN import ecapa_embed
from .speaker_engine import SpeakerEngine


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, kernel_size=12) -> np.ndarray:
    """Kaiser-windowed sinc low-pass (alias_free_torch/filter.py:29-58), used when a checkpoint carries no filter buffers."""
    half = kernel_size // 2
    A = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    t = np.arange(-half, half) + 0.5
    f = 2 * cutoff * np.kaiser(kernel_size, beta) * np.sinc(2 * cutoff * t)
    return (f / f.sum()).astype(np.float32)


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (what remove_weight_norm leaves behind, models.py:254-262)."""
    n = v.float().flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v.float() * (g.float() / n)


def convtr_as_conv(w: torch.Tensor, u: int):
    """ConvTranspose1d weight [Cin,Cout,k] (stride u, padding (k-u)//2, k in {u, 2u}) ->
    (taps[n_taps][Cin][u*Cout], off0, y_shift) for itts_gemm_conv."""
    Cin, Cout, k = w.shape
    pad = (k - u) // 2
    if k == u:
        return w.permute(0, 2, 1).reshape(1, Cin, u * Cout).contiguous(), 0, -pad * Cout
    if k != 2 * u:
        raise ValueError(f"unsupported transposed conv: kernel {k}, stride {u}")
    lo = w[:, :, :u].permute(0, 2, 1).reshape(Cin, u * Cout)  # multiplies x[q]
    hi = w[:, :, u:].permute(0, 2, 1).reshape(Cin, u * Cout)  # multiplies x[q-1]
    return torch.stack([hi, lo], 0).contiguous(), -1, -pad * Cout


class BigVGAN:
    def __init__(self, h, use_cuda_kernel=False):
        self.h = h
        self.use_cuda_kernel = use_cuda_kernel  # accepted for API compatibility; the HIP kernels are always used
        if str(h.get("resblock", "1")) != "1":
            raise NotImplementedError("only resblock '1' (AMPBlock1) is on the IndexTTS inference path")
        if h.get("activation", "snakebeta") != "snakebeta" or not h.get("snake_logscale", True):
            raise NotImplementedError("only log-scale SnakeBeta is supported")
        if h.get("feat_upsample", False):
            raise NotImplementedError("feat_upsample=True is not used by IndexTTS")
        self.rates = list(h["upsample_rates"])
        self.ksizes = list(h["upsample_kernel_sizes"])
        self.res_k = list(h["resblock_kernel_sizes"])
        self.res_d = [list(d) for d in h["resblock_dilation_sizes"]]
        self.c0 = int(h["upsample_initial_channel"])
        self.gpt_dim = int(h["gpt_dim"])
        self.cond_each = bool(h.get("cond_d_vector_in_each_upsampling_layer", True))
        self.dtype = torch.float32
        self.device = torch.device("cpu")
        self._sd = None
        self._built = None

    # ---- nn.Module-like surface -------------------------------------------------------------------------------
    def load_state_dict(self, sd, strict=True):
        self._sd = {k: v.detach() for k, v in sd.items()}
        self._built = None
        return self

    def state_dict(self):
        return dict(self._sd or {})

    def to(self, *args, **kw):
        for a in list(args) + list(kw.values()):
            if isinstance(a, torch.dtype):
                self.dtype = a
            elif isinstance(a, (str, torch.device)):
                self.device = torch.device(a)
        self._built = None
        return self

    def half(self):
        return self.to(torch.float16)

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def float(self):
        return self.to(torch.float32)

    def modules(self):
        return iter(())

    def eval(self):
        return self

    def remove_weight_norm(self):
        """Fold g*v/||v||, cast to the compute dtype, reorder to [tap][Cin][Cout] and pack for the MFMA kernels."""
        self._build()
        return self

    # ---- weight preparation -----------------------------------------------------------------------------------
    def _w(self, prefix):
        sd = self._sd
        if prefix + ".weight" in sd:
            return sd[prefix + ".weight"].float()
        return fold_weight_norm(sd[prefix + ".weight_g"], sd[prefix + ".weight_v"])

    def _build(self):
        if self._built is not None:
            return
        if self._sd is None:
            raise RuntimeError("BigVGAN: load_state_dict() first")
        if self.device.type != "cuda":
            raise nat.NativeError("BigVGAN runs on the HIP kernels only: move it to a cuda device (no CPU fallback)")
        dev, T, sd = self.device, self.dtype, self._sd

        def f32(k):
            return sd[k].to(dev, torch.float32).contiguous()

        def conv_pack(prefix):
            w = self._w(prefix).to(dev)  # [Cout, Cin, k]
            return nat.pack_weight(w.permute(2, 1, 0).to(T).contiguous()), f32(prefix + ".bias")

        P = {}
        P["pre_w"], P["pre_b"] = conv_pack("conv_pre")
        P["post_w"], P["post_b"] = conv_pack("conv_post")
        P["ups"] = []
        for i, (u, k) in enumerate(zip(self.rates, self.ksizes)):
            w = self._w(f"ups.{i}.0").to(dev)  # [Cin, Cout, k]
            taps, off0, shift = convtr_as_conv(w, u)
            P["ups"].append(dict(w=nat.pack_weight(taps.to(T).contiguous()), b=f32(f"ups.{i}.0.bias").repeat(u).contiguous(),
                                 taps=taps.shape[0], off0=off0, shift=shift, u=u, cin=w.shape[0], cout=w.shape[1]))
        P["res"] = []
        for j in range(len(self.rates) * len(self.res_k)):
            k = self.res_k[j % len(self.res_k)]
            dil = self.res_d[j % len(self.res_k)]
            blk = dict(k=k, dil=dil, c1=[], c2=[], act=[])
            for n in range(len(dil)):
                blk["c1"].append(conv_pack(f"resblocks.{j}.convs1.{n}"))
                blk["c2"].append(conv_pack(f"resblocks.{j}.convs2.{n}"))
            for a in range(2 * len(dil)):
                blk["act"].append((f32(f"resblocks.{j}.activations.{a}.act.alpha"), f32(f"resblocks.{j}.activations.{a}.act.beta")))
            P["res"].append(blk)
        P["act_post"] = (f32("activation_post.act.alpha"), f32("activation_post.act.beta"))
        P["cond0"] = (f32("cond_layer.weight")[:, :, 0].contiguous(), f32("cond_layer.bias"))
        P["conds"] = [(f32(f"conds.{i}.weight")[:, :, 0].contiguous(), f32(f"conds.{i}.bias")) for i in range(len(self.rates))] \
            if self.cond_each else None
        fdef = kaiser_sinc_filter()
        upf = sd.get("activation_post.upsample.filter")
        dnf = sd.get("activation_post.downsample.lowpass.filter")
        P["up_f"] = fdef if upf is None else upf.float().reshape(-1).cpu().numpy()
        P["down_f"] = fdef if dnf is None else dnf.float().reshape(-1).cpu().numpy()
        P["spk"] = {k: v.to(dev, torch.float32) for k, v in sd.items() if k.startswith("speaker_encoder.") and
                    "num_batches" not in k}
        self._built = P

    # ---- forward ----------------------------------------------------------------------------------------------
    def speaker_engine(self):
        """The ECAPA-TDNN speaker encoder on the HIP kernels (speaker_engine.py), or None for an fp32 vocoder: a 16-bit vocoder (the
        benched precision) runs it in fp16; fp32 keeps the functional PyTorch form, the parity mode."""
        self._build()
        if self.dtype == torch.float32 or os.environ.get("ITTS_NATIVE_SPEAKER", "1") == "0":
            return None
        if self._built.get("spk_engine") is None:
            self._built["spk_engine"] = SpeakerEngine(self._built["spk"], dtype=torch.float16, device=self.device)
        return self._built["spk_engine"]

    def speaker_embedding(self, mel_ref: torch.Tensor) -> torch.Tensor:
        """mel_ref [B, Tref, 100] -> [B, 1, 512] fp32."""
        self._build()
        mel = mel_ref.to(self.device, torch.float32)
        eng = self.speaker_engine()
        if eng is not None and mel.shape[1] > 8:          # (the reflect padding of the dilated convolutions needs a few frames)
            return torch.stack([eng(mel[i].contiguous()).clone() for i in range(mel.shape[0])], 0)[:, None, :]
        return ecapa_embed(self._built["spk"], mel)

    def _act(self, x, ab, out=None, valid=None):
        P = self._built
        return nat.aa_snake(x, ab[0], ab[1], P["up_f"], P["down_f"], layout=0, out=out, valid_rows=valid)

    def forward(self, x: torch.Tensor, mel_ref: torch.Tensor = None, lens=None, speaker_embedding: torch.Tensor = None,
                taps: dict | None = None, profile: list | None = None):
        """x: GPT latent [B,T,gpt_dim]; mel_ref [B,Tref,100] (or a precomputed speaker_embedding [B,1,512]).
        profile (measurement aid, bench.py): a list that receives one [name, cuda event, FLOP, algorithmic bytes] entry per
        stage boundary -- FLOP = 2 * rows * Cout * Cin * taps of the convolutions launched since the previous entry, bytes =
        every launch's input + output (+ residual / accumulate operand) elements in the storage type.
        lens (ints or an int tensor [B], frames per batch element, <= T): RAGGED batch -- every layer treats element b as a
        sequence of lens[b] * (upsampling so far) rows: convolutions read zeros past it (their own zero padding), the
        anti-aliased activation's replicate padding clamps there, tiles past it are not computed.  Samples
        [0, lens[b] * hop) of row b then equal, bit for bit, what vocoding that element alone returns; the rest of the row
        is unspecified."""
        self._build()
        P, T, dev = self._built, self.dtype, self.device
        B, Tn, _ = x.shape
        vr = None       # valid rows at the current stage (int32 [B] on the device) or None
        if lens is not None:
            lens_t = torch.as_tensor(lens, dtype=torch.int32).reshape(-1)
            if lens_t.numel() != B or int(lens_t.max()) > Tn or int(lens_t.min()) < 0:
                raise ValueError("lens must hold one length in [0, T] per batch element")
            vr = lens_t.to(dev)
        es = torch.empty((), dtype=T).element_size()
        work = [0.0, 0.0]

        def conv_work(rows, cin, cout, k, extra_ops=0):
            work[0] += 2.0 * B * rows * cin * cout * k
            work[1] += B * rows * (cin + cout * (1 + extra_ops)) * es

        def mark(name):
            if profile is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                profile.append([name, ev, work[0], work[1]])
                work[0] = work[1] = 0.0

        mark("start")
        x = x.to(dev, T).contiguous()
        spk = speaker_embedding if speaker_embedding is not None else self.speaker_embedding(mel_ref)
        spk = spk.to(dev, torch.float32).reshape(spk.shape[0], -1)
        if spk.shape[0] == 1 and B > 1:
            spk = spk.expand(B, -1)
        c = self.c0
        cond = F.linear(spk, P["cond0"][0], P["cond0"][1]).contiguous()
        cur = torch.empty(B, Tn, c, dtype=T, device=dev)
        nat.gemm_conv(T, B, Tn, Tn, self.gpt_dim, c, P["pre_w"], x, cur, taps=7, off0=-3, dil=1, bias=P["pre_b"], bias2=cond,
                      valid_rows=vr)
        conv_work(Tn, self.gpt_dim, c, 7)
        mark("conditioning linear + conv_pre")
        if taps is not None:
            taps["conv_pre_cond"] = cur
        nk = len(self.res_k)
        for i, up in enumerate(P["ups"]):
            u, c = up["u"], up["cout"]
            Tu = Tn * u
            b2 = None
            if P["conds"] is not None:
                b2 = F.linear(spk, P["conds"][i][0], P["conds"][i][1]).repeat(1, u).contiguous()
            xu = torch.empty(B, Tu, c, dtype=T, device=dev)
            rows = Tn + 1 if up["taps"] == 2 else Tn
            nat.gemm_conv(T, B, Tn, rows, up["cin"], u * c, up["w"], cur, xu, taps=up["taps"], off0=up["off0"], dil=1,
                          bias=up["b"], bias2=b2, y_bstride=Tu * c, y_shift=up["shift"], y_limit=Tu * c, valid_rows=vr)
            conv_work(rows, up["cin"], u * c, up["taps"])
            if vr is not None:
                vr = vr * u
            if taps is not None:
                taps[f"up{i}_cond"] = xu
            xs = torch.empty_like(xu)
            ba, bb = torch.empty_like(xu), torch.empty_like(xu)
            pp = (torch.empty_like(xu), torch.empty_like(xu))
            for j in range(nk):
                blk = P["res"][i * nk + j]
                k = blk["k"]
                xc = xu
                nd = len(blk["dil"])
                for n, d in enumerate(blk["dil"]):
                    work[1] += 4 * B * Tu * c * es          # two activations: read + write each
                    conv_work(Tu, c, c, k)
                    conv_work(Tu, c, c, k, extra_ops=1 if (n + 1 < nd or j == 0) else 2)   # + residual (+ accumulate)
                    self._act(xc, blk["act"][2 * n], out=ba, valid=vr)
                    nat.gemm_conv(T, B, Tu, Tu, c, c, blk["c1"][n][0], ba, bb, taps=k, off0=-((k * d - d) // 2), dil=d,
                                  bias=blk["c1"][n][1], valid_rows=vr)
                    self._act(bb, blk["act"][2 * n + 1], out=ba, valid=vr)
                    if n + 1 < nd:
                        nat.gemm_conv(T, B, Tu, Tu, c, c, blk["c2"][n][0], ba, pp[n % 2], taps=k, off0=-((k - 1) // 2), dil=1,
                                      bias=blk["c2"][n][1], resid=xc, valid_rows=vr)
                        xc = pp[n % 2]
                    else:  # last conv of the block: + residual, then the 1/3 mean over the three AMP blocks
                        nat.gemm_conv(T, B, Tu, Tu, c, c, blk["c2"][n][0], ba, xs, taps=k, off0=-((k - 1) // 2), dil=1,
                                      bias=blk["c2"][n][1], resid=xc, accumulate=(j > 0), scale=1.0 / nk, valid_rows=vr)
            cur, Tn = xs, Tu
            mark(f"stage {i}: x{u} upsampler + {nk} AMP blocks at C = {c}, T = {Tu}")
            if taps is not None:
                taps[f"stage{i}"] = cur
        a = self._act(cur, P["act_post"], valid=vr)
        y = torch.empty(B, Tn, 1, dtype=T, device=dev)
        nat.gemm_conv(T, B, Tn, Tn, c, 1, P["post_w"], a, y, taps=7, off0=-3, dil=1, bias=P["post_b"], valid_rows=vr)
        wav = torch.empty(B, 1, Tn, dtype=torch.float32, device=dev)
        nat.tanh_pcm(y, wav=wav, pcm=None, apply_tanh=True)
        work[1] += 2 * B * Tn * c * es + B * Tn * 4
        conv_work(Tn, c, 1, 7)
        mark("activation + conv_post + tanh")
        return wav, None

    __call__ = forward
