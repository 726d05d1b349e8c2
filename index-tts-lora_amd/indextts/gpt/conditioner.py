"""Conformer + Perceiver conditioner on the HIP kernels: the prompt front-end of UnifiedVoice.get_conditioning.

One prompt (mel [T, 100], no padding) -> conditioning latents [32, 1280], as 72 launches of libindextts_hip.so instead of the
~250 library launches of the functional PyTorch form (conformer_encoder.py / perceiver.py in this package, which stay as the fp32
parity mode and as the checker of this path).  Structure, all 16-bit storage (fp16 by default) with an fp32 residual stream:

  front   itts_subsample_conv (Conv2d 3x3 stride 2 + ReLU) -> embed.out as a split-K plain GEMM -> itts_rows (slabs + bias, packed copy)
  block   QKV' (norm_mha folded) -> itts_mha_small (relative positions, no rel_shift) -> out-projection (+ residual, packed copy)
          -> pointwise_conv1' (norm_conv folded) -> itts_glu_dwconv_ln_silu -> pointwise_conv2 (+ residual, packed copy)
          -> w_1' (norm_ff folded, SiLU epilogue) -> w_2 (+ residual) -> itts_rows (norm_final in place, packed copy)      9 launches
  perceiver  proj_context' (after_norm folded) into the [latents ; context] operand; per layer: [to_q | to_kv] over that operand ->
          itts_mha_small -> to_out (+ residual) -> ff.0 -> itts_geglu -> ff.2 (+ residual);  itts_rows (RMSNorm)

"X'" = the LayerNorm in front of X folded into the GEMM (itts_skinny_args.ln_c): the GEMM multiplies the raw residual rows and
normalises in its epilogue, as in the decode step.  Follows indextts/gpt/model.py:487-546 (conformer_perceiver branch),
conformer_encoder.py:167-290,360-386, conformer/subsampling.py:111-143, conformer/embedding.py, conformer/attention.py and
perceiver.py:181-312 of the reference; the weight keys are the reference checkpoint's.
"""
from __future__ import annotations

import math
import threading

import torch

from .. import _native as nat
from .conformer_encoder import sinusoid_table


def _fold(W, b, gamma, beta, dtype):
    """LN(h; gamma, beta) W + b = rstd (h W' - mean c) + d  ->  (pack(T(gamma . W)), c, d); W [K, N] fp32 (see GPTEngine)."""
    Wm = W.to(torch.float64)
    Wr = (gamma.to(torch.float64)[:, None] * Wm).to(torch.float32).to(dtype)
    c = Wr.to(torch.float64).sum(0).to(torch.float32).contiguous()
    d = beta.to(torch.float64) @ Wm
    if b is not None:
        d = d + b.to(torch.float64)
    return nat.pack_weight(Wr.contiguous()), c, d.to(torch.float32).contiguous()


class ConditionerEngine:
    ROWS_PER_WG = 32     # row tiles dealt to grid.z: a ~150-row GEMM is ~240 workgroups of 32 rows x 2-3 column tiles

    def __init__(self, W: dict, heads: int = 8, dtype=torch.float16, device="cuda", enc="conditioning_encoder.",
                 per="perceiver_encoder."):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("the conditioner kernels are built for fp16 / bf16 (fp32 is the functional PyTorch form)")
        nat.lib()
        self.dtype, self.device, self.H = dtype, torch.device(device), heads
        dev = self.device

        def f32(k):
            return W[k].detach().to(dev, torch.float32).contiguous()

        def packed(w_kn):
            return nat.pack_weight(w_kn.to(dev, torch.float32).to(dtype).contiguous())

        # ---- Conv2dSubsampling2
        w0 = f32(enc + "embed.conv.0.weight")
        self.C = w0.shape[0]
        self.w_conv = w0.view(self.C, 9).contiguous()
        self.b_conv = f32(enc + "embed.conv.0.bias")
        we = f32(enc + "embed.out.0.weight")                      # [d, C * F2]
        self.d = d = we.shape[0]
        if d != heads * 64:
            raise ValueError("the attention kernel is built for 64-wide heads")
        self.KE = we.shape[1]
        xs = math.sqrt(d)                                          # RelPositionalEncoding: x * sqrt(d), folded into the projection
        self.w_embed = packed(we.t() * xs)
        self.b_embed = (f32(enc + "embed.out.0.bias") * xs).contiguous()
        # ---- encoder blocks
        self.blocks = []
        i = 0
        while f"{enc}encoders.{i}.norm_mha.weight" in W:
            p = f"{enc}encoders.{i}."
            wq, wk, wv = (f32(p + f"self_attn.linear_{n}.weight") for n in "qkv")
            bq, bk, bv = (f32(p + f"self_attn.linear_{n}.bias") for n in "qkv")
            b = {}
            b["w_qkv"], b["c_qkv"], b["d_qkv"] = _fold(torch.cat([wq, wk, wv], 0).t(), torch.cat([bq, bk, bv]),
                                                       f32(p + "norm_mha.weight"), f32(p + "norm_mha.bias"), dtype)
            b["w_pos"] = f32(p + "self_attn.linear_pos.weight")
            b["u"] = f32(p + "self_attn.pos_bias_u").reshape(-1).contiguous()
            b["v"] = f32(p + "self_attn.pos_bias_v").reshape(-1).contiguous()
            b["w_o"], b["b_o"] = packed(f32(p + "self_attn.linear_out.weight").t()), f32(p + "self_attn.linear_out.bias")
            b["w_pw1"], b["c_pw1"], b["d_pw1"] = _fold(f32(p + "conv_module.pointwise_conv1.weight")[:, :, 0].t(),
                                                       f32(p + "conv_module.pointwise_conv1.bias"),
                                                       f32(p + "norm_conv.weight"), f32(p + "norm_conv.bias"), dtype)
            wd = f32(p + "conv_module.depthwise_conv.weight")
            b["w_dw"], b["b_dw"] = wd.view(wd.shape[0], wd.shape[-1]).contiguous(), f32(p + "conv_module.depthwise_conv.bias")
            b["ln_dw"] = (f32(p + "conv_module.norm.weight"), f32(p + "conv_module.norm.bias"))
            b["w_pw2"], b["b_pw2"] = packed(f32(p + "conv_module.pointwise_conv2.weight")[:, :, 0].t()), f32(p + "conv_module.pointwise_conv2.bias")
            b["w_ff1"], b["c_ff1"], b["d_ff1"] = _fold(f32(p + "feed_forward.w_1.weight").t(), f32(p + "feed_forward.w_1.bias"),
                                                       f32(p + "norm_ff.weight"), f32(p + "norm_ff.bias"), dtype)
            b["w_ff2"], b["b_ff2"] = packed(f32(p + "feed_forward.w_2.weight").t()), f32(p + "feed_forward.w_2.bias")
            b["ln_final"] = (f32(p + "norm_final.weight"), f32(p + "norm_final.bias"))
            self.blocks.append(b)
            i += 1
        self.FF = W[f"{enc}encoders.0.feed_forward.w_1.weight"].shape[0]
        # ---- perceiver
        self.lat0 = f32(per + "latents")
        self.NL, self.DL = self.lat0.shape
        if self.NL % 16:
            raise ValueError("the latent count must be a multiple of 16 (the context rows follow them in one packed operand)")
        self.w_ctx, self.c_ctx, self.d_ctx = _fold(f32(per + "proj_context.weight").t(), f32(per + "proj_context.bias"),
                                                   f32(enc + "after_norm.weight"), f32(enc + "after_norm.bias"), dtype)
        self.players = []
        i = 0
        while f"{per}layers.{i}.0.to_q.weight" in W:
            p = f"{per}layers.{i}."
            L = {}
            wq, wkv = f32(p + "0.to_q.weight"), f32(p + "0.to_kv.weight")
            if wq.shape[0] != heads * 64 or wkv.shape[0] != 2 * heads * 64:
                raise ValueError("the attention kernel is built for 64-wide heads")
            L["w_qkv"] = packed(torch.cat([wq, wkv], 0).t())
            L["w_o"] = packed(f32(p + "0.to_out.weight").t())
            w1, b1 = f32(p + "1.0.weight"), f32(p + "1.0.bias")
            inner = w1.shape[0] // 2
            kp = (inner + 31) // 32 * 32
            w1p = torch.zeros(self.DL, 2 * kp, device=dev)
            w1p[:, :inner], w1p[:, kp:kp + inner] = w1[:inner].t(), w1[inner:].t()
            b1p = torch.zeros(2 * kp, device=dev)
            b1p[:inner], b1p[kp:kp + inner] = b1[:inner], b1[inner:]
            w2p = torch.zeros(kp, self.DL, device=dev)
            w2p[:inner] = f32(p + "1.2.weight").t()
            L["w_ff1"], L["b_ff1"], L["w_ff2"], L["b_ff2"], L["kp"] = packed(w1p), b1p.contiguous(), packed(w2p), f32(p + "1.2.bias"), kp
            self.players.append(L)
            i += 1
        self.gamma = f32(per + "norm.gamma")
        self._bufs = {}
        self.launches = 0

    # ------------------------------------------------------------------------------------------------------------------
    def forget(self):
        """Drop the calling thread's per-length buffers (the caller holds no captured graph over them any more)."""
        me = threading.get_ident()
        for k in [k for k in self._bufs if k[2] == me]:
            del self._bufs[k]

    def _buffers(self, T, Fq):
        # per prompt length AND per host thread: replicas of one model (infer.RequestPool: one thread + stream each) share this
        # engine's packed weights but must not share activations
        key = (T, Fq, threading.get_ident())
        b = self._bufs.get(key)
        if b is not None:
            return b
        dev, dt, d, H = self.device, self.dtype, self.d, self.H
        t, f2 = (T - 3) // 2 + 1, (Fq - 3) // 2 + 1
        if self.C * f2 != self.KE:
            raise ValueError(f"{Fq} mel bins give {self.C * f2} features, embed.out takes {self.KE}")
        mtp = (t + 15) // 16
        rows_all = self.NL + t
        mtp_all = (rows_all + 15) // 16
        b = dict(t=t, mtp=mtp, mtp_all=mtp_all, rows_all=rows_all)
        z = lambda *s, dtype=dt: torch.zeros(*s, dtype=dtype, device=dev)      # noqa: E731  (padding rows of packed operands read as 0)
        b["x0"] = z(t, self.KE)
        # split-K of embed.out (K = 25 088): 2 x 4 output tiles x KS slices ~ one round of the chip
        tiles = ((t + 127) // 128) * (d // 128)
        b["ks_e"] = max(1, min(64, 256 // tiles, self.KE // 32))
        b["slab"] = z(b["ks_e"], t, d, dtype=torch.float32)
        b["x"] = z(t, d, dtype=torch.float32)
        b["hb"] = z(mtp * 16 * d)
        b["qkv"] = z(t, 3 * d)
        b["att"] = z(mtp * 16 * d)
        b["pw1"] = z(t, 2 * d)
        b["dw"] = z(mtp * 16 * d)
        b["ff"] = z(mtp * 16 * self.FF)
        pos = sinusoid_table(t, d, dev, torch.float32)
        b["pos"] = [torch.nn.functional.linear(pos, blk["w_pos"]).view(t, H, 64).transpose(0, 1).to(dt).contiguous() for blk in self.blocks]
        b["xall"] = z(mtp_all * 16 * self.DL)
        b["lat"] = z(self.NL, self.DL, dtype=torch.float32)
        b["qkv2"] = z(rows_all, 3 * d)
        b["att2"] = z(self.NL * d)
        kp = max(L["kp"] for L in self.players)
        b["h1"] = z(self.NL, 2 * kp)
        b["g"] = z(self.NL * kp)
        b["out"] = z(self.NL, self.DL, dtype=torch.float32)
        self._bufs[key] = b
        return b

    def __call__(self, mel_tf: torch.Tensor) -> torch.Tensor:
        """mel_tf fp32 [T, F] (time-major, on the device) -> conds fp32 [NL, DL].  Graph-capturable once the buffers of this
        prompt length exist (first call)."""
        if mel_tf.dim() != 2 or mel_tf.dtype != torch.float32 or not mel_tf.is_cuda or not mel_tf.is_contiguous():
            raise nat.NativeError("ConditionerEngine takes a contiguous fp32 device tensor [T, F]")
        T, Fq = mel_tf.shape
        b = self._buffers(T, Fq)
        dt, d, H, t, mtp, R = self.dtype, self.d, self.H, b["t"], b["mtp"], self.ROWS_PER_WG
        n = 0
        x, hb = b["x"], b["hb"]
        # ---- front
        nat.subsample_conv(mel_tf, self.w_conv, self.b_conv, b["x0"])
        nat.gemm_conv(dt, 1, t, t, self.KE, d, self.w_embed, b["x0"], b["slab"], y_f32=True, ksplit=b["ks_e"])
        nat.rows(t, d, dt, slab=b["slab"], nslab=b["ks_e"], bias=self.b_embed, y=x, y_packed=hb)
        n += 3
        scale = 1.0 / math.sqrt(64)
        for blk, pos in zip(self.blocks, b["pos"]):
            qkv = b["qkv"]
            nat.gemm_skinny(dt, t, 3 * d, d, blk["w_qkv"], blk["d_qkv"], x=hb, epi=nat.EPI_STORE, y=qkv, x_packed=True,
                            ln_c=blk["c_qkv"], rows_per_wg=R)
            nat.mha_small(qkv, qkv[:, d:], qkv[:, 2 * d:], b["att"], t, t, H, 3 * d, 3 * d, 3 * d, mtp, scale, pos=pos,
                          bias_u=blk["u"], bias_v=blk["v"])
            nat.gemm_skinny(dt, t, d, d, blk["w_o"], blk["b_o"], x=b["att"], epi=nat.EPI_RESID_F32, yf=x, y=hb, x_packed=True,
                            y_packed=True, rows_per_wg=R)
            nat.gemm_skinny(dt, t, 2 * d, d, blk["w_pw1"], blk["d_pw1"], x=hb, epi=nat.EPI_STORE, y=b["pw1"], x_packed=True,
                            ln_c=blk["c_pw1"], rows_per_wg=R)
            nat.glu_dwconv_ln_silu(b["pw1"], blk["w_dw"], blk["b_dw"], blk["ln_dw"][0], blk["ln_dw"][1], b["dw"], t, d, mtp)
            nat.gemm_skinny(dt, t, d, d, blk["w_pw2"], blk["b_pw2"], x=b["dw"], epi=nat.EPI_RESID_F32, yf=x, y=hb, x_packed=True,
                            y_packed=True, rows_per_wg=R)
            nat.gemm_skinny(dt, t, self.FF, d, blk["w_ff1"], blk["d_ff1"], x=hb, epi=nat.EPI_SILU_STORE, y=b["ff"], x_packed=True,
                            y_packed=True, ln_c=blk["c_ff1"], rows_per_wg=R)
            nat.gemm_skinny(dt, t, d, self.FF, blk["w_ff2"], blk["b_ff2"], x=b["ff"], epi=nat.EPI_RESID_F32, yf=x, x_packed=True,
                            rows_per_wg=R)
            nat.rows(t, d, dt, x=x, norm=1, w=blk["ln_final"][0], b=blk["ln_final"][1], y=x, y_packed=hb)
            n += 9
        # ---- perceiver: operand rows [0, NL) = latents, [NL, NL + t) = proj_context(after_norm(x))
        NL, DL, ma, ra = self.NL, self.DL, b["mtp_all"], b["rows_all"]
        lat, xall = b["lat"], b["xall"]
        nat.gemm_skinny(dt, t, DL, d, self.w_ctx, self.d_ctx, x=hb, epi=nat.EPI_STORE, y=xall, x_packed=True, y_packed=True,
                        ln_c=self.c_ctx, rows_per_wg=R, y_row0=NL, y_mtp=ma)
        nat.rows(NL, DL, dt, x=self.lat0, y=lat, y_packed=xall, y_row0=0, y_mtp=ma)
        n += 2
        for L in self.players:
            qkv2, kp = b["qkv2"], L["kp"]
            nat.gemm_skinny(dt, ra, 3 * d, DL, L["w_qkv"], None, x=xall, epi=nat.EPI_STORE, y=qkv2, x_packed=True, rows_per_wg=R)
            nat.mha_small(qkv2, qkv2[:, d:], qkv2[:, 2 * d:], b["att2"], NL, ra, H, 3 * d, 3 * d, 3 * d, NL // 16, scale)
            nat.gemm_skinny(dt, NL, DL, d, L["w_o"], None, x=b["att2"], epi=nat.EPI_RESID_F32, yf=lat, y=xall, x_packed=True,
                            y_packed=True, y_row0=0, y_mtp=ma)
            # the feed-forward reads the latent rows of the [latents ; context] operand: its first NL // 16 row tiles
            nat.gemm_skinny(dt, NL, 2 * kp, DL, L["w_ff1"], L["b_ff1"], x=xall, epi=nat.EPI_STORE, y=b["h1"], x_packed=True, x_mtp=ma)
            nat.geglu(b["h1"], b["g"], NL, kp)
            nat.gemm_skinny(dt, NL, DL, kp, L["w_ff2"], L["b_ff2"], x=b["g"], epi=nat.EPI_RESID_F32, yf=lat, y=xall, x_packed=True,
                            y_packed=True, y_row0=0, y_mtp=ma)
            n += 6
        nat.rows(NL, DL, dt, x=lat, norm=2, w=self.gamma, y=b["out"])
        self.launches = n + 1
        return b["out"]
