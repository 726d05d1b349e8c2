"""ECAPA-TDNN speaker encoder on the HIP kernels: mel of one prompt [T, 100] -> speaker embedding [512].

~43 launches of libindextts_hip.so instead of the ~330 library launches of the functional PyTorch form (ECAPA_TDNN.py in this
package: the fp32 parity mode and the checker of this path).  16-bit storage (fp16 by default), fp32 accumulation; every
activation is a packed operand [frames][channels]:

  block 0   itts_im2col_reflect (k = 5, reflect padding) -> GEMM with the ReLU + BatchNorm epilogue
  3 x SE-Res2Net   tdnn1 GEMM -> 7 x itts_res2_step (k = 3 dilated, chunk by chunk) -> tdnn2 GEMM -> itts_se_gate ->
            itts_scale_resid, written straight into its 512 columns of the [frames][1536] MFA operand
  MFA GEMM -> itts_col_stats (global context) -> GEMV (the context half of asp.tdnn, folded into its bias) -> asp.tdnn GEMM
  (ReLU + BatchNorm + tanh epilogue) -> asp.conv GEMM -> itts_col_stats (softmax over time, weighted mean / std, asp_bn) -> fc GEMV

Eval-mode BatchNorm is an affine map per channel; it sits BEHIND the ReLU of a TDNN block, so it is an epilogue of the block's
GEMM, not a fold into its weights.  Follows indextts/BigVGAN/ECAPA_TDNN.py:79-130 (TDNNBlock), :470-581 (the network),
nnet/CNN.py:430-488 (reflect "same" padding), nnet/normalization.py:13-108; the weight keys are the reference checkpoint's.
"""
from __future__ import annotations

import threading

import torch

from .. import _native as nat

EPS_BN = 1e-5


class SpeakerEngine:
    ROWS_PER_WG = 32

    def __init__(self, W: dict, dtype=torch.float16, device="cuda", prefix="speaker_encoder."):
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("the speaker-encoder kernels are built for fp16 / bf16 (fp32 is the functional PyTorch form)")
        nat.lib()
        self.dtype, self.device = dtype, torch.device(device)
        dev = self.device

        def f32(k):
            return W[prefix + k].detach().to(dev, torch.float32).contiguous()

        def bn(p):
            s = f32(p + ".weight") / torch.sqrt(f32(p + ".running_var") + EPS_BN)
            return s.contiguous(), (f32(p + ".bias") - f32(p + ".running_mean") * s).contiguous()

        def packed(w_kn):
            return nat.pack_weight(w_kn.to(dev, torch.float32).to(dtype).contiguous())

        def tdnn(p):
            """1 x 1 TDNN block -> (packed [Cin][Cout], bias, (bn scale, bn shift))"""
            return packed(f32(p + ".conv.conv.weight")[:, :, 0].t()), f32(p + ".conv.conv.bias"), bn(p + ".norm.norm")

        w0 = f32("blocks.0.conv.conv.weight")                       # [512, 100, 5]
        self.C, self.F, self.k0 = w0.shape
        self.K0 = (self.F * self.k0 + 31) // 32 * 32
        w0m = torch.zeros(self.K0, self.C, device=dev)
        w0m[: self.F * self.k0] = w0.permute(2, 1, 0).reshape(self.F * self.k0, self.C)     # row = tap * F + input channel
        self.b0 = (packed(w0m), f32("blocks.0.conv.conv.bias"), bn("blocks.0.norm.norm"))
        self.blocks = []
        for bi, dil in ((1, 2), (2, 3), (3, 4)):
            p = f"blocks.{bi}."
            b = dict(dil=dil, tdnn1=tdnn(p + "tdnn1"), tdnn2=tdnn(p + "tdnn2"), res=[])
            for s in range(7):
                q = f"{p}res2net_block.blocks.{s}"
                w = f32(q + ".conv.conv.weight")                   # [64, 64, 3]
                if tuple(w.shape) != (64, 64, 3):
                    raise ValueError("the Res2Net step kernel is built for 64-channel chunks, k = 3")
                b["res"].append((packed(w.permute(2, 1, 0).reshape(192, 64)), f32(q + ".conv.conv.bias"), bn(q + ".norm.norm")))
            b["se"] = (f32(p + "se_block.conv1.conv.weight")[:, :, 0].to(dtype).contiguous(), f32(p + "se_block.conv1.conv.bias"),
                       f32(p + "se_block.conv2.conv.weight")[:, :, 0].to(dtype).contiguous(), f32(p + "se_block.conv2.conv.bias"))
            self.blocks.append(b)
        self.CM = 3 * self.C
        self.mfa = tdnn("mfa")
        wa = f32("asp.tdnn.conv.conv.weight")[:, :, 0]              # [128, 3 * CM]: x | mean | std
        self.A = wa.shape[0]
        self.asp_x = packed(wa[:, : self.CM].t())
        self.asp_ctx = packed(wa[:, self.CM:].t())
        self.asp_b = f32("asp.tdnn.conv.conv.bias")
        self.asp_bn = bn("asp.tdnn.norm.norm")
        self.asp_conv = (packed(f32("asp.conv.conv.weight")[:, :, 0].t()), f32("asp.conv.conv.bias"))
        self.pool_bn = bn("asp_bn.norm")
        self.fc = (packed(f32("fc.conv.weight")[:, :, 0].t()), f32("fc.conv.bias"))
        self.E = self.fc[1].shape[0]
        self._bufs = {}
        self.launches = 0

    def forget(self):
        """Drop the calling thread's per-length buffers (the caller holds no captured graph over them any more)."""
        me = threading.get_ident()
        for k in [k for k in self._bufs if k[1] == me]:
            del self._bufs[k]

    def _buffers(self, T):
        # per prompt length AND per host thread: the vocoder object (and with it this engine) is shared by the replicas of
        # infer.RequestPool, one thread + stream each -- shared packed weights, private activations
        key = (T, threading.get_ident())
        b = self._bufs.get(key)
        if b is None:
            dev, dt, C = self.device, self.dtype, self.C
            mtp = (T + 15) // 16
            z = lambda *s, dtype=dt: torch.zeros(*s, dtype=dtype, device=dev)      # noqa: E731
            b = dict(mtp=mtp, col=z(mtp * 16 * self.K0), x0=z(mtp * 16 * C), y1=z(mtp * 16 * C), cat=z(mtp * 16 * C),
                     y2=z(mtp * 16 * C), gate=z(C, dtype=torch.float32), feats=z(mtp * 16 * self.CM), xm=z(mtp * 16 * self.CM),
                     ctx=z(2 * self.CM), asp_bias=z(self.A, dtype=torch.float32), att=z(mtp * 16 * self.A), logit=z(T, self.CM),
                     pooled=z(2 * self.CM), out=z(self.E, dtype=torch.float32))
            self._bufs[key] = b
        return b

    def __call__(self, mel_tf: torch.Tensor) -> torch.Tensor:
        """mel_tf fp32 [T, F] (time-major, on the device) -> embedding fp32 [E].  Graph-capturable after the first call of a length."""
        if mel_tf.dim() != 2 or mel_tf.dtype != torch.float32 or not mel_tf.is_cuda or not mel_tf.is_contiguous():
            raise nat.NativeError("SpeakerEngine takes a contiguous fp32 device tensor [T, F]")
        T, Fq = mel_tf.shape
        if Fq != self.F:
            raise ValueError(f"{Fq} mel bins, the encoder takes {self.F}")
        b = self._buffers(T)
        dt, C, CM, mtp, R = self.dtype, self.C, self.CM, b["mtp"], self.ROWS_PER_WG
        RA, RAT = nat.EPI_RELU_AFFINE_STORE, nat.EPI_RELU_AFFINE_TANH_STORE

        def gemm(x, w, N, K, y, epi=RA, x_off=0, **kw):
            wp, bias, post = w
            nat.gemm_skinny(dt, T, N, K, wp, bias, x=x[x_off:], epi=epi, y=y, x_packed=True, y_packed=True, rows_per_wg=R,
                            post=post, **kw)

        n = 0
        nat.im2col_reflect(mel_tf, b["col"], self.k0, 1, self.K0, mtp)
        gemm(b["col"], self.b0, C, self.K0, b["x0"])
        n += 2
        res, res_off = b["x0"], 0
        for i, blk in enumerate(self.blocks):
            gemm(res, blk["tdnn1"], C, C, b["y1"], x_off=res_off)
            for s in range(7):
                wp, bias, (sc, sh) = blk["res"][s]
                nat.res2_step(b["y1"], b["cat"], wp, bias, sc, sh, T, mtp, s + 1, blk["dil"], s == 0)
            gemm(b["cat"], blk["tdnn2"], C, C, b["y2"])
            w1, b1, w2, b2 = blk["se"]
            nat.se_gate(b["y2"], w1, b1, w2, b2, b["gate"], T, C, w1.shape[0], mtp)
            out_off = i * (C // 32) * mtp * 512                    # this block's run of k-steps inside the MFA operand
            nat.scale_resid(b["y2"], res[res_off:], b["gate"], b["feats"][out_off:], T, C, mtp)
            res, res_off = b["feats"], out_off
            n += 11
        gemm(b["feats"], self.mfa, CM, CM, b["xm"])
        nat.col_stats(b["xm"], b["ctx"], T, CM, mtp)
        # asp.tdnn over cat(x, mean, std): the two constant thirds are a bias (a GEMV over [mean | std])
        nat.gemm_skinny(dt, 1, self.A, 2 * CM, self.asp_ctx, self.asp_b, x=b["ctx"], epi=nat.EPI_STORE_F32, yf=b["asp_bias"])
        gemm(b["xm"], (self.asp_x, b["asp_bias"], self.asp_bn), self.A, CM, b["att"], epi=RAT)
        nat.gemm_skinny(dt, T, CM, self.A, self.asp_conv[0], self.asp_conv[1], x=b["att"], epi=nat.EPI_STORE, y=b["logit"],
                        x_packed=True, rows_per_wg=R)
        nat.col_stats(b["xm"], b["pooled"], T, CM, mtp, logit=b["logit"], scale=self.pool_bn[0], shift=self.pool_bn[1])
        nat.gemm_skinny(dt, 1, self.E, 2 * CM, self.fc[0], self.fc[1], x=b["pooled"], epi=nat.EPI_STORE_F32, yf=b["out"])
        self.launches = n + 7
        return b["out"]
