"""ORACLE (test infrastructure): CPU fp32 restatement of the conditioned BigVGAN generator forward.

Follows  indextts/BigVGAN/models.py:203-252 (BigVGAN.forward), :65-74 (AMPBlock1.forward), :254-262
(remove_weight_norm == fold w = g*v/||v||, norm over every dim but 0),
indextts/BigVGAN/alias_free_torch/{act.py:10-28, resample.py:10-48, filter.py:29-95} (Activation1d) and
indextts/BigVGAN/activations.py:63-122 (SnakeBeta, log-scale alpha/beta).  The speaker embedding (ECAPA-TDNN,
models.py:204) is an INPUT here: it is host-side product code pinned directly by tests/golden/bigvgan.npz.

Pinned against tests/golden/act1d.npz and bigvgan.npz (reference-run fixtures) in tests/test_oracle_vs_golden.py.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

UPSAMPLE_RATES = (4, 4, 4, 4, 2, 2)
UPSAMPLE_KERNELS = (8, 8, 4, 4, 4, 4)
RES_KERNELS = (3, 7, 11)
RES_DILATIONS = (1, 3, 5)


def kaiser_sinc_filter(cutoff=0.25, half_width=0.3, kernel_size=12) -> np.ndarray:
    """filter.py:29-58 (even kernel): Kaiser-windowed sinc, unit DC gain."""
    half = kernel_size // 2
    A = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50.0 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0) if A >= 21.0 else 0.0)
    t = np.arange(-half, half) + 0.5
    w = np.kaiser(kernel_size, beta)
    f = 2 * cutoff * w * np.sinc(2 * cutoff * t)
    return (f / f.sum()).astype(np.float32)


def fold_weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    n = v.flatten(1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def activation1d(x, alpha_log, beta_log, up_f, down_f):
    """x [B,C,T] -> [B,C,T]: replicate-pad 5, x2 zero-stuffed FIR (gain 2), SnakeBeta, replicate-pad 5/6, stride-2 FIR."""
    B, C, T = x.shape
    fu = torch.as_tensor(up_f, dtype=x.dtype).view(1, 1, -1).expand(C, 1, -1)
    fd = torch.as_tensor(down_f, dtype=x.dtype).view(1, 1, -1).expand(C, 1, -1)
    xp = F.pad(x, (5, 5), mode="replicate")
    u = 2.0 * F.conv_transpose1d(xp, fu, stride=2, groups=C)[..., 15:-15]          # resample.py:29-34
    a = torch.exp(alpha_log).view(1, C, 1)
    b = torch.exp(beta_log).view(1, C, 1)
    s = u + (1.0 / (b + 1e-9)) * torch.sin(u * a).pow(2)                              # activations.py:117-121
    sp = F.pad(s, (5, 6), mode="replicate")                                           # filter.py:77-95
    return F.conv1d(sp, fd, stride=2, groups=C)


class Weights:
    """Folded fp32 weights from a reference-format 'generator' state dict (weight_g/weight_v parametrisation)."""

    def __init__(self, sd: dict):
        t = {k: torch.as_tensor(np.asarray(v)).float() for k, v in sd.items() if "num_batches" not in k}

        def w(prefix):
            if prefix + ".weight" in t:
                return t[prefix + ".weight"]
            return fold_weight_norm(t[prefix + ".weight_g"], t[prefix + ".weight_v"])

        self.t = t
        self.w = w
        f = kaiser_sinc_filter()
        self.up_f = t.get("activation_post.upsample.filter", torch.from_numpy(f)).reshape(-1)
        self.down_f = t.get("activation_post.downsample.lowpass.filter", torch.from_numpy(f)).reshape(-1)


def amp_block(x, W: Weights, j: int, k: int):
    p = f"resblocks.{j}."
    for n, d in enumerate(RES_DILATIONS):
        xt = activation1d(x, W.t[p + f"activations.{2*n}.act.alpha"], W.t[p + f"activations.{2*n}.act.beta"],
                          W.up_f, W.down_f)
        xt = F.conv1d(xt, W.w(p + f"convs1.{n}"), W.t[p + f"convs1.{n}.bias"], dilation=d, padding=(k * d - d) // 2)
        xt = activation1d(xt, W.t[p + f"activations.{2*n+1}.act.alpha"], W.t[p + f"activations.{2*n+1}.act.beta"],
                          W.up_f, W.down_f)
        xt = F.conv1d(xt, W.w(p + f"convs2.{n}"), W.t[p + f"convs2.{n}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def forward(latent, spk, W: Weights, taps: dict | None = None):
    """latent [B,T,1280], spk [B,512,1] (ECAPA embedding, transposed) -> wav [B,1,1024*T]."""
    x = latent.transpose(1, 2)
    x = F.conv1d(x, W.w("conv_pre"), W.t["conv_pre.bias"], padding=3)
    if taps is not None:
        taps["conv_pre"] = x
    x = x + F.conv1d(spk, W.t["cond_layer.weight"], W.t["cond_layer.bias"])
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        x = F.conv_transpose1d(x, W.w(f"ups.{i}.0"), W.t[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        if taps is not None:
            taps[f"up{i}"] = x
        x = x + F.conv1d(spk, W.t[f"conds.{i}.weight"], W.t[f"conds.{i}.bias"])
        xs = None
        for j, rk in enumerate(RES_KERNELS):
            y = amp_block(x, W, 3 * i + j, rk)
            xs = y if xs is None else xs + y
        x = xs / len(RES_KERNELS)
        if taps is not None:
            taps[f"stage{i}"] = x
    x = activation1d(x, W.t["activation_post.act.alpha"], W.t["activation_post.act.beta"], W.up_f, W.down_f)
    x = F.conv1d(x, W.w("conv_post"), W.t["conv_post.bias"], padding=3)
    return torch.tanh(x)


def to_pcm16(wav: torch.Tensor) -> np.ndarray:
    """infer.py:892,911: clamp(32767*wav, +-32767) then astype(int16) (truncation toward zero)."""
    return torch.clamp(32767 * wav, -32767.0, 32767.0).to(torch.float32).numpy().astype(np.int16)
