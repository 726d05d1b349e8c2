"""Conformer conditioning encoder (host-side PyTorch-ROCm; runs once per prompt).

Functional restatement over a flat weight dict that uses the reference's checkpoint key names
(`conditioning_encoder.*`).  Follows indextts/gpt/conformer_encoder.py:167-290,360-386 (layer order,
pre-norm, no macaron FFN, final per-layer LayerNorm), conformer/subsampling.py:111-143 (Conv2dSubsampling2),
conformer/embedding.py (sinusoidal table, x*sqrt(d), pos_emb = table[:T]) and conformer/attention.py
(rel-pos attention WITHOUT rel_shift: scores = ((q+u)k^T + (q+v)p^T)/sqrt(dk)).
"""
import math

import torch
import torch.nn.functional as F


def sinusoid_table(n: int, d: int, device, dtype=torch.float32):
    pos = torch.arange(n, device=device, dtype=torch.float32)[:, None]
    div = torch.exp(torch.arange(0, d, 2, device=device, dtype=torch.float32) * (-math.log(10000.0) / d))
    pe = torch.zeros(n, d, device=device, dtype=torch.float32)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def _ln(x, W, p):
    return F.layer_norm(x, (x.shape[-1],), W[p + ".weight"], W[p + ".bias"], 1e-5)


def _lin(x, W, p):
    return F.linear(x, W[p + ".weight"], W.get(p + ".bias"))


def conformer_encode(W: dict, mel_btf: torch.Tensor, lengths, heads: int = 8, prefix="conditioning_encoder."):
    """mel_btf [B,T,100], lengths [B] -> (x [B,T',512], mask [B,1,T'] bool).
    lengths = None: every row is T frames long (one prompt, or prompts of one length) -- there is no padding, the mask
    operations of the general form are identities and are left out (same values, ~50 launches fewer); mask is returned as None."""
    B, T, _ = mel_btf.shape
    dev = mel_btf.device
    full = lengths is None
    mask = None if full else (torch.arange(T, device=dev)[None, :] < lengths.to(dev)[:, None])[:, None, :]
    # Conv2dSubsampling2
    # Conv2d(1, C, 3, stride 2) as unfold + matmul (a [T'*F', 9] x [9, C] product)
    w0 = W[prefix + "embed.conv.0.weight"]
    patches = F.unfold(mel_btf[:, None], kernel_size=3, stride=2)                  # [B, 9, T'*F']
    t, f = (T - 3) // 2 + 1, (mel_btf.shape[2] - 3) // 2 + 1
    x = F.relu(torch.matmul(w0.view(w0.shape[0], 9), patches) + W[prefix + "embed.conv.0.bias"][None, :, None])
    b, c = x.shape[0], x.shape[1]
    x = x.view(b, c, t, f)
    x = _lin(x.transpose(1, 2).reshape(b, t, c * f), W, prefix + "embed.out.0")
    if not full:
        mask = mask[:, :, 2::2]
    d = x.shape[-1]
    x = x * math.sqrt(d)
    dk = d // heads
    n = 0
    while f"{prefix}encoders.{n}.norm_mha.weight" in W:
        n += 1
    # the sinusoid table and its per-layer projection depend on t and the weights only: computed once, outside any captured graph
    # (kept inside the weight dict itself, so they live and die with these weights)
    const = W.setdefault(("const", prefix), {})
    ck = (t, x.dtype, str(dev))
    if ck not in const and not (dev.type == "cuda" and torch.cuda.is_current_stream_capturing()):
        pos = sinusoid_table(t, d, dev, x.dtype)[None]
        const[ck] = [F.linear(pos, W[f"{prefix}encoders.{i}.self_attn.linear_pos.weight"]).view(1, t, heads, dk).transpose(1, 2)
                     for i in range(n)]
    pps = const.get(ck)
    pos = None if pps is not None else sinusoid_table(t, d, dev, x.dtype)[None]
    for i in range(n):
        p = f"{prefix}encoders.{i}."
        # --- rel-pos self attention
        y = _ln(x, W, p + "norm_mha")
        q = _lin(y, W, p + "self_attn.linear_q").view(B, t, heads, dk)
        k = _lin(y, W, p + "self_attn.linear_k").view(B, t, heads, dk).transpose(1, 2)
        v = _lin(y, W, p + "self_attn.linear_v").view(B, t, heads, dk).transpose(1, 2)
        pp = pps[i] if pps is not None else F.linear(pos, W[p + "self_attn.linear_pos.weight"]).view(1, t, heads, dk).transpose(1, 2)
        qu = (q + W[p + "self_attn.pos_bias_u"]).transpose(1, 2)
        qv = (q + W[p + "self_attn.pos_bias_v"]).transpose(1, 2)
        sc = (qu @ k.transpose(-1, -2) + qv @ pp.transpose(-1, -2)) / math.sqrt(dk)
        if full:
            att = torch.softmax(sc, dim=-1)
        else:
            km = ~mask[:, None]  # [B,1,1,T'] True = padded
            sc = sc.masked_fill(km, float("-inf"))
            att = torch.softmax(sc, dim=-1).masked_fill(km, 0.0)
        y = (att @ v).transpose(1, 2).reshape(B, t, d)
        x = x + _lin(y, W, p + "self_attn.linear_out")
        # --- convolution module
        y = _ln(x, W, p + "norm_conv").transpose(1, 2)
        if not full:
            y = y.masked_fill(~mask, 0.0)
        y = F.glu(torch.matmul(W[p + "conv_module.pointwise_conv1.weight"][:, :, 0], y)
                  + W[p + "conv_module.pointwise_conv1.bias"][None, :, None], dim=1)
        wd = W[p + "conv_module.depthwise_conv.weight"]                                # [C, 1, k] depthwise
        kd = wd.shape[-1]
        yp = F.pad(y, ((kd - 1) // 2, (kd - 1) // 2))
        y = (yp.unfold(2, kd, 1) * wd[None, :, 0, None, :]).sum(-1) + W[p + "conv_module.depthwise_conv.bias"][None, :, None]
        y = F.silu(_ln(y.transpose(1, 2), W, p + "conv_module.norm")).transpose(1, 2)
        y = torch.matmul(W[p + "conv_module.pointwise_conv2.weight"][:, :, 0], y) \
            + W[p + "conv_module.pointwise_conv2.bias"][None, :, None]
        if not full:
            y = y.masked_fill(~mask, 0.0)
        x = x + y.transpose(1, 2)
        # --- feed forward
        y = _ln(x, W, p + "norm_ff")
        y = _lin(F.silu(_lin(y, W, p + "feed_forward.w_1")), W, p + "feed_forward.w_2")
        x = _ln(x + y, W, p + "norm_final")
    return _ln(x, W, prefix + "after_norm"), mask
