"""Perceiver resampler (host-side PyTorch-ROCm): 32 learned latents cross-attend to the Conformer output.

Functional restatement over the reference's `perceiver_encoder.*` keys.  Follows indextts/gpt/perceiver.py:219-269
(PerceiverResampler: proj_context, depth 2, no pre-norm, final RMSNorm = normalize * sqrt(dim) * gamma) and :271-312
(Attention with cross_attn_include_queries: keys/values = cat(latents, context), 8 heads x 64, no biases), GEGLU FFN
(gelu(gate) * x, :181-193)."""
import torch
import torch.nn.functional as F


def perceiver_resample(W: dict, ctx: torch.Tensor, ctx_mask: torch.Tensor, heads: int = 8, prefix="perceiver_encoder."):
    """ctx [B,T',512], ctx_mask [B,32+T'] bool (True = attend; None = attend everywhere) -> conds [B,32,1280]."""
    B = ctx.shape[0]
    if prefix + "proj_context.weight" in W:
        ctx = F.linear(ctx, W[prefix + "proj_context.weight"], W[prefix + "proj_context.bias"])
    lat = W[prefix + "latents"][None].expand(B, -1, -1)
    n = 0
    while f"{prefix}layers.{n}.0.to_q.weight" in W:
        n += 1
    for i in range(n):
        p = f"{prefix}layers.{i}."
        kv_in = torch.cat([lat, ctx], dim=1)
        q = F.linear(lat, W[p + "0.to_q.weight"])
        k, v = F.linear(kv_in, W[p + "0.to_kv.weight"]).chunk(2, dim=-1)
        dh = q.shape[-1] // heads
        q, k, v = (t.view(B, -1, heads, dh).transpose(1, 2) for t in (q, k, v))
        sim = (q @ k.transpose(-1, -2)) * dh ** -0.5
        if ctx_mask is not None:                          # None: nothing is padded
            sim = sim.masked_fill(~ctx_mask[:, None, None, :], -torch.finfo(sim.dtype).max)
        o = (torch.softmax(sim, dim=-1) @ v).transpose(1, 2).reshape(B, -1, heads * dh)
        lat = lat + F.linear(o, W[p + "0.to_out.weight"])
        h = F.linear(lat, W[p + "1.0.weight"], W[p + "1.0.bias"])
        x, gate = h.chunk(2, dim=-1)
        lat = lat + F.linear(F.gelu(gate) * x, W[p + "1.2.weight"], W[p + "1.2.bias"])
    return F.normalize(lat, dim=-1) * (lat.shape[-1] ** 0.5) * W[prefix + "norm.gamma"]
