"""ORACLE (test infrastructure, NOT product code): CPU fp32 restatement of the IndexTTS GPT passes.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  It restates, in
plain torch-CPU ops, the arithmetic of

  * HF GPT-2 blocks as built by indextts/gpt/model.py:263-286 (pre-LN, Conv1D [in,out] + bias, gelu_new,
    eps=1e-5, wpe nulled) -- transformers 4.44.2 modeling_gpt2.py (third-party, pinned in requirements.txt:5);
  * UnifiedVoice.prepare_gpt_inputs               indextts/gpt/model.py:606-667
  * GPT2InferenceModel.forward (prefill / cached)  indextts/gpt/model.py:125-205  (position quirk :163-167)
  * UnifiedVoice.forward(return_latent=True)       indextts/gpt/model.py:548-597, get_logits :459-474

Parity pinning: checked against tests/golden/gpt_small.npz and gpt_full.npz, which were produced by running the
reference itself (tests/golden/make_golden.py) -- see tests/test_oracle_vs_golden.py.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-5
HEAD_DIM = 64


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, EPS)


def gelu_new(x):
    # transformers activations.NewGELUActivation
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


def block(h, W, i, past_kv, key_valid):
    """One GPT-2 block.  h [B,S,D]; past_kv = (k,v) [B,H,P,64] or None; key_valid [B,P+S] bool (attention_mask).

    Query s (absolute position P+s) sees key j iff j <= P+s and key_valid[b,j].  Rows with no visible key
    (left-pad positions) get a zero attention output: their state is never read by a valid row.
    """
    B, S, D = h.shape
    H = D // HEAD_DIM
    p = f"gpt.h.{i}."
    x = layer_norm(h, W[p + "ln_1.weight"], W[p + "ln_1.bias"])
    qkv = x @ W[p + "attn.c_attn.weight"] + W[p + "attn.c_attn.bias"]
    q, k, v = qkv.split(D, dim=-1)
    q = q.view(B, S, H, HEAD_DIM).transpose(1, 2)
    k = k.view(B, S, H, HEAD_DIM).transpose(1, 2)
    v = v.view(B, S, H, HEAD_DIM).transpose(1, 2)
    if past_kv is not None:
        k = torch.cat([past_kv[0], k], dim=2)
        v = torch.cat([past_kv[1], v], dim=2)
    P = k.shape[2] - S
    scores = (q @ k.transpose(-1, -2)) / math.sqrt(HEAD_DIM)  # [B,H,S,P+S]
    qpos = torch.arange(P, P + S)[:, None]
    kpos = torch.arange(P + S)[None, :]
    vis = (kpos <= qpos)[None, None] & key_valid[:, None, None, :]
    scores = scores.masked_fill(~vis, float("-inf"))
    any_vis = vis.any(-1, keepdim=True)
    probs = torch.softmax(scores.masked_fill(~any_vis, 0.0), dim=-1)
    probs = torch.where(any_vis, probs, torch.zeros_like(probs))
    a = (probs @ v).transpose(1, 2).reshape(B, S, D)
    h = h + a @ W[p + "attn.c_proj.weight"] + W[p + "attn.c_proj.bias"]
    x = layer_norm(h, W[p + "ln_2.weight"], W[p + "ln_2.bias"])
    f = gelu_new(x @ W[p + "mlp.c_fc.weight"] + W[p + "mlp.c_fc.bias"])
    h = h + f @ W[p + "mlp.c_proj.weight"] + W[p + "mlp.c_proj.bias"]
    return h, (k, v)


def n_layers(W):
    n = 0
    while f"gpt.h.{n}.ln_1.weight" in W:
        n += 1
    return n


def transformer(emb, W, key_valid, past=None):
    """GPT2Model.forward(inputs_embeds=emb) with nulled wpe: blocks then ln_f.  Returns (hidden, new_past)."""
    h = emb
    new_past = []
    for i in range(n_layers(W)):
        h, kv = block(h, W, i, None if past is None else past[i], key_valid)
        new_past.append(kv)
    return layer_norm(h, W["gpt.ln_f.weight"], W["gpt.ln_f.bias"]), new_past


def mel_head(hidden, W):
    """lm_head = Sequential(final_norm, mel_head)  (model.py:56,193)."""
    x = layer_norm(hidden, W["final_norm.weight"], W["final_norm.bias"])
    return x @ W["mel_head.weight"].t() + W["mel_head.bias"]


def prepare_gpt_inputs(conds, text, W, start_text=0, stop_text=1):
    """model.py:606-667.  conds [1 or B,32,D]; text [B,L] int64 -> (prefix_emb [B,P,D], mask [B,P+1] bool, pad [B])."""
    B, L = text.shape
    D = conds.shape[-1]
    P = conds.shape[1] + L + 2
    emb = torch.zeros(B, P, D)
    mask = torch.ones(B, P + 1, dtype=torch.bool)
    pads = []
    for b in range(B):
        t = text[b][(text[b] != stop_text) & (text[b] != start_text)]
        t = torch.cat([torch.tensor([start_text]), t, torch.tensor([stop_text])])
        te = W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]
        c = conds[0] if conds.shape[0] == 1 else conds[b]
        row = torch.cat([c, te], dim=0)
        pad = P - row.shape[0]
        emb[b, pad:] = row
        mask[b, :pad] = False
        pads.append(pad)
    return emb, mask, torch.tensor(pads)


def decode_prefill(prefix_emb, mask, W, start_mel=8192):
    """First generate() step: prefix + start token at mel position 0 (model.py:152-162).  -> logits[B,V], past."""
    B = prefix_emb.shape[0]
    start = W["mel_embedding.weight"][start_mel] + W["mel_pos_embedding.emb.weight"][0]
    emb = torch.cat([prefix_emb, start.expand(B, 1, -1)], dim=1)
    hidden, past = transformer(emb, W, mask, None)
    return mel_head(hidden[:, -1], W), past


def decode_step(tokens, k_index, mask, past, W):
    """Cached step for the k-th generated token (k>=1 is the first sampled one): mel position k+1
    (= attention_mask.shape[1] - mel_len, model.py:163-167; position 1 is never used).  mask already grown."""
    emb = W["mel_embedding.weight"][tokens] + W["mel_pos_embedding.emb.weight"][k_index + 1]
    hidden, past = transformer(emb[:, None, :], W, mask, past)
    return mel_head(hidden[:, -1], W), past


def latent_pass(conds, text_row, codes_row, W, start_text=0, stop_text=1, start_mel=8192, stop_mel=8193):
    """UnifiedVoice.forward(..., return_latent=True) for ONE utterance whose codes are all valid
    (wav_lengths = len(codes)*1024, as infer.py:864-874 calls it).  -> latent [1, T, D].

    text  -> pad stop, then start | ... | stop        (model.py:575-581)
    codes -> pad stop (set_mel_padding is a no-op for full-length rows), pad stop again, then start | ... (model.py:573-588)
    output = final_norm(gpt(cat(conds, text_emb, mel_emb)))[mel part][:, :-2]   (model.py:459-474,593)
    """
    t = torch.cat([torch.tensor([start_text]), text_row.reshape(-1), torch.tensor([stop_text])])
    te = W["text_embedding.weight"][t] + W["text_pos_embedding.emb.weight"][: t.numel()]
    # mel_codes after both pads: codes + [stop] ; aligned input = [start] + that
    m = torch.cat([torch.tensor([start_mel]), codes_row.reshape(-1), torch.tensor([stop_mel])])
    me = W["mel_embedding.weight"][m] + W["mel_pos_embedding.emb.weight"][: m.numel()]
    emb = torch.cat([conds[0], te, me], dim=0)[None]
    valid = torch.ones(1, emb.shape[1], dtype=torch.bool)
    hidden, _ = transformer(emb, W, valid, None)
    enc = layer_norm(hidden[:, conds.shape[1]:], W["final_norm.weight"], W["final_norm.bias"])
    mel_part = enc[:, -m.numel():]
    return mel_part[:, :-2]
