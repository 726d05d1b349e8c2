"""ECAPA-TDNN speaker encoder (host-side PyTorch-ROCm; once per prompt, cacheable).

Functional restatement over the reference's `speaker_encoder.*` keys.  Follows indextts/BigVGAN/ECAPA_TDNN.py:470-581
(TDNN k5 -> 3 x SE-Res2Net(scale 8, k3, dilation 2/3/4) -> MFA -> attentive statistics pooling with global context ->
BatchNorm -> 1x1 conv), TDNNBlock = BN(ReLU(conv)) (:79-130), reflect "same" padding of d*(k-1)//2 per side
(nnet/CNN.py:430-433,458-488), eval-mode BatchNorm (nnet/normalization.py:13-108).  `lengths` is None on the
inference path (models.py:204 passes lens=None), so every frame is valid."""
import torch
import torch.nn.functional as F

EPS_BN = 1e-5


def _conv(x, W, p, dilation=1):
    """Reflect-'same' dilated Conv1d written as k shifted matmuls (tiny problem sizes: avoids MIOpen's generic conv path)."""
    w = W[p + ".conv.weight"]                      # [Cout, Cin, k]
    k = w.shape[-1]
    pad = dilation * (k - 1) // 2
    if pad:
        x = F.pad(x, (pad, pad), mode="reflect")
    T = x.shape[-1] - dilation * (k - 1)
    y = W[p + ".conv.bias"][None, :, None]
    for j in range(k):
        y = y + torch.matmul(w[:, :, j], x[:, :, j * dilation: j * dilation + T])
    return y


def _bn(x, W, p):
    return F.batch_norm(x, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.0, EPS_BN)


def _tdnn(x, W, p, dilation=1):
    return _bn(F.relu(_conv(x, W, p + ".conv", dilation)), W, p + ".norm.norm")


def ecapa_embed(W: dict, mel_btf: torch.Tensor, prefix="speaker_encoder.") -> torch.Tensor:
    """mel_btf [B,Tref,100] -> [B,1,512]."""
    x = mel_btf.transpose(1, 2)
    x = _tdnn(x, W, prefix + "blocks.0")
    feats = []
    for bi, dil in ((1, 2), (2, 3), (3, 4)):
        p = f"{prefix}blocks.{bi}."
        res = x
        y = _tdnn(x, W, p + "tdnn1")
        chunks = torch.chunk(y, 8, dim=1)
        outs = [chunks[0]]
        prev = None
        for s in range(1, 8):
            inp = chunks[s] if s == 1 else chunks[s] + prev
            prev = _tdnn(inp, W, f"{p}res2net_block.blocks.{s-1}", dil)
            outs.append(prev)
        y = _tdnn(torch.cat(outs, dim=1), W, p + "tdnn2")
        s_ = y.mean(dim=2, keepdim=True)
        s_ = F.relu(_conv(s_, W, p + "se_block.conv1"))
        s_ = torch.sigmoid(_conv(s_, W, p + "se_block.conv2"))
        x = s_ * y + res
        feats.append(x)
    x = _tdnn(torch.cat(feats, dim=1), W, prefix + "mfa")
    # attentive statistics pooling with global context
    L = x.shape[-1]
    mean = x.mean(dim=2, keepdim=True)
    std = torch.sqrt(((x - mean) ** 2).mean(dim=2, keepdim=True).clamp(1e-12))
    a = torch.cat([x, mean.expand(-1, -1, L), std.expand(-1, -1, L)], dim=1)
    a = _conv(torch.tanh(_tdnn(a, W, prefix + "asp.tdnn")), W, prefix + "asp.conv")
    a = torch.softmax(a, dim=2)
    m = (a * x).sum(dim=2)
    s = torch.sqrt((a * (x - m[:, :, None]) ** 2).sum(dim=2).clamp(1e-12))
    pooled = torch.cat([m, s], dim=1)[:, :, None]
    pooled = _bn(pooled, W, prefix + "asp_bn.norm")
    return _conv(pooled, W, prefix + "fc").transpose(1, 2)
