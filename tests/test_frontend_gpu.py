"""The prompt front-end on the HIP kernels (csrc/frontend.hip + gpt/conditioner.py): each kernel against a plain PyTorch fp32
statement of the same operation, and the whole Conformer + Perceiver conditioner against (a) the functional fp32 form of this
package, which the reference-run fixture pins to 1e-3, and (b) that fixture itself.

Tolerances: the kernels store fp16 / bf16 and accumulate in fp32 -- a kernel is held to a few units of its storage type's
resolution on O(1) values (fp16 2^-11, bf16 2^-8); the whole fp16 conditioner to 1e-2 max-abs on latents of RMS ~1."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import synth
import weights

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"
RES = {torch.float16: 2.0 ** -11, torch.bfloat16: 2.0 ** -8}


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,Fq,C", [(300, 100, 512), (41, 20, 64), (3, 5, 8)])
def test_subsample_conv_matches_conv2d(dtype, T, Fq, C):
    from indextts import _native as nat
    mel = rnd(T, Fq, seed=1, scale=2.0)
    w, b = rnd(C, 9, seed=2, scale=0.3), rnd(C, seed=3, scale=0.1)
    t, f2 = (T - 3) // 2 + 1, (Fq - 3) // 2 + 1
    y = torch.empty(t, C * f2, dtype=dtype, device=DEV)
    nat.subsample_conv(mel, w, b, y)
    ref = F.relu(F.conv2d(mel[None, None], w.view(C, 1, 3, 3), b, stride=2))[0]          # [C, t, f2]
    ref = ref.permute(1, 0, 2).reshape(t, C * f2)
    assert (y.float() - ref).abs().max().item() <= 2 * RES[dtype] * max(1.0, ref.abs().max().item())


def mha_ref(q, k, v, H, scale, pos=None, u=None, vb=None):
    Tq, Tk = q.shape[0], k.shape[0]
    qh = q.float().view(Tq, H, 64).transpose(0, 1)
    kh = k.float().view(Tk, H, 64).transpose(0, 1)
    vh = v.float().view(Tk, H, 64).transpose(0, 1)
    if pos is None:
        sc = qh @ kh.transpose(-1, -2)
    else:
        sc = (qh + u.view(H, 1, 64)) @ kh.transpose(-1, -2) + (qh + vb.view(H, 1, 64)) @ pos.float().transpose(-1, -2)
    return (torch.softmax(sc * scale, -1) @ vh).transpose(0, 1).reshape(Tq, H * 64)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("Tq,Tk,H,rel", [(149, 149, 8, True), (32, 181, 8, False), (5, 5, 2, True), (70, 450, 3, True),
                                         (16, 193, 1, False), (1, 1, 1, False)])
def test_mha_small_matches_torch(dtype, Tq, Tk, H, rel):
    """Keys past one 192-row LDS chunk, ragged last 32-key step, one-row edge; with and without the relative-position term."""
    from indextts import _native as nat
    d = H * 64
    buf = (rnd(max(Tq, Tk), 3 * d, seed=5) * 0.7).to(dtype)
    q, k, v = buf[:Tq, :d], buf[:Tk, d:2 * d], buf[:Tk, 2 * d:]
    pos = (rnd(H, Tk, 64, seed=6) * 0.5).to(dtype) if rel else None
    u, vb = (rnd(d, seed=7) * 0.2, rnd(d, seed=8) * 0.2) if rel else (None, None)
    mtp = (Tq + 15) // 16
    out = torch.zeros(mtp * 16 * d, dtype=dtype, device=DEV)
    nat.mha_small(buf, buf[:, d:], buf[:, 2 * d:], out, Tq, Tk, H, 3 * d, 3 * d, 3 * d, mtp, 0.125, pos=pos, bias_u=u, bias_v=vb)
    got = nat.unpack_activation(out, Tq, d).float()
    ref = mha_ref(q, k, v, H, 0.125, pos, u, vb)
    assert torch.isfinite(got).all()
    # probabilities and q + u are rounded to the storage type before their products
    assert (got - ref).abs().max().item() <= 12 * RES[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,C,taps", [(149, 512, 15), (9, 128, 15), (40, 256, 7), (33, 128, 31)])
def test_glu_dwconv_ln_silu_matches_torch(dtype, T, C, taps):
    from indextts import _native as nat
    x = rnd(T, 2 * C, seed=9).to(dtype)
    w, b = rnd(C, taps, seed=10, scale=0.3), rnd(C, seed=11, scale=0.1)
    lw, lb = 1.0 + rnd(C, seed=12, scale=0.1), rnd(C, seed=13, scale=0.1)
    mtp = (T + 15) // 16
    y = torch.zeros(mtp * 16 * C, dtype=dtype, device=DEV)
    nat.glu_dwconv_ln_silu(x, w, b, lw, lb, y, T, C, mtp)
    g = F.glu(x.float().t()[None], dim=1)                                               # [1, C, T]
    c = F.conv1d(g, w[:, None, :], b, padding=(taps - 1) // 2, groups=C)[0].t()           # [T, C]
    ref = F.silu(F.layer_norm(c, (C,), lw, lb, 1e-5))
    got = nat.unpack_activation(y, T, C).float()
    assert (got - ref).abs().max().item() <= 4 * RES[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_rows_modes(dtype):
    from indextts import _native as nat
    M, D = 37, 1280
    x = rnd(M, D, seed=14)
    slab = rnd(5, M, D, seed=15)
    bias, w, b = rnd(D, seed=16), 1.0 + rnd(D, seed=17, scale=0.1), rnd(D, seed=18, scale=0.1)
    # slabs + bias (no x), packed copy inside a taller operand
    y = torch.empty(M, D, device=DEV)
    yp = torch.zeros(5 * 16 * D, dtype=dtype, device=DEV)
    nat.rows(M, D, dtype, slab=slab, nslab=5, bias=bias, y=y, y_packed=yp, y_row0=32, y_mtp=5)
    ref = bias + slab.sum(0)
    assert (y - ref).abs().max().item() < 1e-5
    up = nat.unpack_activation(yp, 80, D)
    assert torch.equal(up[32:32 + M], ref.to(dtype)) or (up[32:32 + M].float() - ref).abs().max().item() <= RES[dtype] * 8
    assert (up[:32] == 0).all() and (up[32 + M:] == 0).all()
    # LayerNorm in place
    xi = x.clone()
    nat.rows(M, D, dtype, x=xi, norm=1, w=w, b=b, y=xi)
    assert (xi - F.layer_norm(x, (D,), w, b, 1e-5)).abs().max().item() < 1e-4
    # l2-normalise * sqrt(D) * gamma
    out = torch.empty(M, D, device=DEV)
    nat.rows(M, D, dtype, x=x, norm=2, w=w, y=out)
    assert (out - F.normalize(x, dim=-1) * math.sqrt(D) * w).abs().max().item() < 1e-4
    with pytest.raises(nat.NativeError):
        nat.rows(M, 2052, dtype, x=x, y=out)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_geglu(dtype):
    from indextts import _native as nat
    M, Kp = 32, 3424
    h = rnd(M, 2 * Kp, seed=19).to(dtype)
    y = torch.zeros(M * Kp, dtype=dtype, device=DEV)
    nat.geglu(h, y, M, Kp)
    ref = F.gelu(h[:, Kp:].float()) * h[:, :Kp].float()
    assert (nat.unpack_activation(y, M, Kp).float() - ref).abs().max().item() <= 2 * RES[dtype] * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_skinny_many_rows_one_launch(dtype):
    """M = 149 rows with the row tiles dealt to grid.z (one launch): LayerNorm-folded + SiLU into a packed y at a row offset,
    the residual epilogue, and a packed x that is the head of a taller operand (x_mtp)."""
    from indextts import _native as nat
    from indextts.gpt.conditioner import _fold
    M, K, N = 149, 512, 2048
    x = rnd(M, K, seed=20) * 3.0 + 1.5
    W, bias = rnd(K, N, seed=21, scale=K ** -0.5), rnd(N, seed=22, scale=0.1)
    gamma, beta = 1.0 + rnd(K, seed=23, scale=0.1), rnd(K, seed=24, scale=0.1)
    wp, c, d = _fold(W, bias, gamma, beta, dtype)
    xp = nat.pack_activation(x.to(dtype))
    mt_all = 14
    y = torch.zeros(mt_all * 16 * N, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, d, x=xp, epi=nat.EPI_SILU_STORE, y=y, x_packed=True, y_packed=True, ln_c=c, rows_per_wg=32,
                    y_row0=32, y_mtp=mt_all)
    ref = F.silu(F.layer_norm(x.to(dtype).float(), (K,), gamma, beta, 1e-5) @ W + bias)
    got = nat.unpack_activation(y, mt_all * 16, N)
    assert (got[32:32 + M].float() - ref).abs().max().item() <= 16 * RES[dtype] * max(1.0, ref.abs().max().item())
    assert (got[:32] == 0).all() and (got[32 + M + 16:] == 0).all()
    pl = nat.skinny_plan(dtype, M, N, K, 1, 32, False, True)
    assert pl["grid"][2] == 5 and pl["row_tiles_per_wg"] == 2
    # residual epilogue over the same rows, unfolded weight; x = the first 149 rows of a 12-tile operand
    wp2 = nat.pack_weight(W[:, :512].to(dtype).contiguous())
    big = torch.zeros(12 * 16, K, dtype=dtype, device=DEV)
    big[:M] = x.to(dtype)
    big[M:] = 7.0
    h = rnd(M, 512, seed=25)
    h0 = h.clone()
    hb = torch.zeros(10 * 16 * 512, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, 512, K, wp2, bias[:512].contiguous(), x=nat.pack_activation(big), epi=nat.EPI_RESID_F32, yf=h, y=hb,
                    x_packed=True, y_packed=True, rows_per_wg=32, x_mtp=12)
    ref2 = h0 + x.to(dtype).float() @ W[:, :512].to(dtype).float() + bias[:512]
    assert (h - ref2).abs().max().item() <= 1e-3 * max(1.0, ref2.abs().max().item())
    assert torch.equal(nat.unpack_activation(hb, M, 512), h.to(dtype))
    with pytest.raises(nat.NativeError):
        nat.gemm_skinny(dtype, M, 512, K, wp2, None, x=xp, epi=nat.EPI_STORE, y=hb, x_packed=True, y_row0=16)   # y_row0 needs a packed y


def _cond_model(dtype):
    from indextts.gpt.model import UnifiedVoice
    m = UnifiedVoice(**dict(weights.reference_config()["gpt"], layers=2))
    m.load_state_dict(weights.gpt_state_dict(2))
    return m.to(DEV).to(dtype)


@pytest.mark.parametrize("frames", [120, 300, 437, 1400])
def test_conditioner_engine_matches_functional_fp32(frames):
    """The whole Conformer + Perceiver conditioner (fp16 kernels) against the functional fp32 form over the same weights;
    437 / 1400 frames: 218 / 699 context rows, many 32-key steps per wave in both attentions, 44 row groups in the GEMMs."""
    m = _cond_model(torch.bfloat16)
    mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, frames), -6.0, 2.0)).to(DEV)
    got = m.get_conditioning(mel, None)
    assert m.conditioner() is not None and m.conditioner().launches == 3 + 9 * 6 + 2 + 6 * 2 + 1
    os.environ["ITTS_NATIVE_CONDITIONER"] = "0"
    try:
        ref = m.get_conditioning(mel, None)
    finally:
        del os.environ["ITTS_NATIVE_CONDITIONER"]
    assert got.shape == ref.shape == (1, 32, 1280) and torch.isfinite(got).all()
    err = (got - ref).abs()
    assert err.max().item() < 1e-2 and err.pow(2).mean().sqrt().item() < 2e-3, (err.max().item(), ref.pow(2).mean().sqrt().item())


def test_conditioner_engine_against_reference_fixture():
    """conds of the reference run (tests/golden/gpt_small.npz, made by make_golden.py from the reference's own modules)."""
    g = np.load(os.path.join(G, "gpt_small.npz"))
    m = _cond_model(torch.bfloat16)
    mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    got = m.get_conditioning(mel, None)
    ref = torch.from_numpy(g["conds"]).to(DEV)
    assert (got - ref).abs().max().item() < 1e-2
    # a batch of equal-length prompts is a loop over prompts; the graph-captured form replays to the same bits
    two = m.get_conditioning(torch.cat([mel, mel * 0.5]), None)
    assert torch.equal(two[0], got[0]) and not torch.equal(two[1], got[0])
    gr = torch.cuda.CUDAGraph()
    eng = m.conditioner()
    row = mel[0].t().contiguous()
    with torch.cuda.graph(gr):
        out = eng(row)
    out.zero_()
    gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, got[0])


def test_conditioner_fp32_mode_keeps_the_functional_form():
    m = _cond_model(torch.float32)
    assert m.conditioner() is None
    with pytest.raises(ValueError):
        from indextts.gpt.conditioner import ConditionerEngine
        ConditionerEngine(m._cond_weights(), dtype=torch.float32)


def _prefix_ref(t, c, text_emb, text_pos, start, stop):
    """prepare_gpt_inputs (reference model.py:606-667) in plain torch ops."""
    B, L = t.shape
    D, C = c.shape[-1], c.shape[1]
    P = C + L + 2
    valid = (t != stop) & (t != start)
    n = valid.sum(dim=1)
    rank = torch.cumsum(valid, dim=1) - 1
    tok = torch.full((B, L + 2), stop, dtype=torch.long, device=t.device)
    tok[:, 0] = start
    rows = torch.arange(B, device=t.device)[:, None].expand(B, L)
    tok[rows[valid], (rank + 1)[valid]] = t[valid]
    te = text_emb[tok] + text_pos[: L + 2][None]
    if c.shape[0] == 1 and B > 1:
        c = c.expand(B, -1, -1)
    row = torch.cat([c, te], dim=1)
    pad = L - n
    idx = torch.arange(P, device=t.device)[None, :] - pad[:, None]
    keep = idx >= 0
    emb = row.gather(1, idx.clamp(min=0)[:, :, None].expand(-1, -1, D)) * keep[:, :, None]
    mask = torch.cat([keep.long(), torch.ones(B, 1, dtype=torch.long, device=t.device)], dim=1)
    return emb, mask, pad.to(torch.int32)


@pytest.mark.parametrize("B,L,C,D,shared", [(32, 60, 32, 1280, True), (3, 7, 32, 64, False), (2, 300, 4, 128, True), (1, 1, 0, 8, True)])
def test_prefix_rows_matches_torch(B, L, C, D, shared):
    """Ids with start / stop ids anywhere (stripped, order kept), rows of different length, more ids than one scan pass."""
    from indextts import _native as nat
    g = torch.Generator().manual_seed(B * 1000 + L)
    start, stop, V = 0, 1, 500
    t = torch.randint(2, V, (B, L), generator=g)
    for b in range(B):
        n = int(torch.randint(0, L + 1, (1,), generator=g))
        t[b, n:] = stop
        if L > 4:
            t[b, int(torch.randint(0, L, (1,), generator=g))] = start     # a stray start id inside the text
    t = t.to(DEV)
    conds = rnd(1 if shared else B, C, D, seed=31)
    te, tp = rnd(V, D, seed=32), rnd(L + 2, D, seed=33)
    emb, mask, pad = nat.prefix_rows(t, conds, te, tp, start, stop)
    remb, rmask, rpad = _prefix_ref(t, conds, te, tp, start, stop)
    assert torch.equal(mask, rmask) and torch.equal(pad, rpad)
    assert torch.equal(emb, remb)


# ---------------------------------------------------------------------------------------------------- speaker encoder
@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_im2col_reflect_and_res2_step(dtype):
    """The two convolution forms of the speaker encoder against F.conv1d over a reflect-padded signal."""
    from indextts import _native as nat
    T, Fq, C = 45, 100, 512
    mtp = (T + 15) // 16
    x = rnd(T, Fq, seed=40)
    Kp = 512
    col = torch.zeros(mtp * 16 * Kp, dtype=dtype, device=DEV)
    nat.im2col_reflect(x, col, 5, 1, Kp, mtp)
    got = nat.unpack_activation(col, T, Kp).float()
    xp = F.pad(x.t()[None], (2, 2), mode="reflect")[0].t()                    # [T + 4, F]
    ref = torch.cat([xp[j:j + T] for j in range(5)], dim=1)                    # [T, 500]
    assert torch.equal(got[:, :500], ref.to(dtype).float()) and (got[:, 500:] == 0).all()
    # one Res2Net step: chunk 3 of y1 plus chunk 2 of cat, dilation 3
    y1 = rnd(T, C, seed=41).to(dtype)
    cat = rnd(T, C, seed=42).to(dtype)
    w, b = rnd(64, 64, 3, seed=43, scale=0.08), rnd(64, seed=44, scale=0.1)
    sc, sh = 1.0 + rnd(64, seed=45, scale=0.1), rnd(64, seed=46, scale=0.1)
    wp = nat.pack_weight(w.permute(2, 1, 0).reshape(192, 64).to(dtype).contiguous())
    for first in (False, True):
        y1p, catp = nat.pack_activation(y1), nat.pack_activation(cat)
        nat.res2_step(y1p, catp, wp, b, sc, sh, T, mtp, 3, 3, first)
        out = nat.unpack_activation(catp, T, C)
        inp = y1[:, 192:256].float() if first else (y1[:, 192:256].float() + cat[:, 128:192].float()).to(dtype).float()
        z = F.conv1d(F.pad(inp.t()[None], (3, 3), mode="reflect"), w.to(dtype).float(), b, dilation=3)[0].t()
        ref = F.relu(z) * sc + sh
        assert (out[:, 192:256].float() - ref).abs().max().item() <= 6 * RES[dtype] * max(1.0, ref.abs().max().item())
        keep = torch.ones(C, dtype=torch.bool)
        keep[192:256] = False
        if first:
            keep[:64] = False
            assert torch.equal(out[:, :64], y1[:, :64])
        assert torch.equal(out[:, keep.to(DEV)], cat[:, keep.to(DEV)])


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_se_gate_scale_resid_col_stats(dtype):
    from indextts import _native as nat
    T, C, H = 77, 512, 128
    mtp = (T + 15) // 16
    y, res = rnd(T, C, seed=50).to(dtype), rnd(T, C, seed=51).to(dtype)
    w1, b1 = rnd(H, C, seed=52, scale=C ** -0.5).to(dtype), rnd(H, seed=53, scale=0.1)
    w2, b2 = rnd(C, H, seed=54, scale=H ** -0.5).to(dtype), rnd(C, seed=55, scale=0.1)
    gate = torch.empty(C, device=DEV)
    yp = nat.pack_activation(y)
    nat.se_gate(yp, w1, b1, w2, b2, gate, T, C, H, mtp)
    ref_g = torch.sigmoid(F.relu(y.float().mean(0) @ w1.float().t() + b1) @ w2.float().t() + b2)
    assert (gate - ref_g).abs().max().item() < 1e-4
    wide = torch.zeros(mtp * 16 * 3 * C, dtype=dtype, device=DEV)             # a [T][3 C] operand; write its middle third
    off = (C // 32) * mtp * 512
    nat.scale_resid(yp, nat.pack_activation(res), gate, wide[off:], T, C, mtp)
    got = nat.unpack_activation(wide, T, 3 * C)
    ref = gate * y.float() + res.float()
    assert (got[:, C:2 * C].float() - ref).abs().max().item() <= 2 * RES[dtype] * max(1.0, ref.abs().max().item())
    assert (got[:, :C] == 0).all() and (got[:, 2 * C:] == 0).all()
    # statistics: plain and softmax-weighted + affine
    x = (rnd(T, C, seed=56) * 1.5 + 0.7).to(dtype)
    xp = nat.pack_activation(x)
    out = torch.empty(2 * C, dtype=dtype, device=DEV)
    nat.col_stats(xp, out, T, C, mtp)
    xf = x.float()
    m = xf.mean(0)
    sd = torch.sqrt(((xf - m) ** 2).mean(0).clamp(1e-12))
    assert (out.float() - torch.cat([m, sd])).abs().max().item() <= 2 * RES[dtype] * 4
    logit = rnd(T, C, seed=57, scale=2.0).to(dtype)
    sc, sh = 1.0 + rnd(2 * C, seed=58, scale=0.1), rnd(2 * C, seed=59, scale=0.1)
    nat.col_stats(xp, out, T, C, mtp, logit=logit, scale=sc, shift=sh)
    a = torch.softmax(logit.float(), dim=0)
    m = (a * xf).sum(0)
    sd = torch.sqrt((a * (xf - m) ** 2).sum(0).clamp(1e-12))
    ref = torch.cat([m, sd]) * sc + sh
    assert (out.float() - ref).abs().max().item() <= 2 * RES[dtype] * 4


def _vocoder(dtype):
    from indextts.BigVGAN.models import BigVGAN
    from indextts.utils.config import Config
    v = BigVGAN(Config(weights.reference_config()["bigvgan"]))
    v.load_state_dict(weights.bigvgan_state_dict())
    v.to(DEV).to(dtype).remove_weight_norm()
    return v


@pytest.mark.parametrize("frames", [300, 57, 1000])
def test_speaker_engine_matches_functional_fp32(frames):
    v = _vocoder(torch.float16)
    mel = torch.from_numpy(synth.uniform("in.ref_mel", (2, frames, 100), -6.0, 2.0)).to(DEV)
    mel[1] *= 0.5
    got = v.speaker_embedding(mel)
    assert v.speaker_engine() is not None and v.speaker_engine().launches == 2 + 3 * 11 + 7
    os.environ["ITTS_NATIVE_SPEAKER"] = "0"
    try:
        ref = v.speaker_embedding(mel)
    finally:
        del os.environ["ITTS_NATIVE_SPEAKER"]
    assert got.shape == ref.shape == (2, 1, 512) and torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err < 2e-2 * max(1.0, ref.abs().max().item()), (err, ref.abs().max().item())
    assert not torch.equal(got[0], got[1])


def test_speaker_engine_against_reference_fixture():
    """spk4 of the reference run (tests/golden/bigvgan.npz)."""
    g = np.load(os.path.join(G, "bigvgan.npz"))
    v = _vocoder(torch.float16)
    got = v.speaker_embedding(torch.from_numpy(g["melref"]).to(DEV))
    ref = torch.from_numpy(g["spk4"]).to(DEV)
    assert (got - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    assert _vocoder(torch.float32).speaker_engine() is None


def test_front_end_buffers_are_per_thread_and_forgettable():
    """Replicas of one model (RequestPool: a thread + stream each) share the engines' packed weights, never their activations."""
    import threading
    m = _cond_model(torch.bfloat16)
    mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(DEV)
    a = m.get_conditioning(mel, None)
    eng = m.conditioner()
    assert len(eng._bufs) == 1
    out = {}

    def other():
        with torch.cuda.stream(torch.cuda.Stream()):
            out["b"] = m.get_conditioning(mel, None)
            torch.cuda.current_stream().synchronize()
            out["n"] = len(eng._bufs)
            eng.forget()
            out["after"] = len(eng._bufs)
    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert out["n"] == 2 and out["after"] == 1 and torch.equal(out["b"], a)
    eng.forget()
    assert len(eng._bufs) == 0
    assert torch.equal(m.get_conditioning(mel, None), a)


def test_indextts_prompt_features_run_on_the_hip_engines():
    """The product path of a 16-bit IndexTTS: _prompt_conds / _prompt_spk (first call eager, second captured, third replayed) go
    through ConditionerEngine / SpeakerEngine -- no silent fall-back to the functional PyTorch forms -- and replay to the bits of
    the eager call; the prompt rows come from itts_prefix_rows with the padding as a host tensor for host-resident ids."""
    from indextts.infer import IndexTTS
    import copy
    cfg = copy.deepcopy(weights.reference_config())
    cfg["gpt"]["layers"] = 2
    tts = IndexTTS.from_weights(cfg, weights.gpt_state_dict(2), weights.bigvgan_state_dict(), device=DEV,
                                precision_config={"gpt": "bf16", "vocoder": "fp16"})
    mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 150), -6.0, 2.0)).to(DEV)
    ce, se = tts.gpt.conditioner(), tts.bigvgan.speaker_engine()
    assert ce is not None and se is not None and ce.launches == 0 and se.launches == 0
    c = [tts._prompt_conds(mel).clone() for _ in range(3)]
    s = [tts._prompt_spk(mel).clone() for _ in range(3)]
    assert ce.launches == 72 and se.launches == 42
    assert torch.equal(c[0], c[1]) and torch.equal(c[1], c[2]) and c[0].shape == (1, 32, 1280)
    assert torch.equal(s[0], s[1]) and torch.equal(s[1], s[2]) and s[0].shape == (1, 1, 512)
    assert all(isinstance(v, tuple) for v in tts._feat_graphs.values())          # both networks captured, none fell back to eager
    ids = torch.tensor([[5, 6, 7, 1, 1], [9, 8, 7, 6, 5]], dtype=torch.int32)
    emb, pad = tts.gpt.prefix_rows(c[0], ids)
    assert not pad.is_cuda and pad.tolist() == [2, 0] and emb.shape == (2, 32 + 5 + 2, 1280)
    emb_d, pad_d = tts.gpt.prefix_rows(c[0], ids.to(DEV))
    assert pad_d.is_cuda and pad_d.tolist() == [2, 0] and torch.equal(emb_d, emb)
