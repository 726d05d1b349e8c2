"""Op-level parity of every HIP kernel (through the C ABI / ctypes) against plain PyTorch fp32 references and the
reference-run fixtures.  Tolerances: fp32 kernels use exact-fp32 MFMA (fmaf-chain numerics) -> 1e-4 abs on O(1) data;
16-bit storage types are compared against the same fp32 reference evaluated on inputs ROUNDED to that type."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
DEV = "cuda"


@pytest.fixture(scope="module")
def nat():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from indextts import _native
    _native.lib()
    return _native


def rnd(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def ref_pack(w, dtype):
    """torch restatement of the packed layout documented in include/indextts_hip.h."""
    taps, K, N = w.shape
    E = 4 if dtype == torch.float32 else 8
    KS = 4 * E
    KT, NT = (K + KS - 1) // KS, (N + 15) // 16
    wpad = torch.zeros(taps, KT * KS, NT * 16, dtype=w.dtype, device=w.device)
    wpad[:, :K, :N] = w
    v = wpad.view(taps, KT, 4, E, NT, 16).permute(0, 4, 1, 2, 5, 3).contiguous()  # [tap][nt][ks][g][c][e]
    return v.view(-1)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_pack_weight(nat, dtype):
    w = rnd(3, 40, 50, seed=1).to(dtype)
    got = nat.pack_weight(w).view(dtype)
    exp = ref_pack(w, dtype)
    assert torch.equal(got, exp)


def test_aa_snake_golden(nat):
    g = np.load(os.path.join(G, "act1d.npz"))
    i = 0
    while f"x{i}" in g:
        x = torch.from_numpy(g[f"x{i}"]).to(DEV)
        a = torch.from_numpy(g[f"alpha{i}"]).to(DEV)
        b = torch.from_numpy(g[f"beta{i}"]).to(DEV)
        y_ref = torch.from_numpy(g[f"y{i}"]).to(DEV)
        y1 = nat.aa_snake(x.contiguous(), a, b, g["up_filter"], g["down_filter"], layout=1)
        assert (y1 - y_ref).abs().max().item() < 2e-5, f"layout1 case {i}"
        y0 = nat.aa_snake(x.transpose(1, 2).contiguous(), a, b, g["up_filter"], g["down_filter"], layout=0)
        assert (y0.transpose(1, 2) - y_ref).abs().max().item() < 2e-5, f"layout0 case {i}"
        for dt_, tol in ((torch.bfloat16, 6e-2), (torch.float16, 8e-3)):
            xh = x.to(dt_)
            yh = nat.aa_snake(xh.transpose(1, 2).contiguous(), a, b, g["up_filter"], g["down_filter"], layout=0)
            yr = nat.aa_snake(xh.float().transpose(1, 2).contiguous(), a, b, g["up_filter"], g["down_filter"], layout=0)
            assert (yh.float() - yr).abs().max().item() < tol
        i += 1
    assert i == 6


def test_layernorm(nat):
    h = rnd(37, 1280, seed=2, scale=3.0) + 0.5
    w, b = rnd(1280, seed=3), rnd(1280, seed=4)
    w2, b2 = rnd(1280, seed=5), rnd(1280, seed=6)
    ref = F.layer_norm(h, (1280,), w, b, 1e-5)
    out = torch.empty_like(h)
    nat.layernorm(h, w, b, out)
    assert (out - ref).abs().max().item() < 2e-5
    ref2 = F.layer_norm(ref, (1280,), w2, b2, 1e-5)
    nat.layernorm(h, w, b, out, w2, b2)
    assert (out - ref2).abs().max().item() < 5e-5
    outh = torch.empty(37, 1280, dtype=torch.bfloat16, device=DEV)
    nat.layernorm(h, w, b, outh)
    assert (outh.float() - ref).abs().max().item() < 0.05


def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1, 1280, 1280), (7, 3840, 1280), (16, 1280, 5120), (32, 5120, 1280), (19, 8194, 1280),
                                   (32, 1280, 5120), (45, 64, 96)])
def test_gemm_skinny_plain(nat, dtype, M, N, K):
    x = rnd(M, K, seed=10).to(dtype)
    w = (rnd(K, N, seed=11) * 0.05).to(dtype)
    bias = rnd(N, seed=12)
    wp = nat.pack_weight(w)
    ref = x.float() @ w.float() + bias
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, bias, x=x, epi=nat.EPI_STORE, y=y)
    assert (y.float() - ref).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    nat.gemm_skinny(dtype, M, N, K, wp, bias, x=x, epi=nat.EPI_GELU_STORE, y=y)
    assert (y.float() - gelu_new(ref)).abs().max().item() < tol * max(1.0, ref.abs().max().item())
    yf = torch.empty(M, N, dtype=torch.float32, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, bias, x=x, epi=nat.EPI_STORE_F32, yf=yf)
    assert (yf - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())
    h0 = rnd(M, N, seed=13)
    hres = h0.clone()
    if N % 4:   # the residual epilogue moves whole 16-byte groups (documented): refused, nothing launched
        with pytest.raises(nat.NativeError):
            nat.gemm_skinny(dtype, M, N, K, wp, None, x=x, epi=nat.EPI_RESID_F32, yf=hres)
        return
    nat.gemm_skinny(dtype, M, N, K, wp, None, x=x, epi=nat.EPI_RESID_F32, yf=hres)
    assert (hres - (h0 + x.float() @ w.float())).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,KSP", [(1, 4), (13, 3), (32, 3), (32, 4)])
def test_ln_reduce_and_splitk_slabs(nat, dtype, M, KSP):
    """out-proj with split-K slabs, then residual-reduce + LayerNorm (+ second LayerNorm), as in one decode block."""
    K, N = 5120, 1280
    x = rnd(M, K, seed=20).to(dtype)
    w = (rnd(K, N, seed=25) * 0.03).to(dtype)
    bias = rnd(N, seed=26)
    wp = nat.pack_weight(w)
    slab = torch.full((KSP, M, N), 123.0, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, None, x=x, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=KSP)
    ref_mm = x.float() @ w.float()
    assert (slab.sum(0) - ref_mm).abs().max().item() < (3e-4 if dtype == torch.float32 else 5e-3) * max(1.0, ref_mm.abs().max().item())
    # each slab is exactly the partial product over its K slice
    kstep = 16 if dtype == torch.float32 else 32
    ks = -(-(K // kstep) // KSP) * kstep          # k-steps are split evenly, the last slice takes the remainder
    for i in range(KSP):
        part = x.float()[:, i * ks:(i + 1) * ks] @ w.float()[i * ks:(i + 1) * ks]
        assert (slab[i] - part).abs().max().item() < (3e-4 if dtype == torch.float32 else 5e-3) * max(1.0, part.abs().max().item())
    h0 = rnd(M, N, seed=27, scale=2.0) + 0.3
    lw, lb = 1 + 0.1 * rnd(N, seed=21), 0.1 * rnd(N, seed=22)
    lw2, lb2 = 1 + 0.1 * rnd(N, seed=23), 0.1 * rnd(N, seed=24)
    h = h0.clone()
    out = torch.empty(M, N, dtype=dtype, device=DEV)
    nat.ln_reduce(h, lw, lb, out, slab=slab, nslab=KSP, bias=bias)
    h_ref = h0 + bias
    for i in range(KSP):
        h_ref = h_ref + slab[i]
    assert torch.equal(h, h_ref)                                  # fixed summation order -> bit-exact
    ref = F.layer_norm(h_ref, (N,), lw, lb, 1e-5)
    tol = 3e-5 if dtype == torch.float32 else 3e-2
    assert (out.float() - ref).abs().max().item() < tol
    h2 = h_ref.clone()
    nat.ln_reduce(h2, lw, lb, out, w2=lw2, b2=lb2)                # no pending update: h untouched
    assert torch.equal(h2, h_ref)
    ref2 = F.layer_norm(ref, (N,), lw2, lb2, 1e-5)
    assert (out.float() - ref2).abs().max().item() < (6e-5 if dtype == torch.float32 else 4e-2)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_skinny_qkv_cache(nat, dtype):
    M, D, H, smax = 5, 1280, 20, 40
    x = rnd(M, D, seed=30).to(dtype)
    w = (rnd(D, 3 * D, seed=33) * 0.03).to(dtype)
    bias = rnd(3 * D, seed=34)
    wp = nat.pack_weight(w)
    ref = x.float() @ w.float() + bias
    q = torch.zeros(M, D, dtype=dtype, device=DEV)
    kc = torch.zeros(M, H, smax, 64, dtype=dtype, device=DEV)
    vc = torch.zeros(M, H, smax, 64, dtype=dtype, device=DEV)
    pos = torch.tensor([17], dtype=torch.int32, device=DEV)
    nat.gemm_skinny(dtype, M, 3 * D, D, wp, bias, x=x, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc, vcache=vc, pos=pos, heads=H,
                    smax=smax)
    tol = 3e-4 if dtype == torch.float32 else 3e-2
    assert (q.float() - ref[:, :D]).abs().max().item() < tol
    assert (kc[:, :, 17, :].reshape(M, D).float() - ref[:, D:2 * D]).abs().max().item() < tol
    assert (vc[:, :, 17, :].reshape(M, D).float() - ref[:, 2 * D:]).abs().max().item() < tol
    kc[:, :, 17, :] = 0
    assert kc.abs().max().item() == 0  # nothing else was touched


CONV_CASES = [
    # (B, T, Cin, Cout, k, dil)
    (1, 37, 1280, 1536, 7, 1), (2, 50, 768, 768, 3, 5), (1, 70, 384, 384, 11, 3), (2, 300, 192, 192, 7, 5),
    (1, 700, 96, 96, 11, 5), (2, 1000, 48, 48, 3, 1), (1, 1500, 24, 24, 11, 5), (2, 900, 24, 1, 7, 1),
    (1, 5, 96, 96, 3, 3),
    # narrow persistent kernel: tile tails, several batch rows, every (k, dilation) class of the AMP blocks
    (3, 777, 48, 48, 7, 3), (2, 1030, 48, 48, 11, 1), (3, 257, 24, 24, 3, 5), (1, 4100, 24, 24, 7, 1), (2, 3, 48, 48, 11, 5),
    # persistent grid of the tiled kernel (N % 64 / N % 96 tiles): more tiles than resident workgroups (each workgroup walks
    # 2-3 tiles with the next tile's rows prefetched under the epilogue), tile counts that are not a multiple of the grid,
    # a ragged last row tile; and the 4-wave side-by-side tile (N % 128) over several batch elements and column blocks
    (5, 20011, 96, 96, 7, 1), (3, 17003, 192, 192, 3, 5), (7, 1111, 96, 192, 7, 3), (5, 1301, 384, 256, 3, 1),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_gemm_conv_conv1d(nat, dtype, case):
    B, T, Cin, Cout, k, dil = case
    x = rnd(B, T, Cin, seed=40).to(dtype)  # channels-last
    w = (rnd(Cout, Cin, k, seed=41) * (1.0 / (Cin * k) ** 0.5)).to(dtype)
    bias = rnd(Cout, seed=42)
    pad = (k * dil - dil) // 2
    ref = F.conv1d(x.float().transpose(1, 2), w.float(), bias, dilation=dil, padding=pad).transpose(1, 2)
    wp = nat.pack_weight(w.permute(2, 1, 0).contiguous())  # [taps][Cin][Cout]
    y = torch.empty(B, T, Cout, dtype=dtype, device=DEV)
    nat.gemm_conv(dtype, B, T, T, Cin, Cout, wp, x, y, taps=k, off0=-pad, dil=dil, bias=bias)
    tol = {torch.float32: 2e-4, torch.float16: 6e-3, torch.bfloat16: 4e-2}[dtype]
    assert (y.float() - ref).abs().max().item() < tol
    # fused epilogue: y = y_prev + scale * (conv + bias + bias2[b] + resid)
    bias2 = rnd(B, Cout, seed=43)
    resid = rnd(B, T, Cout, seed=44).to(dtype)
    yprev = rnd(B, T, Cout, seed=45).to(dtype)
    y2 = yprev.clone()
    nat.gemm_conv(dtype, B, T, T, Cin, Cout, wp, x, y2, taps=k, off0=-pad, dil=dil, bias=bias, bias2=bias2, resid=resid,
                  accumulate=True, scale=1.0 / 3.0)
    ref2 = yprev.float() + (ref + bias2[:, None, :] + resid.float()) / 3.0
    assert (y2.float() - ref2).abs().max().item() < 2 * tol


RAGGED = [
    # (T, Cin, Cout, k, dil, lens): every kernel class of the vocoder -- side-by-side tile, persistent 64 / 96-column tiles
    # (several tiles per workgroup, skipped ones in between), narrow kernel, plain GEMM (k = 1)
    (300, 256, 128, 7, 3, [300, 1, 129, 0, 257]), (2500, 64, 192, 11, 1, [2500, 700, 1, 2499]),
    (9000, 96, 96, 7, 5, [17, 9000, 4097, 0, 5000, 129]), (1200, 48, 48, 7, 3, [1200, 255, 256, 257, 5]),
    (3000, 24, 24, 11, 5, [1, 3000, 64, 1000]), (700, 384, 256, 1, 1, [700, 130, 0, 129]),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", RAGGED)
def test_gemm_conv_ragged_batch_equals_single_runs(nat, dtype, case):
    """valid_rows: a batch of sequences of different lengths (padded to T) gives, on each element's valid rows, the bits
    that running that element alone at its own length gives -- input rows past the end are the convolution's zero padding --
    with and without the residual / accumulate epilogue; rows past the end are left alone where a whole tile is skipped."""
    T, Cin, Cout, k, dil, lens = case
    B = len(lens)
    x = rnd(B, T, Cin, seed=60).to(dtype)
    w = (rnd(Cout, Cin, k, seed=61) * (1.0 / (Cin * k) ** 0.5)).to(dtype)
    bias, resid = rnd(Cout, seed=62), rnd(B, T, Cout, seed=63).to(dtype)
    pad = (k * dil - dil) // 2
    wp = nat.pack_weight(w.permute(2, 1, 0).contiguous())
    vr = torch.tensor(lens, dtype=torch.int32, device=DEV)
    for epi in (False, True):
        y = torch.full((B, T, Cout), 7.0, dtype=dtype, device=DEV)
        kw = dict(resid=resid, accumulate=True, scale=0.5) if epi else {}
        nat.gemm_conv(dtype, B, T, T, Cin, Cout, wp, x, y, taps=k, off0=-pad, dil=dil, bias=bias, valid_rows=vr, **kw)
        for b, n in enumerate(lens):
            if n == 0:
                continue
            y1 = torch.full((1, n, Cout), 7.0, dtype=dtype, device=DEV)
            kw1 = dict(resid=resid[b:b + 1, :n].contiguous(), accumulate=True, scale=0.5) if epi else {}
            nat.gemm_conv(dtype, 1, n, n, Cin, Cout, wp, x[b:b + 1, :n].contiguous(), y1, taps=k, off0=-pad, dil=dil, bias=bias, **kw1)
            assert torch.equal(y[b, :n], y1[0]), (epi, b, n)
        if not epi:   # far past an element's end nothing was written
            for b, n in enumerate(lens):
                if n + 1024 + pad < T:
                    assert (y[b, n + 1024 + pad:] == 7.0).all(), b


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C", [24, 48, 96, 128])
def test_aa_snake_ragged_batch_equals_single_runs(nat, dtype, C):
    from indextts.BigVGAN.models import kaiser_sinc_filter
    f = kaiser_sinc_filter()
    lens = [1, 777, 0, 76, 77, 500, 13]
    B, T = len(lens), 777
    x = rnd(B, T, C, seed=70).to(dtype)
    al, be = rnd(C, seed=71) * 0.3, rnd(C, seed=72) * 0.3
    vr = torch.tensor(lens, dtype=torch.int32, device=DEV)
    y = nat.aa_snake(x, al, be, f, f, valid_rows=vr)
    for b, n in enumerate(lens):
        if n:
            y1 = nat.aa_snake(x[b:b + 1, :n].contiguous(), al, be, f, f)
            assert torch.equal(y[b, :n], y1[0]), (b, n)


UPS = [(1, 9, 1536, 768, 8, 4), (2, 33, 768, 384, 8, 4), (1, 100, 384, 192, 4, 4), (2, 130, 192, 96, 4, 4),
       (1, 500, 96, 48, 4, 2), (2, 700, 48, 24, 4, 2)]


def convtr_as_conv(w, u):
    """ConvTranspose1d weight [Cin,Cout,k] (stride u, padding (k-u)//2) -> ([taps][Cin][u*Cout], off0, y_shift).

    t' + pad = q*u + s:  y[t'] = x[q] W[:,:,s] + x[q-1] W[:,:,s+u]   (second term only when k = 2u)."""
    Cin, Cout, k = w.shape
    pad = (k - u) // 2
    if k == u:
        taps = w.permute(2, 0, 1).reshape(1, u, Cin, Cout).permute(0, 2, 1, 3).reshape(1, Cin, u * Cout)
        return taps.contiguous(), 0, -pad * Cout
    assert k == 2 * u
    lo = w[:, :, :u].permute(0, 2, 1).reshape(Cin, u * Cout)   # pairs with x[q]
    hi = w[:, :, u:].permute(0, 2, 1).reshape(Cin, u * Cout)   # pairs with x[q-1]
    return torch.stack([hi, lo], 0).contiguous(), -1, -pad * Cout


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("case", UPS)
def test_gemm_conv_transposed(nat, dtype, case):
    B, T, Cin, Cout, k, u = case
    x = rnd(B, T, Cin, seed=50).to(dtype)
    w = (rnd(Cin, Cout, k, seed=51) * (1.0 / (Cin * k / u) ** 0.5)).to(dtype)
    bias = rnd(Cout, seed=52)
    ref = F.conv_transpose1d(x.float().transpose(1, 2), w.float(), bias, stride=u, padding=(k - u) // 2).transpose(1, 2)
    Tout = T * u
    assert ref.shape[1] == Tout
    wt, off0, y_shift = convtr_as_conv(w, u)
    wp = nat.pack_weight(wt)
    y = torch.full((B, Tout, Cout), 7.0, dtype=dtype, device=DEV)
    rows = T + 1 if wt.shape[0] == 2 else T
    nat.gemm_conv(dtype, B, T, rows, Cin, u * Cout, wp, x, y, taps=wt.shape[0], off0=off0, dil=1,
                  bias=bias.repeat(u), y_bstride=Tout * Cout, y_shift=y_shift, y_limit=Tout * Cout)
    tol = 2e-4 if dtype == torch.float32 else 6e-3
    assert (y.float() - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_conv_plain_gemm_residual(nat, dtype):
    M, K, N = 300, 1280, 3840
    x = rnd(M, K, seed=60).to(dtype)
    w = (rnd(K, N, seed=61) * 0.03).to(dtype)
    bias = rnd(N, seed=62)
    wp = nat.pack_weight(w)
    ref = x.float() @ w.float() + bias
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    nat.gemm_conv(dtype, 1, M, M, K, N, wp, x, y, bias=bias, act=1)
    tol = 3e-4 if dtype == torch.float32 else 3e-2
    assert (y.float() - gelu_new(ref)).abs().max().item() < tol
    w2 = (rnd(K, K, seed=63) * 0.03).to(dtype)
    wp2 = nat.pack_weight(w2)
    h0 = rnd(M, K, seed=64)
    h = h0.clone()
    nat.gemm_conv(dtype, 1, M, M, K, K, wp2, x, h, bias=bias[:K].contiguous(), y_f32=True, resid=h)
    assert (h - (h0 + x.float() @ w2.float() + bias[:K])).abs().max().item() < 3e-4 * (1 if dtype == torch.float32 else 50)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_decode_reads_the_shared_first_keys_from_row_zero(nat, dtype):
    """kv_share = (p0 << 8) | C: the first C keys / values of every row are taken from cache row 0, positions [p0, p0 + C).
    With the rows' own copies holding the same bytes the output does not change by a bit (rows with different left paddings,
    a row whose context is shorter than C, a skipped row); with row 0's copy altered every other row's output changes
    (the copy really is what is read) while a zero word reads the rows' own copies again."""
    B, H, smax, pos, C = 6, 4, 320, 300, 32
    q = rnd(B, H * 64, seed=170).to(dtype)
    kc = rnd(B, H, smax, 64, seed=171).to(dtype)
    vc = rnd(B, H, smax, 64, seed=172).to(dtype)
    pads = [7, 0, 100, 255, 290, 40]                       # row 4: context 11 < C
    p0 = pads[0]
    for b in range(1, B):
        n = min(C, pos + 1 - pads[b])
        kc[b, :, pads[b]:pads[b] + n] = kc[0, :, p0:p0 + n]
        vc[b, :, pads[b]:pads[b] + n] = vc[0, :, p0:p0 + n]
    pad = torch.tensor(pads, dtype=torch.int32, device=DEV)
    posd = torch.tensor([pos], dtype=torch.int32, device=DEV)
    skip = torch.tensor([0, 0, 0, 0, 0, 1], dtype=torch.int32, device=DEV)
    word = torch.tensor([(p0 << 8) | C], dtype=torch.int32, device=DEV)
    zero = torch.zeros(1, dtype=torch.int32, device=DEV)
    outs = []
    for share in (None, zero, word):
        o = torch.full((B, H * 64), 7.0, dtype=dtype, device=DEV)
        nat.attn_decode(q, kc, vc, o, pad, posd, B, H, smax, skip_rows=skip, kv_share=share)
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert (outs[2][5] == 7.0).all() and torch.isfinite(outs[2].float()).all()
    kc2, vc2 = kc.clone(), vc.clone()
    kc2[0, :, p0:p0 + C] += 1.0                            # only row 0's copy changes
    o2 = torch.empty_like(outs[0])
    nat.attn_decode(q, kc2, vc2, o2, pad, posd, B, H, smax, kv_share=word)
    for b in range(1, 5):
        assert not torch.equal(o2[b], outs[0][b])
    o3 = torch.empty_like(outs[0])
    nat.attn_decode(q, kc2, vc2, o3, pad, posd, B, H, smax, kv_share=zero)
    assert torch.equal(o3[1:5], outs[0][1:5])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("smax,pos", [(200, 150), (900, 777), (64, 0)])
def test_attn_decode(nat, dtype, smax, pos):
    B, H = 5, 20
    q = rnd(B, H * 64, seed=70).to(dtype)
    kc = rnd(B, H, smax, 64, seed=71).to(dtype)
    vc = rnd(B, H, smax, 64, seed=72).to(dtype)
    pad = torch.tensor([0, 3, 17, 149, 150], dtype=torch.int32, device=DEV).clamp(max=pos)
    out = torch.empty(B, H * 64, dtype=dtype, device=DEV)
    nat.attn_decode(q, kc, vc, out, pad, torch.tensor([pos], dtype=torch.int32, device=DEV), B, H, smax)
    qf = q.float().view(B, H, 1, 64)
    sc = (qf @ kc.float().transpose(-1, -2)) / 8.0
    j = torch.arange(smax, device=DEV)[None, None, None, :]
    vis = (j <= pos) & (j >= pad[:, None, None, None])
    sc = sc.masked_fill(~vis, float("-inf"))
    ref = (torch.softmax(sc, -1) @ vc.float()).view(B, H * 64)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (out.float() - ref).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("S", [47, 130])
def test_attn_prefill(nat, dtype, S):
    B, H, smax = 3, 20, 160
    D = H * 64
    qkv = rnd(B, S, 3 * D, seed=80).to(dtype)
    pad = torch.tensor([0, 5, 40], dtype=torch.int32, device=DEV)
    out = torch.empty(B, S, D, dtype=dtype, device=DEV)
    kc = torch.zeros(B, H, smax, 64, dtype=dtype, device=DEV)
    vc = torch.zeros(B, H, smax, 64, dtype=dtype, device=DEV)
    nat.attn_prefill(qkv, out, kc, vc, pad, B, S, H, smax)
    q, k, v = (t.float().view(B, S, H, 64).transpose(1, 2) for t in qkv.split(D, dim=-1))
    sc = (q @ k.transpose(-1, -2)) / 8.0
    i = torch.arange(S, device=DEV)
    vis = (i[None, :] <= i[:, None])[None, None] & (i[None, None, None, :] >= pad[:, None, None, None])
    sc = sc.masked_fill(~vis, float("-inf"))
    anyv = vis.any(-1, keepdim=True)
    pr = torch.where(anyv, torch.softmax(sc.masked_fill(~anyv, 0.0), -1), torch.zeros_like(sc))
    ref = (pr @ v).transpose(1, 2).reshape(B, S, D)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (out.float() - ref).abs().max().item() < tol
    assert torch.equal(kc[:, :, :S].float(), k)
    assert torch.equal(vc[:, :, :S].float(), v)
    out2 = torch.empty_like(out)
    nat.attn_prefill(qkv, out2, None, None, pad, B, S, H, smax)
    assert torch.equal(out, out2)


def test_sample_matches_hf_fixture(nat):
    from oracle import sampling_ref
    g = np.load(os.path.join(G, "sampling.npz"))
    logits = torch.from_numpy(g["logits"]).to(DEV)
    B, V = logits.shape
    gen = g["history"][:, 31:]  # generated part; the fake prefix contributes ids {1, 8192}
    cap = 64
    history = torch.zeros(B, cap, dtype=torch.int32, device=DEV)
    history[:, :gen.shape[1]] = torch.from_numpy(gen).to(torch.int32).to(DEV)
    extra = torch.tensor([1, 8192], dtype=torch.int32, device=DEV)
    for do_sample in (True, False):
        tokens = torch.zeros(B, dtype=torch.int32, device=DEV)
        finished = torch.zeros(B, dtype=torch.int32, device=DEV)
        state = torch.zeros(8, dtype=torch.int32, device=DEV)
        state[0] = gen.shape[1]
        state[1] = 100
        dbg = torch.empty(B, V, dtype=torch.float32, device=DEV)
        nat.sample(logits, tokens, history.clone(), finished, state, extra, None, 10.0, 0.8, 30, 0.8, do_sample, 1234,
                   8193, dbg)
        torch.cuda.synchronize()
        st = state.cpu().numpy()
        assert st[0] == gen.shape[1] + 1 and st[1] == 101 and st[3] == 0
        d = dbg.cpu().numpy()
        if do_sample:
            exp = g["after_topp"]
            assert np.array_equal(np.isfinite(d), np.isfinite(exp))
            np.testing.assert_allclose(d[np.isfinite(d)], exp[np.isfinite(exp)], rtol=1e-6)
            for b in range(B):
                u = sampling_ref.uniform01(1234, b, gen.shape[1])
                assert int(tokens[b]) == sampling_ref.pick(exp[b], u)
        else:
            np.testing.assert_allclose(d, g["after_penalty"], rtol=1e-6)
            assert np.array_equal(tokens.cpu().numpy(), np.argmax(g["after_penalty"], -1))


def test_sample_finish_and_force(nat):
    B, V = 3, 8194
    logits = torch.zeros(B, V, device=DEV)
    logits[0, 8193] = 50.0  # row 0 emits the stop token
    logits[1, 77] = 50.0
    logits[2, 99] = 50.0
    tokens = torch.zeros(B, dtype=torch.int32, device=DEV)
    history = torch.zeros(B, 16, dtype=torch.int32, device=DEV)
    finished = torch.zeros(B, dtype=torch.int32, device=DEV)
    state = torch.zeros(8, dtype=torch.int32, device=DEV)
    force = torch.tensor([-1, -1, 1], dtype=torch.int32, device=DEV)
    for step in range(3):
        nat.sample(logits, tokens, history, finished, state, None, force, 1.0, 1.0, 30, 0.8, False, 0, 8193)
    torch.cuda.synchronize()
    assert history[:, :3].cpu().tolist() == [[8193, 8193, 8193], [77, 77, 77], [99, 8193, 8193]]
    assert finished.cpu().tolist() == [1, 0, 1]
    assert state.cpu().tolist()[:4] == [3, 3, 2, 0]
    # no_advance: the kernel leaves step counter / cache position to the caller, who advances them in the next step's
    # first LayerNorm launch (itts_ln_reduce state_bump) -- same histories and flags
    tokens2, history2, finished2 = torch.zeros_like(tokens), torch.zeros_like(history), torch.zeros_like(finished)
    state2 = torch.zeros(8, dtype=torch.int32, device=DEV)
    h = torch.zeros(B, 1280, device=DEV)
    xn = torch.zeros(B, 1280, dtype=torch.bfloat16, device=DEV)
    w, b = torch.ones(1280, device=DEV), torch.zeros(1280, device=DEV)
    for step in range(3):
        nat.sample(logits, tokens2, history2, finished2, state2, None, force, 1.0, 1.0, 30, 0.8, False, 0, 8193, no_advance=True)
        assert state2.cpu().tolist()[:2] == [step, step]
        nat.ln_reduce(h, w, b, xn, state_bump=state2[0:2])
    torch.cuda.synchronize()
    assert torch.equal(history2, history) and torch.equal(finished2, finished) and torch.equal(tokens2, tokens)
    assert state2.cpu().tolist()[:4] == [3, 3, 2, 0]


def test_tanh_pcm(nat):
    x = rnd(3, 1000, seed=90, scale=2.0)
    wav = torch.empty_like(x)
    pcm = torch.empty(3, 1000, dtype=torch.int16, device=DEV)
    nat.tanh_pcm(x, wav, pcm)
    ref = torch.tanh(x)
    assert (wav - ref).abs().max().item() < 2e-6
    exp = torch.clamp(32767 * wav, -32767.0, 32767.0).cpu().numpy().astype(np.int16)
    assert np.array_equal(pcm.cpu().numpy(), exp)


@pytest.mark.parametrize("do_sample,lp", [(True, 0.0), (False, 0.0), (True, 1.0)])
def test_beam_step_matches_oracle(nat, do_sample, lp):
    """itts_beam_step over several steps (EOS becomes likely half-way) against oracle/beam_ref.py: same tokens, same
    source rows, same done flags at every step, same best hypotheses at the end."""
    from oracle import beam_ref
    B, nb, V, cap, steps = 3, 3, 8194, 64, 14
    R = B * nb
    sp = dict(do_sample=do_sample, top_k=30, top_p=0.8, temperature=1.0 if not do_sample else 0.9, repetition_penalty=10.0)
    rng = np.random.default_rng(11)
    ref = beam_ref.BeamSearch(B, nb, sp, [1] * 5 + [8192], eos=8193, length_penalty=lp, seed=77)
    i32 = dict(dtype=torch.int32, device=DEV)
    tokens, src = torch.zeros(R, **i32), torch.zeros(R, **i32)
    scores = torch.zeros(R, device=DEV)
    scores.view(B, nb)[:, 1:] = -1e9
    hist = torch.zeros(2, R, cap, **i32)
    hyp_score, hyp_len = torch.zeros(B, nb, device=DEV), torch.zeros(B, nb, **i32)
    hyp_tok, n_hyp = torch.zeros(B, nb, cap, **i32), torch.zeros(B, **i32)
    worst, done, state = torch.full((B,), 1e9, device=DEV), torch.zeros(B, **i32), torch.zeros(8, **i32)
    state[1] = 40
    extra = torch.tensor([1, 8192], **i32)
    for k in range(steps):
        lg = (rng.normal(size=(R, V)) * 2.5).astype(np.float32)
        if k >= 5:
            lg[:, 8193] += 6.0 + k   # EOS enters the candidate set
        t_ref, s_ref = ref.step(lg)
        nat.beam_step(torch.from_numpy(lg).to(DEV), nb, tokens, src, scores, hist, hyp_score, hyp_len, hyp_tok, n_hyp, worst, done,
                      state, extra, sp["repetition_penalty"], sp["temperature"], sp["top_k"], sp["top_p"], do_sample, lp, 77, 8193)
        assert tokens.cpu().tolist() == t_ref.tolist(), f"tokens at step {k}"
        assert src.cpu().tolist() == s_ref.tolist(), f"source rows at step {k}"
        assert [bool(x) for x in done.cpu().tolist()] == ref.done, f"done flags at step {k}"
        assert int(state[0]) == k + 1 and int(state[1]) == 41 + k
        live = [b for b in range(B) if not ref.done[b]]
        got = scores.cpu().numpy().reshape(B, nb)
        assert np.allclose(got[live], ref.scores[live], rtol=1e-5, atol=1e-4)
        h = hist[(k + 1) & 1].cpu().numpy().reshape(B, nb, cap)
        for b in live:
            for j in range(nb):
                assert h[b, j, : k + 1].tolist() == ref.hist[b][j][6:], (k, b, j)
        if ref.all_done():
            break
    assert ref.all_done() or k == steps - 1
    for b in range(B):
        want = sorted((float(s), t) for s, t in ref.hyps[b].beams)
        got = sorted((float(hyp_score[b, i]), hyp_tok[b, i, : int(hyp_len[b, i])].cpu().tolist()) for i in range(int(n_hyp[b])))
        assert len(want) == len(got)
        for (ws, wt), (gs, gt) in zip(want, got):
            assert abs(ws - gs) < 1e-3 and wt == gt


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_beam_reorder_kv(nat, dtype):
    L, B, nb, H, smax, ctx = 3, 4, 3, 2, 64, 37
    R = B * nb
    kc = rnd(L, R, H, smax, 64, seed=91).to(dtype)
    vc = rnd(L, R, H, smax, 64, seed=92).to(dtype)
    src = torch.tensor([2, 0, 0, 3, 4, 5, 7, 8, 6, 10, 10, 9], dtype=torch.int32, device=DEV)  # element 1 maps to itself
    state = torch.zeros(8, dtype=torch.int32, device=DEV)
    state[1] = ctx
    k0, v0 = kc.clone(), vc.clone()
    nat.beam_reorder_kv(kc, vc, src, state, B, nb)
    idx = src.long()
    assert torch.equal(kc[:, :, :, :ctx], k0[:, idx][:, :, :, :ctx]) and torch.equal(vc[:, :, :, :ctx], v0[:, idx][:, :, :, :ctx])
    assert torch.equal(kc[:, :, :, ctx:], k0[:, :, :, ctx:]) and torch.equal(vc[:, :, :, ctx:], v0[:, :, :, ctx:])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,N,K", [(33, 3840, 1280), (64, 5120, 1280), (96, 1280, 5120), (77, 8194, 1280), (96, 3840, 1280)])
def test_gemm_skinny_many_rows_one_weight_pass(nat, dtype, M, N, K):
    """33..96 rows (4 / 6 row tiles) in ONE launch: same results as the 32-row launches row block by row block, bit for
    bit (a row's dot products do not depend on its neighbours), and close to the fp32 reference."""
    x = rnd(M, K, seed=130).to(dtype)
    w = (rnd(K, N, seed=131) * 0.05).to(dtype)
    bias = rnd(N, seed=132)
    wp = nat.pack_weight(w)
    y = torch.empty(M, N, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, bias, x=x, epi=nat.EPI_GELU_STORE, y=y)
    ref = gelu_new(x.float() @ w.float() + bias)
    assert (y.float() - ref).abs().max().item() < 2e-2 * max(1.0, ref.abs().max().item())
    for r0 in range(0, M, 32):
        m = min(32, M - r0)
        yb = torch.empty(m, N, dtype=dtype, device=DEV)
        nat.gemm_skinny(dtype, m, N, K, wp, bias, x=x[r0:r0 + m].contiguous(), epi=nat.EPI_GELU_STORE, y=yb)
        assert torch.equal(y[r0:r0 + m], yb), r0
    plan = nat.skinny_plan(dtype, M, N, K)
    assert plan["grid"][0] * plan["grid"][1] <= 256


@pytest.mark.parametrize("dtype,M", [(torch.bfloat16, 32), (torch.bfloat16, 13), (torch.bfloat16, 96), (torch.float32, 32), (torch.float16, 40)])
def test_packed_activation_layout_end_to_end(nat, dtype, M):
    """The packed activation layout (include/indextts_hip.h) through its producers and its consumer: itts_ln_reduce
    (y_packed), itts_attn_decode (out_packed), itts_gemm_skinny (x_packed, y_packed with gelu) give the SAME
    BITS as the row-major forms, and the packing matches the documented index formula (nat.pack_activation)."""
    D, H = 1280, 20
    # the layout itself: element (m, k) where the header says
    E = 4 if dtype == torch.float32 else 8
    KS, mtp = 4 * E, (M + 15) // 16
    x = rnd(M, D, seed=200).to(dtype)
    xp = nat.pack_activation(x)
    assert xp.numel() == mtp * 16 * D
    for (m, k) in ((0, 0), (M - 1, D - 1), (M // 2, 333), (min(17, M - 1), 64)):
        off = ((k // KS * mtp + m // 16) * 64 + (k % KS) // E * 16 + m % 16) * E + k % E
        assert xp[off].item() == x[m, k].item()
    assert torch.equal(nat.unpack_activation(xp, M, D), x)
    # LayerNorm producer
    h = rnd(M, D, seed=201, scale=2.0)
    lw, lb = 1.0 + 0.1 * rnd(D, seed=202), 0.1 * rnd(D, seed=203)
    y_rm = torch.empty(M, D, dtype=dtype, device=DEV)
    y_pk = torch.zeros(mtp * 16 * D, dtype=dtype, device=DEV)
    nat.ln_reduce(h.clone(), lw, lb, y_rm)
    nat.ln_reduce(h.clone(), lw, lb, y_pk, y_packed=True)
    assert torch.equal(nat.unpack_activation(y_pk, M, D), y_rm)
    # GEMM consumer + gelu producer
    w = (rnd(D, 4 * D, seed=204) * 0.03).to(dtype)
    wp = nat.pack_weight(w)
    bias = rnd(4 * D, seed=205)
    f_rm = torch.empty(M, 4 * D, dtype=dtype, device=DEV)
    f_pk = torch.zeros(mtp * 16 * 4 * D, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, 4 * D, D, wp, bias, x=y_rm, epi=nat.EPI_GELU_STORE, y=f_rm)
    nat.gemm_skinny(dtype, M, 4 * D, D, wp, bias, x=y_pk, epi=nat.EPI_GELU_STORE, y=f_pk, x_packed=True, y_packed=True)
    assert torch.equal(nat.unpack_activation(f_pk, M, 4 * D), f_rm)
    # split-K slabs from a packed operand
    w2 = (rnd(4 * D, D, seed=206) * 0.03).to(dtype)
    wp2 = nat.pack_weight(w2)
    KSP = 3
    slab_rm, slab_pk = torch.zeros(KSP, M, D, device=DEV), torch.zeros(KSP, M, D, device=DEV)
    nat.gemm_skinny(dtype, M, D, 4 * D, wp2, None, x=f_rm, epi=nat.EPI_SLAB_F32, yf=slab_rm, ksplit=KSP)
    nat.gemm_skinny(dtype, M, D, 4 * D, wp2, None, x=f_pk, epi=nat.EPI_SLAB_F32, yf=slab_pk, ksplit=KSP, x_packed=True)
    assert torch.equal(slab_pk, slab_rm)
    # decode attention producer
    smax, ctx = 64, 41
    q = rnd(M, D, seed=207).to(dtype)
    kc, vc = rnd(M, H, smax, 64, seed=208).to(dtype), rnd(M, H, smax, 64, seed=209).to(dtype)
    pad = torch.zeros(M, dtype=torch.int32, device=DEV)
    pad[M // 2] = 7
    pos = torch.tensor([ctx - 1], dtype=torch.int32, device=DEV)
    a_rm = torch.empty(M, D, dtype=dtype, device=DEV)
    a_pk = torch.zeros(mtp * 16 * D, dtype=dtype, device=DEV)
    nat.attn_decode(q, kc, vc, a_rm, pad, pos, M, H, smax)
    nat.attn_decode(q, kc, vc, a_pk, pad, pos, M, H, smax, out_packed=True)
    assert torch.equal(nat.unpack_activation(a_pk, M, D), a_rm)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("r", [4, 16, 24])
def test_runtime_lora_fused_into_output_projection(nat, dtype, r):
    """y = x W + (x A^T) B^T s  with A^T s carried as extra output columns of the split-K projection and B^T applied by the
    [residual-reduce + LayerNorm] launch (itts_ln_reduce lora_b): equal to the same pipeline on the MERGED weight W + A^T B^T s
    up to rounding of the merged weight to T, and to the fp32 formula."""
    M, K, D, KSP, s_ = 32, 5120, 1280, 3, 2.0
    x = rnd(M, K, seed=300).to(dtype)
    W = rnd(K, D, seed=301) * 0.03
    A = rnd(r, K, seed=302) * 0.05          # peft lora_A.weight [r, in]
    Bm = rnd(D, r, seed=303) * 0.05         # peft lora_B.weight [out, r]
    bias = rnd(D, seed=304)
    lw, lb = 1.0 + 0.1 * rnd(D, seed=305), 0.1 * rnd(D, seed=306)
    h0 = rnd(M, D, seed=307, scale=2.0)
    rp = (r + 15) // 16 * 16
    ext = torch.zeros(K, D + rp, device=DEV)
    ext[:, :D] = W
    ext[:, D:D + r] = A.t() * s_
    wp_ext = nat.pack_weight(ext.to(dtype))
    slab = torch.zeros(KSP, M, D + rp, device=DEV)
    nat.gemm_skinny(dtype, M, D + rp, K, wp_ext, None, x=x, epi=nat.EPI_SLAB_F32, yf=slab, ksplit=KSP)
    h, y = h0.clone(), torch.empty(M, D, dtype=dtype, device=DEV)
    nat.ln_reduce(h, lw, lb, y, slab=slab, nslab=KSP, bias=bias, slab_stride=D + rp, lora_b=Bm.t().contiguous())
    xf = x.float()
    xa = xf @ (A.t() * s_).to(dtype).float()
    h_ref = h0 + bias + xf @ W.to(dtype).float() + xa @ Bm.t()
    tol = 2e-3 if dtype == torch.float32 else 3e-2
    assert (h - h_ref).abs().max().item() < tol
    assert (y.float() - F.layer_norm(h_ref, (D,), lw, lb, 1e-5)).abs().max().item() < (2e-3 if dtype == torch.float32 else 5e-2)
    # the merged-weight pipeline gives the same residual stream within the rounding of the merged weight
    wp_m = nat.pack_weight((W + (A.t() @ Bm.t()) * s_).to(dtype))
    slab_m = torch.zeros(KSP, M, D, device=DEV)
    nat.gemm_skinny(dtype, M, D, K, wp_m, None, x=x, epi=nat.EPI_SLAB_F32, yf=slab_m, ksplit=KSP)
    h_m, y_m = h0.clone(), torch.empty(M, D, dtype=dtype, device=DEV)
    nat.ln_reduce(h_m, lw, lb, y_m, slab=slab_m, nslab=KSP, bias=bias)
    assert (h - h_m).abs().max().item() < (2e-3 if dtype == torch.float32 else 0.25)
    # and without an adapter the extra arguments change nothing
    h_p, y_p = h0.clone(), torch.empty(M, D, dtype=dtype, device=DEV)
    nat.ln_reduce(h_p, lw, lb, y_p, slab=slab_m, nslab=KSP, bias=bias, slab_stride=D)
    assert torch.equal(h_p, h_m) and torch.equal(y_p, y_m)


def test_gemm_skinny_qkv_cache_epilogue_many_rows(nat):
    """QKV epilogue at 96 rows (batch 32 x 3 beams): q rows, and k/v scattered into the cache at *pos, per head."""
    dtype, M, H, smax, pos = torch.bfloat16, 96, 20, 64, 17
    D = H * 64
    x = rnd(M, D, seed=140).to(dtype)
    w = (rnd(D, 3 * D, seed=141) * 0.03).to(dtype)
    bias = rnd(3 * D, seed=142)
    wp = nat.pack_weight(w)
    q = torch.empty(M, D, dtype=dtype, device=DEV)
    kc = torch.zeros(M, H, smax, 64, dtype=dtype, device=DEV)
    vc = torch.zeros_like(kc)
    posd = torch.tensor([pos], dtype=torch.int32, device=DEV)
    nat.gemm_skinny(dtype, M, 3 * D, D, wp, bias, x=x, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc, vcache=vc, pos=posd, heads=H, smax=smax)
    ref = (x.float() @ w.float() + bias)
    tol = 2e-2 * max(1.0, ref.abs().max().item())
    assert (q.float() - ref[:, :D]).abs().max().item() < tol
    assert (kc[:, :, pos].reshape(M, D).float() - ref[:, D:2 * D]).abs().max().item() < tol
    assert (vc[:, :, pos].reshape(M, D).float() - ref[:, 2 * D:]).abs().max().item() < tol
    kc[:, :, pos] = 0
    vc[:, :, pos] = 0
    assert kc.abs().max().item() == 0 and vc.abs().max().item() == 0


@pytest.mark.parametrize("dtype,M", [(torch.bfloat16, 32), (torch.bfloat16, 96), (torch.bfloat16, 5), (torch.float16, 48)])
@pytest.mark.parametrize("N,epi", [(3840, "qkv"), (5120, "gelu"), (256, "store")])
def test_gemm_skinny_layernorm_folded_into_the_gemm(nat, dtype, M, N, epi):
    """LN(h; gamma, beta) W + b with the LayerNorm folded into the GEMM (itts_skinny_args.ln_c): the kernel multiplies the RAW
    rows by pack(gamma . W), takes mean / rstd of the rows from its own operand fragments (matrix pipe) and applies
    rstd (acc - mean c) + d in the epilogue.  Held against (a) the exact fp32 formula on the T-rounded rows -- what the kernel
    computes, up to fp32 summation order -- and (b) the two-launch form it replaces (LayerNorm launch -> plain GEMM), whose
    only difference is WHERE the rounding to T happens.  Rows with a large common offset (|mean| = 8 sigma) included: the
    cancellation in acc - mean c is the risk of this form."""
    D, H = 1280, 20
    if epi == "qkv":
        N = 3 * D
    h = rnd(M, D, seed=400, scale=1.5)
    h[M // 2] += 12.0                                   # |mean| >> sigma on one row
    h[0, 7] = 60.0                                      # an outlier feature
    gamma, beta = 1.0 + 0.2 * rnd(D, seed=401), 0.2 * rnd(D, seed=402)
    W = rnd(D, N, seed=403) * 0.03
    b = rnd(N, seed=404)
    hb = h.to(dtype)
    Wr = (gamma[:, None] * W).to(dtype)
    c = Wr.double().sum(0).float().contiguous()
    d = (beta.double() @ W.double() + b.double()).float().contiguous()
    wp = nat.pack_weight(Wr.contiguous())
    x_pk = nat.pack_activation(hb)
    # exact formula on the rounded operands
    hd = hb.double()
    mean, var = hd.mean(1, keepdim=True), hd.var(1, unbiased=False, keepdim=True)
    ref = ((hd - mean) / torch.sqrt(var + 1e-5)) @ Wr.double() + d.double()
    if epi == "gelu":
        ref = F.gelu(ref, approximate="tanh")
    ref = ref.float()
    mtp = (M + 15) // 16
    if epi == "qkv":
        smax, pos = 64, 9
        q = torch.empty(M, D, dtype=dtype, device=DEV)
        kc = torch.zeros(M, H, smax, 64, dtype=dtype, device=DEV)
        vc = torch.zeros_like(kc)
        posd = torch.tensor([pos], dtype=torch.int32, device=DEV)
        word = torch.zeros(1, dtype=torch.int32, device=DEV)
        nat.gemm_skinny(dtype, M, N, D, wp, d, x=x_pk, epi=nat.EPI_QKV_CACHE, y=q, kcache=kc, vcache=vc, pos=posd, heads=H, smax=smax,
                        x_packed=True, ln_c=c, bump=word)
        assert word.item() == 1                          # the launch advances the word it was given, once
        got = torch.cat([q.float(), kc[:, :, pos].reshape(M, D).float(), vc[:, :, pos].reshape(M, D).float()], 1)
    else:
        y = torch.zeros(mtp * 16 * N, dtype=dtype, device=DEV)
        nat.gemm_skinny(dtype, M, N, D, wp, d, x=x_pk, epi=nat.EPI_GELU_STORE if epi == "gelu" else nat.EPI_STORE, y=y,
                        x_packed=True, y_packed=True, ln_c=c)
        got = nat.unpack_activation(y, M, N).float()
        # the row-major operand gives the same bits
        y2 = torch.empty(M, N, dtype=dtype, device=DEV)
        nat.gemm_skinny(dtype, M, N, D, wp, d, x=hb, epi=nat.EPI_GELU_STORE if epi == "gelu" else nat.EPI_STORE, y=y2, ln_c=c)
        assert torch.equal(y2, nat.unpack_activation(y, M, N))
    # (a) the kernel's own arithmetic: only the output rounding to T (2^-9 relative) and fp32 summation order separate them
    err = (got - ref).abs()
    assert (err / (ref.abs() + 1.0)).max().item() < 6e-3, (err.max().item(), ref.abs().max().item())
    # (b) the form it replaces: LayerNorm launch (rounds LN(h) to T) -> GEMM on W (rounded to T)
    xn = torch.empty(M, D, dtype=dtype, device=DEV)
    nat.ln_reduce(h.clone(), gamma, beta, xn)
    old = xn.float() @ W.to(dtype).float() + b
    if epi == "gelu":
        old = F.gelu(old, approximate="tanh")
    ref_true = F.layer_norm(h, (D,), gamma, beta, 1e-5) @ W + b
    if epi == "gelu":
        ref_true = F.gelu(ref_true, approximate="tanh")
    e_new = (got - ref_true).pow(2).mean().sqrt().item()
    e_old = (old - ref_true).pow(2).mean().sqrt().item()
    assert e_new < 2.5 * e_old + 1e-3, (e_new, e_old)    # same accuracy class as the two-launch form


@pytest.mark.parametrize("dtype,M", [(torch.bfloat16, 32), (torch.bfloat16, 96), (torch.bfloat16, 21), (torch.float32, 16), (torch.float16, 40)])
@pytest.mark.parametrize("K", [1280, 5120])
@pytest.mark.parametrize("rows_per_wg,wide", [(0, False), (16, False), (16, True), (32, True)])
def test_gemm_skinny_residual_epilogue_with_packed_copy(nat, dtype, M, K, rows_per_wg, wide):
    """h += x W + b in the GEMM's epilogue (ITTS_EPI_RESID_F32, no split-K: one owner per element) with a T-typed packed copy of
    the new rows beside it -- the producer side of the LayerNorm-folded decode block -- in every launch geometry (rows dealt
    to grid.z, 16-wave workgroups).  The fp32 result equals the split-K + itts_ln_reduce form up to summation order."""
    N = 1280
    x = rnd(M, K, seed=410).to(dtype)
    w = (rnd(K, N, seed=411) * 0.03).to(dtype)
    wp = nat.pack_weight(w)
    b = rnd(N, seed=412)
    h0 = rnd(M, N, seed=413, scale=2.0)
    mtp = (M + 15) // 16
    h = h0.clone()
    hb = torch.zeros(mtp * 16 * N, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, b, x=nat.pack_activation(x), epi=nat.EPI_RESID_F32, yf=h, y=hb, x_packed=True, y_packed=True,
                    rows_per_wg=rows_per_wg, wide_wg=wide)
    ref = h0.double() + x.double() @ w.double() + b.double()
    assert (h.double() - ref).abs().max().item() < (2e-4 if dtype == torch.float32 else 2e-3)
    assert torch.equal(nat.unpack_activation(hb, M, N), h.to(dtype))          # the copy is the rounded new row
    if M % 16:
        assert nat.unpack_activation(hb, mtp * 16, N)[M:].float().abs().max().item() == 0   # padding rows untouched
    # row-major copy, no copy
    h2, hb2 = h0.clone(), torch.empty(M, N, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, M, N, K, wp, b, x=x, epi=nat.EPI_RESID_F32, yf=h2, y=hb2, rows_per_wg=rows_per_wg, wide_wg=wide)
    assert torch.equal(h2, h) and torch.equal(hb2, h.to(dtype))
    h3 = h0.clone()
    nat.gemm_skinny(dtype, M, N, K, wp, b, x=x, epi=nat.EPI_RESID_F32, yf=h3, rows_per_wg=rows_per_wg, wide_wg=wide)
    assert torch.equal(h3, h)
    plan = nat.skinny_plan(dtype, M, N, K, 1, rows_per_wg, wide)
    assert plan["grid"][2] * plan["row_tiles_per_wg"] * 16 >= min(M, 16 if dtype == torch.float32 else 96) and plan["waves"] in (8, 16)


@pytest.mark.parametrize("M,K", [(2016, 1280), (2016, 5120), (300, 5120), (1000, 1280)])
@pytest.mark.parametrize("ks", [2, 3])
def test_gemm_conv_split_k_slabs_and_ln_reduce(nat, M, K, ks):
    """Plain GEMM with split-K inside one launch (itts_conv_args.ksplit: the K slices run as separate tiles, partial products into
    fp32 slabs) followed by itts_ln_reduce (residual + bias + slabs in order, then LayerNorm): the prefill's out-projection / FC2
    when their 128 x 128 output tiles alone do not fill the chip.  Held against the one-launch GEMM with the residual epilogue +
    itts_layernorm (same products, another summation order: fp32 rounding apart) and the fp64 formula."""
    dtype, N = torch.bfloat16, 1280
    x = rnd(M, K, seed=440).to(dtype)
    w = (rnd(K, N, seed=441) * 0.03).to(dtype)
    wp = nat.pack_weight(w)
    b = rnd(N, seed=442)
    lw, lb = 1.0 + 0.1 * rnd(N, seed=443), 0.1 * rnd(N, seed=444)
    h0 = rnd(M, N, seed=445, scale=2.0)
    slab = torch.zeros(ks, M, N, device=DEV)
    nat.gemm_conv(dtype, 1, M, M, K, N, wp, x, slab, y_f32=True, ksplit=ks)
    KT = K // 32
    for s_ in range(ks):                                       # every slab holds its slice's partial products
        k0, k1 = (s_ * KT) // ks * 32, ((s_ + 1) * KT) // ks * 32
        ref_s = x[:, k0:k1].double() @ w[k0:k1].double()
        assert (slab[s_].double() - ref_s).abs().max().item() < 2e-3 * max(1.0, ref_s.abs().max().item()), s_
    h, xn = h0.clone(), torch.empty(M, N, dtype=dtype, device=DEV)
    nat.ln_reduce(h, lw, lb, xn, slab=slab, nslab=ks, bias=b)
    h1 = h0.clone()
    nat.gemm_conv(dtype, 1, M, M, K, N, wp, x, h1, bias=b, y_f32=True, resid=h1)
    xn1 = torch.empty(M, N, dtype=dtype, device=DEV)
    nat.layernorm(h1, lw, lb, xn1)
    assert (h - h1).abs().max().item() < 1e-3
    assert (xn.float() - xn1.float()).abs().max().item() < 4e-2
    ref = h0.double() + b.double() + x.double() @ w.double()
    assert (h.double() - ref).abs().max().item() < 2e-3
    with pytest.raises(nat.NativeError):                       # slabs carry no bias / residual: refused
        nat.gemm_conv(dtype, 1, M, M, K, N, wp, x, slab, y_f32=True, ksplit=ks, bias=b)


def test_gemm_conv_plain_wide_tiles_for_few_rows(nat):
    """N = 5120 at the prefill's ~2 000 rows: 128 x 160 output tiles (512 of them: one round of the chip) instead of 128 x 128 (640:
    two rounds) -- same products per output element as at a row count that takes the 128 x 128 kernel."""
    dtype, K, N = torch.bfloat16, 1280, 5120
    w = (rnd(K, N, seed=451) * 0.03).to(dtype)
    wp = nat.pack_weight(w)
    b = rnd(N, seed=452)
    x = rnd(4000, K, seed=450).to(dtype)
    big = torch.empty(4000, N, dtype=dtype, device=DEV)
    nat.gemm_conv(dtype, 1, 4000, 4000, K, N, wp, x, big, bias=b, act=1)          # 32 x 40 tiles: the 128 x 128 kernel
    for M in (2016, 1999, 2048):
        y = torch.empty(M, N, dtype=dtype, device=DEV)
        nat.gemm_conv(dtype, 1, M, M, K, N, wp, x[:M].contiguous(), y, bias=b, act=1)    # 16 x 32 tiles of 128 x 160
        assert torch.equal(y, big[:M]), M
    ref = gelu_new(x[:64].float() @ w.float() + b)
    assert (big[:64].float() - ref).abs().max().item() < 3e-2 * max(1.0, ref.abs().max().item())


def test_embed_step_packed_copy_bump_and_position_clamp(nat):
    """itts_embed_step: fp32 rows + the T-typed packed copy, the word it advances, per-row clocks, and the clamp of the
    position index (a finished slot that keeps stepping must not read past the table)."""
    B, D, V, P = 21, 1280, 50, 12
    table, ptab = rnd(V, D, seed=420), rnd(P, D, seed=421)
    tokens = torch.arange(B, dtype=torch.int32, device=DEV) % V
    step = torch.tensor([5], dtype=torch.int32, device=DEV)
    word = torch.tensor([40], dtype=torch.int32, device=DEV)
    step0 = torch.zeros(B, dtype=torch.int32, device=DEV)
    step0[3] = 4
    step0[4] = -100                                       # position 107 -> clamped to the last row
    step0[5] = 50                                         # negative position -> clamped to row 0
    for dtype in (torch.bfloat16, torch.float32):
        h = torch.zeros(B, D, device=DEV)
        hp = torch.zeros(nat.packed_rows(B) * D, dtype=dtype, device=DEV)
        nat.embed_step(tokens, table, ptab, step, 2, h, bump=word, row_step0=step0, h_packed=hp)
        pos = (5 - step0 + 2).clamp(0, P - 1).long()
        ref = table[tokens.long()] + ptab[pos]
        assert torch.equal(h, ref)
        assert torch.equal(nat.unpack_activation(hp, B, D), ref.to(dtype))
    assert word.item() == 42 and step.item() == 5


@pytest.mark.parametrize("dtype,bs", [(torch.bfloat16, 16), (torch.bfloat16, 64), (torch.float32, 32), (torch.float16, 16)])
def test_paged_kv_kernels_equal_the_contiguous_cache(nat, dtype, bs):
    """The paged cache through its three kernel families against contiguous cache rows holding the same keys / values:
    QKV epilogue (append at *pos through the block table), decode attention (block ids out of a register, also for the shared
    first keys read from row 0), packed prefill (prompt rows in) and the cached-prefix read of the latent pass -- identical
    bits.  The position window sits ACROSS the table's ring boundary (positions around 64 * bs) and the blocks are dealt in a
    shuffled order; every unmapped table entry points at the scratch block 0."""
    torch.manual_seed(5)
    B, H, D = 5, 4, 256
    TAB = 64
    base = TAB * bs - 37                                  # windows straddle the ring's wrap-around
    pads = [base + v for v in (0, 9, 21, 3, 30)]
    ctx_end = base + 75                                   # current write position
    smax = ctx_end + 8
    kc_r = torch.zeros(B, H, smax, 64, dtype=dtype, device=DEV)
    vc_r = torch.zeros_like(kc_r)
    nblk = 1 + B * ((ctx_end + 8 - base) // bs + 2)
    pool_k = torch.zeros(nblk, H, bs, 64, dtype=dtype, device=DEV)
    pool_v = torch.zeros_like(pool_k)
    tab = np.zeros((B, TAB), dtype=np.int32)
    free = list(np.random.default_rng(3).permutation(np.arange(1, nblk)))
    for b in range(B):
        for bi in range(pads[b] // bs, (ctx_end + 7) // bs + 1):
            tab[b, bi % TAB] = free.pop()
    tab_d = torch.from_numpy(tab).to(DEV)

    def to_pool(rows_k, rows_v):
        for b in range(B):
            for j in range(pads[b], ctx_end + 1):
                blk, off = int(tab[b, (j // bs) % TAB]), j % bs
                pool_k[blk, :, off] = rows_k[b, :, j]
                pool_v[blk, :, off] = rows_v[b, :, j]

    # --- packed prefill writes the prompt rows (and the prefix read gets them back)
    lens = [ctx_end - p for p in pads]                    # rows [pad_b, ctx_end)
    M = sum(lens)
    qkv = (torch.randn(M, 3 * D, device=DEV) * 0.5).to(dtype)
    row_off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    shift = torch.tensor(pads, dtype=torch.int32, device=DEV)
    out_r, out_p = torch.empty(M, D, dtype=dtype, device=DEV), torch.empty(M, D, dtype=dtype, device=DEV)
    nat.attn_prefill_packed(qkv, out_r, kc_r, vc_r, row_off, shift, B, max(lens), H, smax)
    nat.attn_prefill_packed(qkv, out_p, pool_k, pool_v, row_off, shift, B, max(lens), H, 0, kv_tab=tab_d, kv_bs=bs)
    assert torch.equal(out_r, out_p)
    ref_k, ref_v = pool_k.clone(), pool_v.clone()
    pool_k.zero_()
    pool_v.zero_()
    to_pool(kc_r, vc_r)
    assert torch.equal(pool_k, ref_k) and torch.equal(pool_v, ref_v) and pool_k[0].abs().max().item() == 0
    mq = [6, 3, 9, 1, 4]                                  # query rows behind the cached prefix
    Mq = sum(mq)
    qkv2 = (torch.randn(Mq, 3 * D, device=DEV) * 0.5).to(dtype)
    ro2 = torch.tensor(np.concatenate([[0], np.cumsum(mq)]), dtype=torch.int32, device=DEV)
    pre_len = torch.tensor(lens, dtype=torch.int32, device=DEV)
    pre_row = torch.arange(B, dtype=torch.int32, device=DEV)
    o_r, o_p = torch.empty(Mq, D, dtype=dtype, device=DEV), torch.empty(Mq, D, dtype=dtype, device=DEV)
    nat.attn_prefill_prefix(qkv2, o_r, kc_r, vc_r, ro2, pre_len, pre_row, shift, B, max(mq), H, smax)
    nat.attn_prefill_prefix(qkv2, o_p, pool_k, pool_v, ro2, pre_len, pre_row, shift, B, max(mq), H, 0, kv_tab=tab_d, kv_bs=bs)
    assert torch.equal(o_r, o_p)
    # --- QKV epilogue appends at *pos
    x = (torch.randn(B, D, device=DEV) * 0.5).to(dtype)
    w = (torch.randn(D, 3 * D, device=DEV) * 0.05).to(dtype)
    wp = nat.pack_weight(w)
    bias = torch.randn(3 * D, device=DEV)
    posd = torch.tensor([ctx_end], dtype=torch.int32, device=DEV)
    q_r, q_p = torch.empty(B, D, dtype=dtype, device=DEV), torch.empty(B, D, dtype=dtype, device=DEV)
    nat.gemm_skinny(dtype, B, 3 * D, D, wp, bias, x=x, epi=nat.EPI_QKV_CACHE, y=q_r, kcache=kc_r, vcache=vc_r, pos=posd, heads=H, smax=smax)
    nat.gemm_skinny(dtype, B, 3 * D, D, wp, bias, x=x, epi=nat.EPI_QKV_CACHE, y=q_p, kcache=pool_k, vcache=pool_v, pos=posd, heads=H,
                    smax=0, kv_tab=tab_d, kv_bs=bs)
    assert torch.equal(q_r, q_p)
    ref_k, ref_v = pool_k.clone(), pool_v.clone()
    to_pool(kc_r, vc_r)
    assert torch.equal(pool_k, ref_k) and torch.equal(pool_v, ref_v)
    # --- decode attention over [pad_b, *pos], with and without the shared first keys, and with skipped rows
    padd = torch.tensor(pads, dtype=torch.int32, device=DEV)
    skip = torch.tensor([0, 0, 1, 0, 0], dtype=torch.int32, device=DEV)
    C = 11
    for b in range(1, B):                                 # make the promise true: the first C keys of every row are row 0's
        kc_r[b, :, pads[b]:pads[b] + C] = kc_r[0, :, pads[0]:pads[0] + C]
        vc_r[b, :, pads[b]:pads[b] + C] = vc_r[0, :, pads[0]:pads[0] + C]
    to_pool(kc_r, vc_r)
    for share in (None, torch.tensor([(pads[0] << 8) | C], dtype=torch.int32, device=DEV)):
        a_r = torch.full((B, D), 7.0, dtype=dtype, device=DEV)
        a_p = a_r.clone()
        nat.attn_decode(q_r, kc_r, vc_r, a_r, padd, posd, B, H, smax, skip_rows=skip, kv_share=share)
        nat.attn_decode(q_r, pool_k, pool_v, a_p, padd, posd, B, H, 0, skip_rows=skip, kv_share=share, kv_tab=tab_d, kv_bs=bs)
        assert torch.equal(a_r, a_p)
        assert (a_p[2] == 7.0).all() and not (a_p[1] == 7.0).any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_decode_row_table_equals_permuted_cache(nat, dtype):
    """Beam search: attention through the KV row table (itts_beam_kv_rows + kv_rows argument) equals attention over a cache
    whose rows were permuted by itts_beam_reorder_kv -- over several steps with changing parents."""
    B, nb, H, smax = 2, 3, 4, 64
    R, D = B * nb, H * 64
    kc, vc = rnd(1, R, H, smax, 64, seed=210).to(dtype), rnd(1, R, H, smax, 64, seed=211).to(dtype)
    kc2, vc2 = kc.clone(), vc.clone()
    pad = torch.tensor([0, 0, 0, 3, 3, 3], dtype=torch.int32, device=DEV)
    tbl = torch.zeros(2, R, smax, dtype=torch.int32, device=DEV)
    tbl[0] = torch.arange(R, dtype=torch.int32, device=DEV)[:, None]
    state_t = torch.zeros(8, dtype=torch.int32, device=DEV)   # table form: [0] step, [1] position
    state_c = torch.zeros(8, dtype=torch.int32, device=DEV)
    P0 = 20
    srcs = [[0, 0, 1, 5, 3, 3], [2, 1, 1, 3, 4, 5], [0, 1, 2, 4, 4, 4], [1, 1, 1, 3, 5, 3]]
    for step, src_l in enumerate(srcs):
        src = torch.tensor(src_l, dtype=torch.int32, device=DEV)
        # what the beam step kernel does to the loop state: one more token, one more position
        state_t[0], state_t[1] = step + 1, P0 + step
        state_c[0], state_c[1] = step + 1, P0 + step
        nat.beam_kv_rows(tbl, src, state_t)
        nat.beam_reorder_kv(kc2, vc2, src, state_c, B, nb)
        # the transformer step appends position P for every row (same new k/v in both forms)
        P = P0 + step
        newk, newv = rnd(R, H, 64, seed=220 + step).to(dtype), rnd(R, H, 64, seed=230 + step).to(dtype)
        kc[0, :, :, P], vc[0, :, :, P] = newk, newv
        kc2[0, :, :, P], vc2[0, :, :, P] = newk, newv
        q = rnd(R, D, seed=240 + step).to(dtype)
        pos = torch.tensor([P], dtype=torch.int32, device=DEV)
        out_t, out_c = torch.empty(R, D, dtype=dtype, device=DEV), torch.empty(R, D, dtype=dtype, device=DEV)
        nat.attn_decode(q, kc[0], vc[0], out_t, pad, pos, R, H, smax, kv_rows=tbl, kv_step=state_t[0:1])
        nat.attn_decode(q, kc2[0], vc2[0], out_c, pad, pos, R, H, smax)
        assert torch.equal(out_t, out_c), step
    assert not torch.equal(kc, kc2)   # the table form never moved a cache row


@pytest.mark.parametrize("nb", [2, 3, 5])
def test_attn_decode_row_table_shared_prefix(nat, nb):
    """Table entries of several rows naming ONE physical row (the prompt cached once per batch element, prefill(beams=n),
    plus partly shared generated positions) against a gather of the cache into private rows: same bits, ragged padding, a
    context that is not a multiple of the chunk."""
    dtype = torch.bfloat16
    B, H, smax, S, ctx = 3, 3, 320, 150, 301
    R, D = B * nb, H * 64
    kc, vc = rnd(R, H, smax, 64, seed=310).to(dtype), rnd(R, H, smax, 64, seed=311).to(dtype)
    pad = torch.tensor([0, 17, 149], dtype=torch.int32, device=DEV).repeat_interleave(nb)
    g = torch.Generator().manual_seed(5)
    rows = torch.arange(R, dtype=torch.int32)
    first = (rows // nb) * nb
    t1 = rows[:, None].repeat(1, smax)
    t1[:, :S] = (rows // nb)[:, None]                            # the prompt: cached once, in row b
    anc = first[:, None] + torch.randint(0, nb, (R, smax), generator=g, dtype=torch.int32)
    common = first[:, None] + torch.randint(0, nb, (B, smax), generator=g, dtype=torch.int32).repeat_interleave(nb, 0)
    mix = torch.where(torch.rand(R, smax, generator=g) < 0.3, anc, common)   # 30 % private ancestors
    t1[:, S:ctx - 1] = mix[:, S:ctx - 1]
    tbl = torch.stack([torch.zeros_like(t1), t1]).to(DEV)
    step = torch.tensor([1], dtype=torch.int32, device=DEV)       # parity 1
    pos = torch.tensor([ctx - 1], dtype=torch.int32, device=DEV)
    q = rnd(R, D, seed=312).to(dtype)
    out_t, out_g = torch.empty(R, D, dtype=dtype, device=DEV), torch.empty(R, D, dtype=dtype, device=DEV)
    nat.attn_decode(q, kc, vc, out_t, pad, pos, R, H, smax, kv_rows=tbl, kv_step=step)
    idx = tbl[1].long()[:, None, :, None].expand(R, H, smax, 64)  # private copies: row r position j <- row tbl[r][j]
    nat.attn_decode(q, torch.gather(kc, 0, idx), torch.gather(vc, 0, idx), out_g, pad, pos, R, H, smax)
    assert torch.isfinite(out_t.float()).all() and torch.equal(out_t, out_g)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_prefill_prefix_equals_packed_rows_with_the_prefix_inside(nat, dtype):
    """itts_attn_prefill_prefix (query rows behind keys / values that live in a KV cache, at any row and offset of it) gives,
    bit for bit, the rows itts_attn_prefill_packed gives when the prefix's k / v rows are part of qkv -- prefix lengths 0,
    short, a multiple of the 64-key tile and long; query counts below and above one 64-row tile; cache rows permuted."""
    B, H, smax = 4, 3, 256
    D = H * 64
    pre = [0, 37, 128, 201]          # cached keys per element
    mq = [70, 5, 64, 131]            # query rows per element
    pos0 = [0, 11, 3, 40]            # cache position of the first prefix key
    crow = [2, 0, 3, 1]              # cache row that holds it
    tot = [a + b for a, b in zip(pre, mq)]
    full = rnd(sum(tot), 3 * D, seed=97).to(dtype)
    off_f = np.concatenate([[0], np.cumsum(tot)])
    out_f = torch.empty(sum(tot), D, dtype=dtype, device=DEV)
    nat.attn_prefill_packed(full, out_f, None, None, torch.tensor(off_f, dtype=torch.int32, device=DEV), None, B, max(tot), H, smax)
    kc = rnd(B, H, smax, 64, seed=98).to(dtype)     # noise everywhere else: nothing outside the prefix may be read
    vc = rnd(B, H, smax, 64, seed=99).to(dtype)
    q_rows = []
    for b in range(B):
        blk = full[off_f[b]: off_f[b + 1]].view(tot[b], 3, H, 64)
        kc[crow[b], :, pos0[b]: pos0[b] + pre[b]] = blk[: pre[b], 1].transpose(0, 1)
        vc[crow[b], :, pos0[b]: pos0[b] + pre[b]] = blk[: pre[b], 2].transpose(0, 1)
        q_rows.append(torch.arange(off_f[b] + pre[b], off_f[b + 1]))
    q_rows = torch.cat(q_rows).to(DEV)
    qkv = full[q_rows].contiguous()
    qkv.view(-1, 3, D)[:, 0]                          # (queries of the prefix rows are not needed at all)
    out = torch.empty(qkv.shape[0], D, dtype=dtype, device=DEV)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)   # noqa: E731
    nat.attn_prefill_prefix(qkv, out, kc, vc, i32(np.concatenate([[0], np.cumsum(mq)])), i32(pre), i32(crow), i32(pos0), B, max(mq),
                            H, smax)
    assert torch.isfinite(out.float()).all()
    assert torch.equal(out, out_f[q_rows])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attn_prefill_packed_equals_left_padded(nat, dtype):
    """Packed rows (no padding rows at all) give the attention outputs and the KV cache contents of the left-padded form
    for every real position."""
    B, H, S, smax = 3, 2, 150, 192
    D = H * 64
    pads = [0, 37, 101]
    qkv = rnd(B, S, 3 * D, seed=95).to(dtype)
    pad = torch.tensor(pads, dtype=torch.int32, device=DEV)
    out_ref = torch.empty(B, S, D, dtype=dtype, device=DEV)
    kc_ref, vc_ref = torch.zeros(B, H, smax, 64, dtype=dtype, device=DEV), torch.zeros(B, H, smax, 64, dtype=dtype, device=DEV)
    nat.attn_prefill(qkv, out_ref, kc_ref, vc_ref, pad, B, S, H, smax)
    rows = torch.cat([torch.arange(b * S + pads[b], (b + 1) * S) for b in range(B)]).to(DEV)
    off = [0]
    for p in pads:
        off.append(off[-1] + S - p)
    row_off = torch.tensor(off, dtype=torch.int32, device=DEV)
    q_p = qkv.view(B * S, 3 * D)[rows].contiguous()
    out_p = torch.empty(q_p.shape[0], D, dtype=dtype, device=DEV)
    kc, vc = torch.zeros_like(kc_ref), torch.zeros_like(vc_ref)
    nat.attn_prefill_packed(q_p, out_p, kc, vc, row_off, pad, B, S, H, smax)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (out_p.float() - out_ref.view(B * S, D)[rows].float()).abs().max().item() < tol
    for b in range(B):
        assert torch.equal(kc[b, :, pads[b]:S], kc_ref[b, :, pads[b]:S]) and torch.equal(vc[b, :, pads[b]:S], vc_ref[b, :, pads[b]:S])
        assert kc[b, :, :pads[b]].abs().max().item() == 0 if pads[b] else True


def test_aa_snake_c24_pairs_batch_elements_same_bits():
    """fp16, C = 24, dense even batch: two batch elements share one 48-channel slice of the MFMA activation; every channel's FIR
    sums are its own, so the result equals, bit for bit, the single-element form (odd batch: one 32-channel slice per element)."""
    from indextts import _native as nat
    from indextts.BigVGAN.models import kaiser_sinc_filter
    T, C = 36000, 24
    g = torch.Generator().manual_seed(24)
    x = (torch.randn(4, T, C, generator=g) * 0.9).to(DEV).half()
    al, be = (torch.randn(C, generator=g) * 0.3).to(DEV), (torch.randn(C, generator=g) * 0.3).to(DEV)
    f = kaiser_sinc_filter()
    paired = torch.empty_like(x)
    nat.aa_snake(x, al, be, f, f, layout=0, out=paired)                      # B = 4: pairs (0, 1), (2, 3)
    for b in range(4):
        one = torch.empty(1, T, C, dtype=torch.float16, device=DEV)
        nat.aa_snake(x[b:b + 1].contiguous(), al, be, f, f, layout=0, out=one)   # B = 1: the unpaired form
        assert torch.equal(one[0], paired[b]), b
    ragged = torch.zeros_like(x)
    nat.aa_snake(x, al, be, f, f, layout=0, out=ragged, valid_rows=torch.tensor([T, 100, T, 20000], dtype=torch.int32, device=DEV))
    assert torch.equal(ragged[0], paired[0]) and torch.equal(ragged[2], paired[2])   # (a ragged batch takes the unpaired form)
