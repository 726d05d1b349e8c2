"""ctypes binding of libindextts_hip.so (the C ABI declared in include/indextts_hip.h).

The product path has NO CPU/PyTorch fallback for the hot ops: if the shared library is missing or a call fails, an
exception is raised.  torch is used here only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

F32, BF16, F16 = 0, 1, 2
EPI_STORE, EPI_GELU_STORE, EPI_RESID_F32, EPI_QKV_CACHE, EPI_STORE_F32, EPI_SLAB_F32, EPI_SILU_STORE = 0, 1, 2, 3, 4, 5, 6
EPI_RELU_AFFINE_STORE, EPI_RELU_AFFINE_TANH_STORE = 7, 8

_DT = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_lib", "libindextts_hip.so")
# tools/ only: ITTS_HIP_LIB=<.../libindextts_hip_diag.so> loads the diagnostic build (tuning overrides, time stamps) instead
_LIB_OVERRIDE = os.environ.get("ITTS_HIP_LIB")


class NativeError(RuntimeError):
    pass


class SkinnyArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("wp", C.c_void_p),
                ("bias", C.c_void_p), ("x", C.c_void_p), ("epi", C.c_int), ("y", C.c_void_p), ("yf", C.c_void_p),
                ("kcache", C.c_void_p), ("vcache", C.c_void_p), ("pos", C.c_void_p), ("heads", C.c_int),
                ("smax", C.c_int), ("ksplit", C.c_int), ("ln_c", C.c_void_p), ("ln_eps", C.c_float), ("bump", C.c_void_p),
                ("rows_per_wg", C.c_int), ("wide_wg", C.c_int), ("kv_tab", C.c_void_p), ("kv_bs", C.c_int),
                ("x_packed", C.c_int), ("y_packed", C.c_int), ("y_row0", C.c_int), ("y_mtp", C.c_int), ("x_mtp", C.c_int),
                ("post_scale", C.c_void_p), ("post_shift", C.c_void_p)]


class MhaArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("Tq", C.c_int), ("Tk", C.c_int), ("H", C.c_int), ("q", C.c_void_p), ("k", C.c_void_p),
                ("v", C.c_void_p), ("q_stride", C.c_int64), ("k_stride", C.c_int64), ("v_stride", C.c_int64), ("pos", C.c_void_p),
                ("bias_u", C.c_void_p), ("bias_v", C.c_void_p), ("scale", C.c_float), ("out", C.c_void_p), ("out_mtp", C.c_int)]


class RowsArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("D", C.c_int), ("x", C.c_void_p), ("slab", C.c_void_p), ("nslab", C.c_int),
                ("bias", C.c_void_p), ("norm", C.c_int), ("w", C.c_void_p), ("b", C.c_void_p), ("eps", C.c_float),
                ("y", C.c_void_p), ("y_packed", C.c_void_p), ("y_row0", C.c_int), ("y_mtp", C.c_int)]


class LnReduceArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("M", C.c_int), ("D", C.c_int), ("h", C.c_void_p), ("slab", C.c_void_p), ("nslab", C.c_int),
                ("slab_stride", C.c_int), ("bias", C.c_void_p), ("w", C.c_void_p), ("b", C.c_void_p), ("w2", C.c_void_p),
                ("b2", C.c_void_p), ("y", C.c_void_p), ("y_packed", C.c_int), ("state_bump", C.c_void_p), ("lora_b", C.c_void_p),
                ("lora_r", C.c_int)]


class ConvArgs(C.Structure):
    _fields_ = [("dtype", C.c_int), ("B", C.c_int), ("Tin", C.c_int), ("Tout", C.c_int), ("Cin", C.c_int),
                ("N", C.c_int), ("taps", C.c_int), ("off0", C.c_int), ("dil", C.c_int), ("x", C.c_void_p),
                ("x_bstride", C.c_int64), ("wp", C.c_void_p), ("bias", C.c_void_p), ("bias2", C.c_void_p),
                ("act", C.c_int), ("y", C.c_void_p), ("y_f32", C.c_int), ("y_bstride", C.c_int64),
                ("y_shift", C.c_int64), ("y_limit", C.c_int64), ("resid", C.c_void_p), ("accumulate", C.c_int),
                ("scale", C.c_float), ("valid_rows", C.c_void_p), ("ksplit", C.c_int)]


class SampleArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("B", C.c_int), ("V", C.c_int), ("ldl", C.c_int), ("tokens", C.c_void_p),
                ("history", C.c_void_p), ("hist_cap", C.c_int), ("finished", C.c_void_p), ("state", C.c_void_p),
                ("extra_ids", C.c_void_p), ("n_extra", C.c_int), ("force_stop", C.c_void_p),
                ("rep_penalty", C.c_float), ("temperature", C.c_float), ("top_p", C.c_float), ("top_k", C.c_int),
                ("do_sample", C.c_int), ("seed", C.c_uint64), ("stop_token", C.c_int), ("dbg_scores", C.c_void_p),
                ("no_advance", C.c_int), ("row_step0", C.c_void_p)]


class BeamArgs(C.Structure):
    _fields_ = [("logits", C.c_void_p), ("B", C.c_int), ("num_beams", C.c_int), ("V", C.c_int), ("ldl", C.c_int),
                ("tokens", C.c_void_p), ("src", C.c_void_p), ("beam_scores", C.c_void_p), ("hist", C.c_void_p),
                ("hist_cap", C.c_int), ("hyp_score", C.c_void_p), ("hyp_len", C.c_void_p), ("hyp_tok", C.c_void_p),
                ("n_hyp", C.c_void_p), ("worst", C.c_void_p), ("done", C.c_void_p), ("state", C.c_void_p),
                ("extra_ids", C.c_void_p), ("n_extra", C.c_int), ("rep_penalty", C.c_float), ("temperature", C.c_float),
                ("top_p", C.c_float), ("length_penalty", C.c_float), ("top_k", C.c_int), ("do_sample", C.c_int),
                ("seed", C.c_uint64), ("eos_token", C.c_int),
                ("cand_scores", C.c_void_p), ("cand_ids", C.c_void_p), ("cand_n", C.c_void_p)]


BEAM_CAND = 1024   # ITTS_BEAM_CAND: candidate slots per row in the beam step's scratch


_SIGNATURES = {
    "itts_abi_version": (C.c_int, []),
    "itts_last_error": (C.c_char_p, []),
    "itts_packed_bytes": (C.c_int64, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "itts_pack_weight": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_aa_snake_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "itts_gemm_skinny": (C.c_int, [C.POINTER(SkinnyArgs), C.c_void_p]),
    "itts_skinny_plan": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int)]),
    "itts_gemm_conv": (C.c_int, [C.POINTER(ConvArgs), C.c_void_p]),
    "itts_layernorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_ln_reduce": (C.c_int, [C.POINTER(LnReduceArgs), C.c_void_p]),
    "itts_embed_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "itts_attn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_void_p]),
    "itts_attn_prefill": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                    C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_attn_prefill_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "itts_attn_prefill_prefix": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "itts_attn_prefill_shared": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                           C.c_void_p]),
    "itts_sample": (C.c_int, [C.POINTER(SampleArgs), C.c_void_p]),
    "itts_beam_step": (C.c_int, [C.POINTER(BeamArgs), C.c_void_p]),
    "itts_beam_kv_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "itts_beam_reorder_kv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, C.c_int64, C.c_int, C.c_void_p]),
    "itts_tanh_pcm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "itts_subsample_conv": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_mha_small": (C.c_int, [C.POINTER(MhaArgs), C.c_void_p]),
    "itts_glu_dwconv_ln_silu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "itts_rows": (C.c_int, [C.POINTER(RowsArgs), C.c_void_p]),
    "itts_geglu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_im2col_reflect": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_res2_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_se_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                               C.c_int, C.c_int, C.c_void_p]),
    "itts_scale_resid": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "itts_col_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_void_p]),
    "itts_kv_share_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                     C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "itts_prefix_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)
_lib = None


def lib():
    """Load (once) and return the shared library; raise NativeError if it is not built."""
    global _lib
    if _lib is None:
        path = _LIB_OVERRIDE or LIB_PATH
        if not os.path.exists(path):
            raise NativeError(f"{path} not found: build it with `make -C index-tts-lora_amd/csrc` "
                              f"(or python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        L = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.itts_abi_version() != 8:
            raise NativeError("libindextts_hip.so ABI version mismatch")
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        raise NativeError(f"{what} failed (code {rc}): {lib().itts_last_error().decode()}")


def dt(t: torch.dtype) -> int:
    return _DT[t]


def debug_set(key: int, value: int):
    """Tuning override of the DIAGNOSTIC build (tools/ only; include/indextts_hip_diag.h).  The product library has no
    such entry point and no mutable knobs."""
    L = lib()
    try:
        fn = L.itts_debug_set
    except AttributeError:
        raise NativeError("itts_debug_set exists only in libindextts_hip_diag.so: run the tool with "
                          "ITTS_HIP_LIB=index-tts-lora_amd/indextts/_lib/libindextts_hip_diag.so (make -C index-tts-lora_amd/csrc diag)")
    fn.restype, fn.argtypes = C.c_int, [C.c_int, C.c_int]
    _check(fn(int(key), int(value)), "itts_debug_set")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


# Held around every CUDA-graph capture and by helper threads while they enqueue work (infer.BatchPipeline): a capture in
# the default (global) error mode is invalidated by allocations / synchronisation on any other thread.
CAPTURE_LOCK = threading.RLock()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """The current torch stream's hipStream_t.  One of these per native launch: the raw getters (what torch's own compiled code
    paths use) cost ~0.3 us, torch.cuda.current_stream() ~8 us -- 5 ms of host time per bench step, and the prefill's ~25 us
    kernels are launched from Python at that rate."""
    if _raw_stream is not None and _raw_device is not None:
        return C.c_void_p(_raw_stream(_raw_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None:
            if not t.is_cuda:
                raise NativeError("HIP kernels need device tensors (no CPU fallback)")
            if not t.is_contiguous():
                raise NativeError("HIP kernels need contiguous tensors")


# ------------------------------------------------------------------------------------------------------ wrappers
def pack_weight(w: torch.Tensor) -> torch.Tensor:
    """w [K,N] or [taps,K,N] (device, T) -> packed uint8 buffer in MFMA B-fragment order."""
    if w.dim() == 2:
        w = w[None]
    w = w.contiguous()
    _dev(w)
    taps, K, N = w.shape
    nbytes = lib().itts_packed_bytes(taps, K, N, dt(w.dtype))
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    _check(lib().itts_pack_weight(_p(w), _p(out), taps, K, N, dt(w.dtype), _stream()), "itts_pack_weight")
    return out


def aa_snake(x, alpha_log, beta_log, up_f, down_f, layout=0, out=None, valid_rows=None):
    """x [B,T,C] (layout 0) or [B,C,T] (layout 1); filters are HOST float32 tensors/arrays of 12 taps.
    valid_rows int32 [B] (layout 0): per-element sequence lengths of a ragged batch (rows past them are not written)."""
    _dev(x, alpha_log, beta_log)
    if layout == 0:
        B, T, Cn = x.shape
    else:
        B, Cn, T = x.shape
    y = torch.empty_like(x) if out is None else out
    uf = (C.c_float * 12)(*[float(v) for v in up_f])
    df = (C.c_float * 12)(*[float(v) for v in down_f])
    _check(lib().itts_aa_snake_fwd(_p(x), _p(y), _p(alpha_log), _p(beta_log), C.cast(uf, C.c_void_p),
                                   C.cast(df, C.c_void_p), B, T, Cn, dt(x.dtype), layout, _p(valid_rows), _stream()),
           "itts_aa_snake_fwd")
    return y


def gemm_skinny(dtype, M, N, K, wp, bias=None, x=None, epi=EPI_STORE, y=None, yf=None, kcache=None, vcache=None, pos=None,
                heads=0, smax=0, ksplit=1, x_packed=False, y_packed=False, ln_c=None, ln_eps=1e-5, bump=None, rows_per_wg=0,
                wide_wg=False, kv_tab=None, kv_bs=0, y_row0=0, y_mtp=0, x_mtp=0, post=None):
    """ln_c fp32 [N]: LayerNorm folded into the GEMM -- x holds the RAW rows, wp = pack(gamma . W), bias = beta W + b
    (see itts_skinny_args).  EPI_RESID_F32: yf += x W + bias, and y (optional, T) receives a copy of the new rows.
    bump: int32 device word the launch increments.  rows_per_wg / wide_wg: launch-geometry hints.
    y_row0 / y_mtp: a packed y's rows land at [y_row0, y_row0 + M) of an operand of y_mtp row tiles; x_mtp: a packed x is the
    first M rows of an operand of x_mtp row tiles.  post = (scale, shift) fp32 [N]: the EPI_RELU_AFFINE_* epilogues."""
    a = SkinnyArgs()
    a.dtype, a.M, a.N, a.K = dt(dtype), M, N, K
    a.wp, a.bias, a.x = _p(wp), _p(bias), _p(x)
    a.epi, a.y, a.yf = epi, _p(y), _p(yf)
    a.kcache, a.vcache, a.pos, a.heads, a.smax, a.ksplit = _p(kcache), _p(vcache), _p(pos), heads, smax, ksplit
    a.x_packed, a.y_packed = int(bool(x_packed)), int(bool(y_packed))
    a.ln_c, a.ln_eps, a.bump = _p(ln_c), float(ln_eps), _p(bump)
    a.rows_per_wg, a.wide_wg = int(rows_per_wg), int(bool(wide_wg))
    a.kv_tab, a.kv_bs = _p(kv_tab), int(kv_bs)
    a.y_row0, a.y_mtp, a.x_mtp = int(y_row0), int(y_mtp), int(x_mtp)
    if post is not None:
        a.post_scale, a.post_shift = _p(post[0]), _p(post[1])
    _check(lib().itts_gemm_skinny(C.byref(a), _stream()), "itts_gemm_skinny")


def subsample_conv(mel, w, b, y):
    """mel fp32 [T, F], w fp32 [C, 9], b [C] -> y T [T2, C * F2] = relu(conv2d 3x3 stride 2), laid out for the Linear behind it."""
    _dev(mel, w, b, y)
    T, Fq = mel.shape
    _check(lib().itts_subsample_conv(_p(mel), _p(w), _p(b), _p(y), T, Fq, w.shape[0], dt(y.dtype), _stream()), "itts_subsample_conv")
    return y


def mha_small(q, k, v, out, Tq, Tk, H, q_stride, k_stride, v_stride, out_mtp, scale, pos=None, bias_u=None, bias_v=None):
    """Short-sequence attention (head dim 64) into the packed layout; pos / bias_u / bias_v: the Conformer's relative-position term.
    q / k / v may be views into one buffer (pointer + row stride in elements)."""
    a = MhaArgs()
    a.dtype, a.Tq, a.Tk, a.H = dt(out.dtype), Tq, Tk, H
    a.q, a.k, a.v = _p(q), _p(k), _p(v)
    a.q_stride, a.k_stride, a.v_stride = q_stride, k_stride, v_stride
    a.pos, a.bias_u, a.bias_v = _p(pos), _p(bias_u), _p(bias_v)
    a.scale, a.out, a.out_mtp = float(scale), _p(out), int(out_mtp)
    _check(lib().itts_mha_small(C.byref(a), _stream()), "itts_mha_small")
    return out


def glu_dwconv_ln_silu(x, w, b, ln_w, ln_b, y, T, Cn, y_mtp, eps=1e-5):
    """x T [T, 2C] -> GLU -> depthwise conv (w fp32 [C, taps]) -> LayerNorm -> SiLU -> y T packed [T, C]."""
    _dev(x, w, b, ln_w, ln_b, y)
    _check(lib().itts_glu_dwconv_ln_silu(_p(x), _p(w), _p(b), _p(ln_w), _p(ln_b), _p(y), T, Cn, w.shape[1], int(y_mtp), float(eps),
                                         dt(y.dtype), _stream()), "itts_glu_dwconv_ln_silu")
    return y


def rows(M, D, dtype, x=None, slab=None, nslab=0, bias=None, norm=0, w=None, b=None, eps=1e-5, y=None, y_packed=None, y_row0=0,
         y_mtp=0):
    """Row operations on an fp32 stream: (x) + bias + sum(slabs) -> [LayerNorm | l2-normalise * sqrt(D) * w] -> y fp32 and / or a
    packed T copy (itts_rows)."""
    a = RowsArgs()
    a.dtype, a.M, a.D = dt(dtype), M, D
    a.x, a.slab, a.nslab, a.bias = _p(x), _p(slab), int(nslab), _p(bias)
    a.norm, a.w, a.b, a.eps = int(norm), _p(w), _p(b), float(eps)
    a.y, a.y_packed, a.y_row0, a.y_mtp = _p(y), _p(y_packed), int(y_row0), int(y_mtp)
    _check(lib().itts_rows(C.byref(a), _stream()), "itts_rows")


def geglu(h, y, M, Kp, y_mtp=0):
    """h T [M, 2 Kp] (x | gate) -> y T packed [M, Kp] = gelu(gate) * x."""
    _dev(h, y)
    _check(lib().itts_geglu(_p(h), _p(y), M, Kp, int(y_mtp), dt(y.dtype), _stream()), "itts_geglu")
    return y


def prefix_rows(text, conds, text_emb, text_pos, start_tok, stop_tok):
    """text int64 [B, L], conds fp32 [1 | B, C, D] -> (emb fp32 [B, P, D], mask int64 [B, P + 1], pad int32 [B]), P = C + L + 2:
    the GPT prompt rows of prepare_gpt_inputs in one launch (itts_prefix_rows)."""
    _dev(text, conds, text_emb, text_pos)
    B, L = text.shape
    Bc, Cn, D = conds.shape
    P = Cn + L + 2
    emb = torch.empty(B, P, D, dtype=torch.float32, device=text.device)
    mask = torch.empty(B, P + 1, dtype=torch.int64, device=text.device)
    pad = torch.empty(B, dtype=torch.int32, device=text.device)
    _check(lib().itts_prefix_rows(_p(text), _p(conds), Bc, _p(text_emb), _p(text_pos), _p(emb), _p(mask), _p(pad), B, L, Cn, D,
                                  int(start_tok), int(stop_tok), text_emb.shape[0], text_pos.shape[0], _stream()), "itts_prefix_rows")
    return emb, mask, pad


def im2col_reflect(x, y, taps, dil, Kp, y_mtp):
    """x fp32 [T, F] -> y T packed [T, Kp]: the k-tap reflect-padded convolution's operand (itts_im2col_reflect)."""
    _dev(x, y)
    _check(lib().itts_im2col_reflect(_p(x), _p(y), x.shape[0], x.shape[1], taps, dil, Kp, y_mtp, dt(y.dtype), _stream()), "itts_im2col_reflect")


def res2_step(y1, cat, wp, bias, scale, shift, T, mtp, chunk, dil, first):
    _check(lib().itts_res2_step(_p(y1), _p(cat), _p(wp), _p(bias), _p(scale), _p(shift), T, mtp, chunk, dil, int(bool(first)),
                                dt(cat.dtype), _stream()), "itts_res2_step")


def se_gate(y, w1, b1, w2, b2, gate, T, Cn, H, mtp):
    _dev(y, w1, b1, w2, b2, gate)
    _check(lib().itts_se_gate(_p(y), _p(w1), _p(b1), _p(w2), _p(b2), _p(gate), T, Cn, H, mtp, dt(y.dtype), _stream()), "itts_se_gate")


def scale_resid(y, res, gate, out, T, Cn, mtp):
    """out = gate * y + res over packed [T, Cn] operands (res / out may be k-step runs of a wider packed operand: views)."""
    _check(lib().itts_scale_resid(_p(y), _p(res), _p(gate), _p(out), T, Cn, mtp, dt(y.dtype), _stream()), "itts_scale_resid")


def col_stats(x, out, T, Cn, mtp, logit=None, scale=None, shift=None):
    """out T [2 Cn] = [mean | std] over time of packed x [T, Cn], optionally softmax(logit)-weighted and affine-mapped."""
    _check(lib().itts_col_stats(_p(x), _p(logit), _p(scale), _p(shift), _p(out), T, Cn, mtp, dt(out.dtype), _stream()), "itts_col_stats")


def packed_rows(M: int) -> int:
    """Rows a packed-activation buffer must provide for M logical rows (16-row tiles)."""
    return (M + 15) // 16 * 16


def pack_activation(x: torch.Tensor) -> torch.Tensor:
    """Row-major [M, K] -> the packed activation layout (include/indextts_hip.h), as a flat tensor of packed_rows(M)*K
    elements.  Host-side torch restatement for tests and tools; the kernels write this layout themselves."""
    M, K = x.shape
    E = 4 if x.dtype == torch.float32 else 8
    KS = 4 * E
    mtp = (M + 15) // 16
    xp = torch.zeros(mtp * 16, K, dtype=x.dtype, device=x.device)
    xp[:M] = x
    v = xp.view(mtp, 16, K // KS, 4, E).permute(2, 0, 3, 1, 4).contiguous()   # [ks][mt][g][c][e]
    return v.view(-1)


def unpack_activation(xp: torch.Tensor, M: int, K: int) -> torch.Tensor:
    E = 4 if xp.dtype == torch.float32 else 8
    KS = 4 * E
    mtp = (M + 15) // 16
    v = xp.view(K // KS, mtp, 4, 16, E).permute(1, 3, 0, 2, 4).contiguous().view(mtp * 16, K)
    return v[:M]


def skinny_plan(dtype, M, N, K, ksplit=1, rows_per_wg=0, wide_wg=False, fold=False) -> dict:
    """Launch geometry itts_gemm_skinny would use (host-only)."""
    out = (C.c_int * 8)()
    _check(lib().itts_skinny_plan(dt(dtype), M, N, K, ksplit, int(rows_per_wg), int(bool(wide_wg)), int(bool(fold)), out),
           "itts_skinny_plan")
    return dict(grid=(out[0], out[1], out[6]), waves=out[2], tiles_per_wg=out[3], ksteps_per_wave=out[4], lds=out[5],
                row_tiles_per_wg=out[7])


def gemm_conv(dtype, B, Tin, Tout, Cin, N, wp, x, y, taps=1, off0=0, dil=1, x_bstride=None, bias=None, bias2=None,
              act=0, y_f32=False, y_bstride=None, y_shift=0, y_limit=None, resid=None, accumulate=False, scale=1.0,
              valid_rows=None, ksplit=1):
    """valid_rows int32 [B]: ragged batch -- input rows >= valid_rows[b] read as zeros, tiles that only see them are skipped."""
    a = ConvArgs()
    a.dtype, a.B, a.Tin, a.Tout, a.Cin, a.N = dt(dtype), B, Tin, Tout, Cin, N
    a.taps, a.off0, a.dil = taps, off0, dil
    a.x, a.x_bstride = _p(x), Tin * Cin if x_bstride is None else x_bstride
    a.wp, a.bias, a.bias2, a.act = _p(wp), _p(bias), _p(bias2), act
    a.y, a.y_f32 = _p(y), int(y_f32)
    a.y_bstride = Tout * N if y_bstride is None else y_bstride
    a.y_shift = y_shift
    a.y_limit = Tout * N if y_limit is None else y_limit
    a.resid, a.accumulate, a.scale = _p(resid), int(accumulate), float(scale)
    a.valid_rows = _p(valid_rows)
    a.ksplit = int(ksplit)
    _check(lib().itts_gemm_conv(C.byref(a), _stream()), "itts_gemm_conv")


def layernorm(h, w, b, out, w2=None, b2=None):
    """h fp32 [M,D] -> out (T or fp32) [M,D]."""
    _dev(h, w, b, out)
    M, D = h.shape
    y_f32 = out.dtype == torch.float32
    d = F32 if y_f32 else dt(out.dtype)
    _check(lib().itts_layernorm(_p(h), _p(w), _p(b), _p(w2), _p(b2), _p(out), int(y_f32), M, D, d, _stream()),
           "itts_layernorm")
    return out


def ln_reduce(h, w, b, out, slab=None, nslab=0, bias=None, w2=None, b2=None, state_bump=None, y_packed=False, slab_stride=0,
              lora_b=None):
    """h fp32 [M,D] (updated in place when nslab > 0) -> out T [M,D] = LN(h + bias + sum(slabs) [+ (x A) lora_b]).
    state_bump: int32[2] device words incremented once by the launch (decode loop: step counter and cache position).
    lora_b fp32 [r, D] (= B^T alpha/r-free: the scale rides on A): runtime LoRA of the producing projection, whose slabs are
    slab_stride = D + 16*ceil(r/16) wide."""
    M, D = h.shape
    a = LnReduceArgs()
    a.dtype, a.M, a.D = dt(out.dtype), M, D
    a.h, a.slab, a.nslab, a.slab_stride = _p(h), _p(slab), int(nslab), int(slab_stride)
    a.bias, a.w, a.b, a.w2, a.b2 = _p(bias), _p(w), _p(b), _p(w2), _p(b2)
    a.y, a.y_packed, a.state_bump = _p(out), int(bool(y_packed)), _p(state_bump)
    if lora_b is not None:
        a.lora_b, a.lora_r = _p(lora_b), lora_b.shape[0]
    _check(lib().itts_ln_reduce(C.byref(a), _stream()), "itts_ln_reduce")
    return out


def embed_step(tokens, table, pos_table, step, pos_add, h, bump=None, row_step0=None, h_packed=None):
    """bump (int32 device word or None) is incremented once (the decode loop's cache position).
    row_step0 (int32 [B] or None): the loop step at which each row started (slot refill).
    h_packed (T, packed activation layout, or None): T-typed copy of the rows for the LayerNorm-folded QKV GEMM."""
    B, D = h.shape
    d = F32 if h_packed is None else dt(h_packed.dtype)
    _check(lib().itts_embed_step(_p(tokens), _p(table), _p(pos_table), _p(step), pos_add, _p(h), B, D, _p(bump), _p(row_step0),
                                 pos_table.shape[0], _p(h_packed), d, _stream()), "itts_embed_step")


def attn_decode(q, kcache, vcache, out, pad, pos, B, H, smax, out_packed=False, kv_rows=None, kv_step=None, skip_rows=None,
                kv_share=None, kv_tab=None, kv_bs=0):
    """kv_rows int32 [2][B][smax] + kv_step (device word): beam-search row table instead of permuted cache rows.
    skip_rows int32 [B]: rows with a nonzero entry are left out (their slice of `out` is not written).
    kv_share (device word (p0 << 8) | C): the first C keys of every row equal cache row 0's positions [p0, p0 + C)."""
    _check(lib().itts_attn_decode(_p(q), _p(kcache), _p(vcache), _p(out), _p(pad), _p(pos), B, H, smax, dt(q.dtype),
                                  int(bool(out_packed)), _p(kv_rows), _p(kv_step), _p(skip_rows), _p(kv_share), _p(kv_tab), int(kv_bs),
                                  _stream()),
           "itts_attn_decode")


def attn_prefill(qkv, out, kcache, vcache, pad, B, S, H, smax):
    _check(lib().itts_attn_prefill(_p(qkv), _p(out), _p(kcache), _p(vcache), _p(pad), B, S, H, smax, dt(qkv.dtype),
                                   _stream()), "itts_attn_prefill")


def attn_prefill_packed(qkv, out, kcache, vcache, row_off, cache_shift, B, Smax, H, smax, kv_tab=None, kv_bs=0):
    """Packed rows (no padding): row_off int32 [B+1]; cache row of local row i = cache_shift[b] + i.
    kv_tab / kv_bs: the caches are a paged pool behind a block table (include/indextts_hip.h)."""
    _check(lib().itts_attn_prefill_packed(_p(qkv), _p(out), _p(kcache), _p(vcache), _p(row_off), _p(cache_shift), B, Smax, H,
                                          smax, dt(qkv.dtype), _p(kv_tab), int(kv_bs), _stream()), "itts_attn_prefill_packed")


def attn_prefill_prefix(qkv, out, kcache, vcache, row_off, pre_len, pre_row, pre_pos0, B, Smax, H, smax, kv_tab=None, kv_bs=0):
    """Packed query rows behind a cached prefix: element b = pre_len[b] keys of cache row pre_row[b] from position pre_pos0[b],
    then its rows of qkv (see include/indextts_hip.h)."""
    if not (kcache.is_contiguous() and vcache.is_contiguous()):
        raise NativeError("itts_attn_prefill_prefix: the caches must be contiguous [rows][H][smax][64] views")
    _check(lib().itts_attn_prefill_prefix(_p(qkv), _p(out), _p(kcache), _p(vcache), _p(row_off), _p(pre_len), _p(pre_row),
                                          _p(pre_pos0), B, Smax, H, smax, dt(qkv.dtype), _p(kv_tab), int(kv_bs), _stream()),
           "itts_attn_prefill_prefix")


def attn_prefill_shared(qkv, out, kcache, vcache, row_off, pre_len, pre_row0, w_row, w_pos0, E, Smax, H, smax, kv_tab=None, kv_bs=0):
    """Packed elements behind a prefix block that lives in qkv itself (computed once, shared); own rows appended to the caches
    (see include/indextts_hip.h)."""
    if kcache is not None and not (kcache.is_contiguous() and vcache.is_contiguous()):
        raise NativeError("itts_attn_prefill_shared: the caches must be contiguous [rows][H][smax][64] views")
    _check(lib().itts_attn_prefill_shared(_p(qkv), _p(out), _p(kcache), _p(vcache), _p(row_off), _p(pre_len), _p(pre_row0),
                                          _p(w_row), _p(w_pos0), E, Smax, H, smax, dt(qkv.dtype), _p(kv_tab), int(kv_bs), _stream()),
           "itts_attn_prefill_shared")


def kv_share_rows(kc, vc, B, H, Cn, p0, pad, smax, kv_tab=None, kv_bs=0):
    """kc / vc: the whole caches [L, ...] (all layers in one launch): row 0's positions [p0, p0 + Cn) -> [pad[b], pad[b] + Cn) of
    rows 1 .. B-1 (itts_kv_share_rows)."""
    _dev(kc, vc, pad)
    _check(lib().itts_kv_share_rows(_p(kc), _p(vc), kc.shape[0], kc.stride(0), B, H, Cn, p0, _p(pad), smax, _p(kv_tab), int(kv_bs),
                                    dt(kc.dtype), _stream()), "itts_kv_share_rows")


def sample(logits, tokens, history, finished, state, extra_ids, force_stop, rep_penalty, temperature, top_k, top_p,
           do_sample, seed, stop_token, dbg_scores=None, no_advance=False, row_step0=None):
    a = SampleArgs()
    B, V = logits.shape
    a.logits, a.B, a.V, a.ldl = _p(logits), B, V, logits.stride(0)
    a.tokens, a.history, a.hist_cap = _p(tokens), _p(history), history.shape[1]
    a.finished, a.state = _p(finished), _p(state)
    a.extra_ids, a.n_extra = _p(extra_ids), 0 if extra_ids is None else extra_ids.numel()
    a.force_stop = _p(force_stop)
    a.rep_penalty, a.temperature, a.top_p = float(rep_penalty), float(temperature), float(top_p)
    a.top_k, a.do_sample, a.seed, a.stop_token = int(top_k), int(bool(do_sample)), int(seed), int(stop_token)
    a.dbg_scores = _p(dbg_scores)
    a.no_advance = int(bool(no_advance))
    a.row_step0 = _p(row_step0)
    _check(lib().itts_sample(C.byref(a), _stream()), "itts_sample")


def beam_step(logits, num_beams, tokens, src, beam_scores, hist, hyp_score, hyp_len, hyp_tok, n_hyp, worst, done, state,
              extra_ids, rep_penalty, temperature, top_k, top_p, do_sample, length_penalty, seed, eos_token, scratch=None):
    """logits fp32 [B*num_beams, V]; hist int32 [2, B*num_beams, cap]; see include/indextts_hip.h (itts_beam_args).
    scratch = (cand_scores fp32 [R, BEAM_CAND], cand_ids int32 [R, BEAM_CAND], cand_n int32 [R]) from beam_scratch(R);
    allocated per call when omitted (tests) -- a captured decode loop must pass persistent buffers."""
    a = BeamArgs()
    R, V = logits.shape
    if scratch is None:
        scratch = beam_scratch(R, logits.device)
    cs, cid, cn = scratch
    assert cs.shape[0] >= R and cs.shape[1] == BEAM_CAND and cid.shape == cs.shape and cn.numel() >= R
    a.cand_scores, a.cand_ids, a.cand_n = _p(cs), _p(cid), _p(cn)
    a.logits, a.B, a.num_beams, a.V, a.ldl = _p(logits), R // num_beams, int(num_beams), V, logits.stride(0)
    a.tokens, a.src, a.beam_scores = _p(tokens), _p(src), _p(beam_scores)
    a.hist, a.hist_cap = _p(hist), hist.shape[2]
    a.hyp_score, a.hyp_len, a.hyp_tok, a.n_hyp, a.worst = _p(hyp_score), _p(hyp_len), _p(hyp_tok), _p(n_hyp), _p(worst)
    a.done, a.state = _p(done), _p(state)
    a.extra_ids, a.n_extra = _p(extra_ids), 0 if extra_ids is None else extra_ids.numel()
    a.rep_penalty, a.temperature, a.top_p, a.length_penalty = float(rep_penalty), float(temperature), float(top_p), float(length_penalty)
    a.top_k, a.do_sample, a.seed, a.eos_token = int(top_k), int(bool(do_sample)), int(seed), int(eos_token)
    _check(lib().itts_beam_step(C.byref(a), _stream()), "itts_beam_step")


def beam_scratch(rows, device="cuda"):
    """Scratch between the two launches of a beam step (per-row candidates -> pooling), uninitialised."""
    return (torch.empty(rows, BEAM_CAND, dtype=torch.float32, device=device),
            torch.empty(rows, BEAM_CAND, dtype=torch.int32, device=device),
            torch.empty(rows, dtype=torch.int32, device=device))


def beam_kv_rows(kv_rows, src, state):
    """kv_rows int32 [2][R][smax]: permute the row table by src (call right after beam_step)."""
    _, R, smax = kv_rows.shape
    _check(lib().itts_beam_kv_rows(_p(kv_rows), _p(src), _p(state), R, smax, _stream()), "itts_beam_kv_rows")


def beam_reorder_kv(kc, vc, src, state, B, num_beams):
    """kc / vc: T [L][rows][H][smax][64]; rows of each batch element are permuted in place by src."""
    L, rows, H, smax, hd = kc.shape
    assert hd == 64 and rows >= B * num_beams
    _check(lib().itts_beam_reorder_kv(_p(kc), _p(vc), _p(src), _p(state), L, B, int(num_beams), H, smax, kc.stride(0),
                                      dt(kc.dtype), _stream()), "itts_beam_reorder_kv")


def tanh_pcm(x, wav=None, pcm=None, apply_tanh=True):
    _dev(x)
    _check(lib().itts_tanh_pcm(_p(x), _p(wav), _p(pcm), x.numel(), dt(x.dtype), int(apply_tanh), _stream()),
           "itts_tanh_pcm")
