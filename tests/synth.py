"""Deterministic, RNG-library-independent synthetic tensors for fixtures and tests.

Real IndexTTS checkpoints are not available offline, so every fixture and parity test uses
weights regenerated from a *name hash*: the same state-dict key always yields the same values, at any shape, on any
machine, without committing hundreds of MB.  Values come from numpy's Philox bit generator keyed by crc32(name)
(24-bit uniforms from its 32-bit draws), so they do not depend on torch's RNG implementation.
"""
from __future__ import annotations

import re
import zlib

import numpy as np


def uniform(name: str, shape, lo: float = -1.0, hi: float = 1.0) -> np.ndarray:
    """float32 array of `shape`, i.i.d. uniform in [lo, hi), fully determined by `name`."""
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    # Generator.random(float32) = (next_uint32 >> 8) * 2**-24 on the Philox stream keyed by the name hash.
    u = np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode("utf-8")))).random(n, dtype=np.float32)
    out = (lo + (hi - lo) * u).astype(np.float32)
    return out.reshape(tuple(shape))


def normal_like(name: str, shape, std: float) -> np.ndarray:
    """Zero-mean uniform with the requested standard deviation (uniform on [-std*sqrt3, std*sqrt3))."""
    a = float(std) * 3.0 ** 0.5
    return uniform(name, shape, -a, a)


_NORM_W = re.compile(r"(ln_1|ln_2|ln_f|final_norm|after_norm|norm_ff|norm_mha|norm_conv|norm_final|conv_module\.norm|"
                     r"norm\.norm)\.weight$")


def gpt_param(name: str, shape) -> np.ndarray:
    """Synthetic value for one UnifiedVoice state-dict entry (reference key names)."""
    shape = tuple(shape)
    if name.endswith("pos_enc.pe"):
        raise KeyError("pe is a deterministic buffer, not synthesised")
    if _NORM_W.search(name) or name.endswith("norm.gamma"):
        return 1.0 + 0.1 * uniform(name, shape)
    if name.endswith(".bias"):
        return 0.02 * uniform(name, shape)
    if name.endswith("pos_bias_u") or name.endswith("pos_bias_v"):
        return 0.1 * uniform(name, shape)
    if name.startswith("gpt.h.") and name.endswith(".weight"):
        return normal_like(name, shape, 0.02)  # GPT-2 style init
    if "embedding" in name and name.endswith(".weight"):
        return normal_like(name, shape, 0.02)
    if name == "perceiver_encoder.latents":
        return normal_like(name, shape, 0.02)
    if name.endswith(".weight"):
        # generic linear / conv: std = 1/sqrt(fan_in)
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return normal_like(name, shape, 1.0 / np.sqrt(max(fan_in, 1)))
    raise KeyError(f"no synthesis rule for {name} {shape}")


def bigvgan_param(name: str, shape) -> np.ndarray:
    """Synthetic value for one BigVGAN 'generator' state-dict entry in the *weight-norm parametrised* form."""
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith(".filter"):
        raise KeyError("filters are deterministic buffers, not synthesised")
    if name.endswith("act.alpha") or name.endswith("act.beta"):
        return 0.5 * uniform(name, shape)  # log-scale snake parameters (config: snake_logscale)
    if name.endswith("running_mean"):
        return 0.05 * uniform(name, shape)
    if name.endswith("running_var"):
        return 0.75 + 0.25 * uniform(name, shape)
    if name.endswith("norm.norm.weight"):
        return 1.0 + 0.1 * uniform(name, shape)
    if name.endswith(".bias"):
        return 0.02 * uniform(name, shape)
    if name.endswith("weight_v"):
        return uniform(name, shape)
    if name.endswith("weight_g"):
        if name.startswith("ups."):
            g = 1.2
        elif name.startswith("conv_pre"):
            g = 1.0
        elif name.startswith("conv_post"):
            g = 0.25
        else:
            g = 0.6
        return g * (1.0 + 0.1 * uniform(name, shape))
    if name.endswith(".weight"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        return normal_like(name, shape, 1.0 / np.sqrt(max(fan_in, 1)))
    raise KeyError(f"no synthesis rule for {name} {shape}")


def fill_state_dict(sd_shapes: dict, rule) -> dict:
    """{name: shape} -> {name: np.ndarray} using `rule`, skipping entries the rule declines (deterministic buffers)."""
    out = {}
    for k, shp in sd_shapes.items():
        try:
            out[k] = rule(k, shp)
        except KeyError:
            pass
    return out
