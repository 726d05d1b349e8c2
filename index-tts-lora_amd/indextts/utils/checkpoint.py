"""Checkpoint loading with the reference's file formats (indextts/utils/checkpoint.py:23-89).

`gpt.pth` is a torch.save pickle of either a bare state dict or {'model': sd, ['speaker_conditions': {id: np[32,D]}],
['speakers': [...]]}; tensors may be fp32/fp16/bf16; keys are loaded non-strictly.  A sibling .yaml, if present, is
returned as the config dict (with 'speakers' merged in)."""
import logging
import os
import re

import torch
import yaml

logger = logging.getLogger("indextts")


def _safe_load(model_pth: str):
    """torch.load restricted to tensors, containers and the numpy arrays of `speaker_conditions` (weights_only unpickler
    with an allowlist): a checkpoint file cannot run code.  ITTS_TRUST_CHECKPOINTS=1 restores the reference's unrestricted
    pickle load (checkpoint.py:25) for files that carry other Python objects and that the operator trusts."""
    if os.environ.get("ITTS_TRUST_CHECKPOINTS") == "1":
        return torch.load(model_pth, map_location="cpu", weights_only=False)
    import numpy as np
    allow = [np.ndarray, np.dtype]
    for mod, name in (("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "_reconstruct"),
                      ("numpy._core.multiarray", "scalar"), ("numpy.core.multiarray", "scalar")):
        try:
            allow.append(getattr(__import__(mod, fromlist=[name]), name))
        except Exception:  # noqa: BLE001  (module layout differs between numpy 1.x and 2.x)
            pass
    for name in ("Float32DType", "Float64DType", "Float16DType", "Int64DType", "Int32DType"):
        t = getattr(getattr(np, "dtypes", None), name, None)
        if t is not None:
            allow.append(t)
    try:
        with torch.serialization.safe_globals(allow):
            return torch.load(model_pth, map_location="cpu", weights_only=True)
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"{model_pth}: refused by the restricted checkpoint loader ({e}); set ITTS_TRUST_CHECKPOINTS=1 "
                           f"to load a trusted file with the unrestricted pickle loader") from e


def load_checkpoint(model, model_pth: str) -> dict:
    ckpt = _safe_load(model_pth)
    if isinstance(ckpt, dict) and "speaker_conditions" in ckpt:
        for sid, arr in ckpt["speaker_conditions"].items():
            t = torch.as_tensor(arr).float()
            if t.dim() == 2:
                t = t.unsqueeze(0)
            setattr(model, f"mean_condition_{sid}", t)
        logger.info("loaded %d speaker conditions", len(ckpt["speaker_conditions"]))
    sd = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    if "mean_condition" in sd and hasattr(model, "mean_condition"):
        model.mean_condition = sd["mean_condition"]
        sd = {k: v for k, v in sd.items() if k != "mean_condition"}
    model.load_state_dict(sd, strict=False)
    cfg = {}
    info = re.sub(r"\.(pth|pt)$", ".yaml", model_pth)
    if info != model_pth and os.path.exists(info):
        with open(info, "r") as f:
            cfg = yaml.safe_load(f) or {}
    if isinstance(ckpt, dict) and "speakers" in ckpt:
        cfg["speakers"] = ckpt["speakers"]
    return cfg
