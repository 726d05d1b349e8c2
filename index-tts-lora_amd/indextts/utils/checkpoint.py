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


def load_checkpoint(model, model_pth: str) -> dict:
    ckpt = torch.load(model_pth, map_location="cpu", weights_only=False)
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
