"""Synthetic full-size state dicts in the reference checkpoint key format (see tests/synth.py)."""
import json
import os

import torch

import synth

G = os.path.join(os.path.dirname(__file__), "golden")
D = 1280


def gpt_layer_shapes(layers):
    s = {}
    for i in range(layers):
        p = f"gpt.h.{i}."
        s.update({p + "ln_1.weight": (D,), p + "ln_1.bias": (D,), p + "attn.c_attn.weight": (D, 3 * D),
                  p + "attn.c_attn.bias": (3 * D,), p + "attn.c_proj.weight": (D, D), p + "attn.c_proj.bias": (D,),
                  p + "ln_2.weight": (D,), p + "ln_2.bias": (D,), p + "mlp.c_fc.weight": (D, 4 * D),
                  p + "mlp.c_fc.bias": (4 * D,), p + "mlp.c_proj.weight": (4 * D, D), p + "mlp.c_proj.bias": (D,)})
    return s


def gpt_state_dict(layers, with_conditioner=True):
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(G, "gpt_cond_shapes.json"))).items()}
    if not with_conditioner:
        shapes = {k: v for k, v in shapes.items() if not k.startswith(("conditioning_encoder.", "perceiver_encoder."))}
    shapes = {k: v for k, v in shapes.items() if not k.startswith("text_head.")}
    shapes.update(gpt_layer_shapes(layers))
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, synth.gpt_param).items()}


def bigvgan_state_dict():
    shapes = {k: tuple(v) for k, v in json.load(open(os.path.join(G, "bigvgan_shapes.json"))).items()}
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, synth.bigvgan_param).items()}


def reference_config():
    """The model section of finetune_models/config.yaml (values only; SURVEY.md §0 item 2, config.yaml:52-107)."""
    return {
        "version": 1.5,
        "dataset": {"bpe_model": "bpe.model", "sample_rate": 24000},
        "gpt": {"model_dim": 1280, "max_mel_tokens": 800, "max_text_tokens": 600, "heads": 20,
                "use_mel_codes_as_input": True, "mel_length_compression": 1024, "layers": 24,
                "number_text_tokens": 12000, "number_mel_codes": 8194, "start_mel_token": 8192, "stop_mel_token": 8193,
                "start_text_token": 0, "stop_text_token": 1, "train_solo_embeddings": False,
                "condition_type": "conformer_perceiver",
                "condition_module": {"output_size": 512, "linear_units": 2048, "attention_heads": 8, "num_blocks": 6,
                                     "input_layer": "conv2d2", "perceiver_mult": 2}},
        "bigvgan": {"resblock": "1", "upsample_rates": [4, 4, 4, 4, 2, 2], "upsample_kernel_sizes": [8, 8, 4, 4, 4, 4],
                    "upsample_initial_channel": 1536, "resblock_kernel_sizes": [3, 7, 11],
                    "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "feat_upsample": False,
                    "speaker_embedding_dim": 512, "cond_d_vector_in_each_upsampling_layer": True, "gpt_dim": 1280,
                    "activation": "snakebeta", "snake_logscale": True, "num_mels": 100},
        "gpt_checkpoint": "gpt.pth", "bigvgan_checkpoint": "bigvgan_generator.pth",
    }
