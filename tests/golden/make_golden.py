#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (this container only).

    python tests/golden/make_golden.py [--only gpt_small,gpt_full,sampling,act1d,bigvgan]

The reference (/root/reference, read-only) is imported in place through tests/golden/_ref_harness.py; weights are
the name-hashed synthetic tensors of tests/synth.py at the REAL shapes (no checkpoints exist offline, SURVEY.md §0
item 10).  Only inputs/outputs are written (.npz); no reference source is copied.  The decode loop is driven
manually (prefill call with past=None, then one-token calls with the returned cache and a mask grown by one):
transformers 5.15's generate() never feeds the prefix to this model (SURVEY.md §8c "version hazard"), whereas the
manual drive reproduces the 4.44.2 control flow of indextts/gpt/model.py:125-205 exactly.
"""
import argparse
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import _ref_harness  # noqa: E402

_ref_harness.install()

import torch  # noqa: E402
import yaml  # noqa: E402

import synth  # noqa: E402

torch.set_grad_enabled(False)
torch.set_num_threads(8)

CFG = yaml.safe_load(open(os.path.join(_ref_harness.REFERENCE_ROOT, "finetune_models", "config.yaml")))


class AttrDict(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"  wrote {path}  ({os.path.getsize(path) / 1e6:.2f} MB)")


# ------------------------------------------------------------------------------------------------ inputs
def gpt_inputs():
    cond_mel = synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)
    cond_mel2 = synth.uniform("in.cond_mel2", (1, 100, 301), -6.0, 2.0)
    lens = [12, 9, 6]
    L = max(lens)
    text = np.full((3, L), 1, dtype=np.int64)  # right-padded with stop_text_token like pad_tokens_cat (infer.py:554-566)
    for i, n in enumerate(lens):
        u = synth.uniform(f"in.text{i}", (n,), 0.0, 1.0)
        text[i, :n] = 2 + np.floor(u * (12000 - 2)).astype(np.int64)
    return cond_mel, cond_mel2, text, lens


def build_gpt(layers):
    from indextts.gpt.model import UnifiedVoice

    g = dict(CFG["gpt"])
    g["layers"] = layers
    m = UnifiedVoice(**g).eval()
    sd = m.state_dict()
    new = synth.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, synth.gpt_param)
    missing = [k for k in sd if k not in new]
    assert all(k.endswith("pos_enc.pe") for k in missing), missing
    m.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=False)
    m.post_init_gpt2_config(use_deepspeed=False, kv_cache=True, half=False)
    return m


def penalised(logits, history, penalty):
    from transformers import RepetitionPenaltyLogitsProcessor

    return RepetitionPenaltyLogitsProcessor(penalty)(history, logits.clone())


def make_gpt(tag, layers, steps):
    print(f"[{tag}] layers={layers}")
    t0 = time.time()
    m = build_gpt(layers)
    cond_mel_np, cond_mel2_np, text_np, lens = gpt_inputs()
    cond_mel = torch.from_numpy(cond_mel_np)
    text = torch.from_numpy(text_np)
    cml = torch.tensor([cond_mel.shape[-1]])
    out = {}
    # conditioner (model.py:524-529)
    conds = m.get_conditioning(cond_mel, cml)
    out["conds"] = conds.numpy()
    cond_mel2 = torch.from_numpy(cond_mel2_np)
    out["conds2"] = m.get_conditioning(cond_mel2, torch.tensor([cond_mel2.shape[-1]])).numpy()
    # prefix (model.py:606-667)
    fake, emb, mask = m.prepare_gpt_inputs(conds, text)
    out["fake_inputs"] = fake.numpy()
    out["prefix_emb"] = emb.numpy()
    out["attention_mask"] = mask.numpy()
    im = m.inference_model
    im.store_mel_emb(emb)
    # manual cached drive, greedy with repetition_penalty=10 (tests/padding_test.py:35-46 settings)
    history = fake.clone()
    r = im(input_ids=fake, attention_mask=mask, past_key_values=None, use_cache=True, return_dict=True)
    past = r.past_key_values
    logits_all, codes = [], []
    cur_mask = mask
    for s in range(steps):
        logits = r.logits[:, -1, :].float()
        logits_all.append(logits.numpy().copy())
        nxt = penalised(logits, history, 10.0).argmax(-1)
        codes.append(nxt.numpy().copy())
        history = torch.cat([history, nxt[:, None]], dim=1)
        if s + 1 == steps:
            break
        cur_mask = torch.cat([cur_mask, torch.ones(cur_mask.shape[0], 1, dtype=cur_mask.dtype)], dim=1)
        r = im(input_ids=nxt[:, None], attention_mask=cur_mask, past_key_values=past, use_cache=True, return_dict=True)
        past = r.past_key_values
    out["logits"] = np.stack(logits_all, 0)  # [steps, 3, 8194]
    out["codes"] = np.stack(codes, 1)  # [3, steps]
    # same drive for row 0 alone (no left padding) -> batch/pad invariance reference (tests/padding_test.py:69-97)
    fake1, emb1, mask1 = m.prepare_gpt_inputs(conds, text[2:3, :lens[2]])
    im.store_mel_emb(emb1)
    r1 = im(input_ids=fake1, attention_mask=mask1, past_key_values=None, use_cache=True, return_dict=True)
    out["logits_row2_alone_step0"] = r1.logits[:, -1, :].float().numpy()
    # latent pass (model.py:548-597) for row 0 with the generated codes
    T = steps
    codes0 = torch.from_numpy(out["codes"][0:1, :T].copy())
    lat = m(cond_mel, text[0:1, :lens[0]], torch.tensor([lens[0]]), codes0, torch.tensor([T * 1024]),
            cond_mel_lengths=cml, return_latent=True)
    out["latent_row0"] = lat.numpy()
    out["text"] = text_np
    out["text_lens"] = np.array(lens)
    save(tag, **out)
    print(f"[{tag}] done in {time.time() - t0:.1f}s; logits abs max {np.abs(out['logits']).max():.3f} "
          f"latent rms {np.sqrt((out['latent_row0'] ** 2).mean()):.3f}")


# ------------------------------------------------------------------------------------------------ sampling
def make_sampling():
    from transformers import (RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper,
                              TopPLogitsWarper)

    print("[sampling]")
    B, V = 4, 8194
    logits = torch.from_numpy(synth.uniform("in.sample_logits", (B, V), -8.0, 8.0))
    # make a few near-ties and a dominant token to exercise top-p edges
    logits[1, 100] = 30.0
    logits[2, 7] = logits[2, 9] = 7.9990234375
    hist = torch.ones(B, 40, dtype=torch.long)
    hist[:, 30] = 8192
    gen = np.floor(synth.uniform("in.sample_hist", (B, 9), 0.0, 8192.0)).astype(np.int64)
    hist[:, 31:] = torch.from_numpy(gen)
    out = {"logits": logits.numpy(), "history": hist.numpy()}
    s = RepetitionPenaltyLogitsProcessor(10.0)(hist, logits.clone())
    out["after_penalty"] = s.numpy().copy()
    s = TemperatureLogitsWarper(0.8)(hist, s)
    out["after_temperature"] = s.numpy().copy()
    s = TopKLogitsWarper(30)(hist, s)
    out["after_topk"] = s.numpy().copy()
    s = TopPLogitsWarper(0.8)(hist, s)
    out["after_topp"] = s.numpy().copy()
    out["probs"] = torch.softmax(s, -1).numpy()
    save("sampling", **out)


# ------------------------------------------------------------------------------------------------ activation
def make_act1d():
    from indextts.BigVGAN.activations import SnakeBeta
    from indextts.BigVGAN.alias_free_torch import Activation1d

    print("[act1d]")
    out = {}
    for i, (B, C, T) in enumerate([(2, 24, 257), (1, 768, 8), (1, 48, 1), (1, 48, 5), (1, 96, 13), (2, 192, 64)]):
        act = Activation1d(activation=SnakeBeta(C, alpha_logscale=True))
        act.act.alpha.data = torch.from_numpy(0.5 * synth.uniform(f"act{i}.alpha", (C,)))
        act.act.beta.data = torch.from_numpy(0.5 * synth.uniform(f"act{i}.beta", (C,)))
        x = torch.from_numpy(synth.uniform(f"act{i}.x", (B, C, T), -3.0, 3.0))
        y = act(x)
        out[f"x{i}"] = x.numpy()
        out[f"alpha{i}"] = act.act.alpha.data.numpy()
        out[f"beta{i}"] = act.act.beta.data.numpy()
        out[f"y{i}"] = y.numpy()
        if i == 0:
            out["up_filter"] = act.upsample.filter.numpy().reshape(-1)
            out["down_filter"] = act.downsample.lowpass.filter.numpy().reshape(-1)
    save("act1d", **out)


# ------------------------------------------------------------------------------------------------ BigVGAN
def build_bigvgan():
    from indextts.BigVGAN.models import BigVGAN

    h = AttrDict(CFG["bigvgan"])
    g = BigVGAN(h, use_cuda_kernel=False)
    sd = g.state_dict()
    new = synth.fill_state_dict({k: tuple(v.shape) for k, v in sd.items()}, synth.bigvgan_param)
    missing = [k for k in sd if k not in new]
    assert all(k.endswith(".filter") for k in missing), missing
    g.load_state_dict({k: torch.from_numpy(v) for k, v in new.items()}, strict=False)
    g.remove_weight_norm()
    return g.eval()


def make_bigvgan():
    print("[bigvgan]")
    t0 = time.time()
    g = build_bigvgan()
    out = {}
    # -- T=4 with intermediates
    lat = torch.from_numpy(synth.uniform("in.latent4", (1, 4, 1280), -1.7, 1.7))
    mel = torch.from_numpy(synth.uniform("in.melref", (1, 120, 100), -6.0, 2.0))
    caps = {}
    hooks = []
    hooks.append(g.speaker_encoder.register_forward_hook(lambda m, i, o: caps.__setitem__("spk", o.detach().clone())))
    hooks.append(g.conv_pre.register_forward_hook(lambda m, i, o: caps.__setitem__("conv_pre", o.detach().clone())))
    for i in range(6):
        hooks.append(g.ups[i][0].register_forward_hook(
            lambda m, i_, o, i=i: caps.__setitem__(f"up{i}", o.detach().clone())))
    for j in range(18):
        hooks.append(g.resblocks[j].register_forward_hook(
            lambda m, i_, o, j=j: caps.__setitem__(f"rb{j}", o.detach().clone())))
    hooks.append(g.activation_post.register_forward_hook(
        lambda m, i_, o: caps.__setitem__("act_post", o.detach().clone())))
    wav, _ = g(lat, mel)
    for hk in hooks:
        hk.remove()
    out["latent4"] = lat.numpy()
    out["melref"] = mel.numpy()
    out["spk4"] = caps["spk"].numpy()
    out["conv_pre4"] = caps["conv_pre"].numpy()  # before + cond_layer(spk)
    for i in range(6):
        out[f"up{i}_4"] = caps[f"up{i}"].numpy()  # ConvTranspose1d output, before + conds[i](spk)
        xs = caps[f"rb{3 * i}"].clone()
        xs += caps[f"rb{3 * i + 1}"]
        xs += caps[f"rb{3 * i + 2}"]
        out[f"stage{i}_4"] = (xs / 3).numpy()
    out["rb0_4"] = caps["rb0"].numpy()
    out["act_post4"] = caps["act_post"].numpy()
    out["wav4"] = wav.numpy()
    print(f"  T=4: wav rms {wav.pow(2).mean().sqrt():.4f} absmax {wav.abs().max():.4f}; "
          + " ".join(f"s{i}={caps[f'rb{3*i}'].pow(2).mean().sqrt():.2f}" for i in range(6)))
    # -- T=8 final only
    lat8 = torch.from_numpy(synth.uniform("in.latent8", (1, 8, 1280), -1.7, 1.7))
    wav8, _ = g(lat8, mel)
    out["latent8"] = lat8.numpy()
    out["wav8"] = wav8.numpy()
    # -- batch 2, T=5, longer reference mel
    lat2 = torch.from_numpy(synth.uniform("in.latent_b2", (2, 5, 1280), -1.7, 1.7))
    mel2 = torch.from_numpy(synth.uniform("in.melref_b2", (2, 150, 100), -6.0, 2.0))
    wav2, _ = g(lat2, mel2)
    out["latent_b2"] = lat2.numpy()
    out["melref_b2"] = mel2.numpy()
    out["spk_b2"] = g.speaker_encoder(mel2).numpy()
    out["wav_b2"] = wav2.numpy()
    save("bigvgan", **out)
    print(f"[bigvgan] done in {time.time() - t0:.1f}s")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="gpt_small,gpt_full,sampling,act1d,bigvgan")
    args = ap.parse_args()
    todo = set(args.only.split(","))
    if "sampling" in todo:
        make_sampling()
    if "act1d" in todo:
        make_act1d()
    if "gpt_small" in todo:
        make_gpt("gpt_small", 2, 8)
    if "gpt_full" in todo:
        make_gpt("gpt_full", 24, 6)
    if "bigvgan" in todo:
        make_bigvgan()
