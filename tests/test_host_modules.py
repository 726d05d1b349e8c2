"""Host-side (PyTorch) product modules vs reference-run fixtures: Conformer+Perceiver conditioner, ECAPA-TDNN."""
import json
import os

import numpy as np
import torch

import synth
from indextts.BigVGAN.ECAPA_TDNN import ecapa_embed
from indextts.gpt.conformer_encoder import conformer_encode
from indextts.gpt.perceiver import perceiver_resample

torch.set_grad_enabled(False)
G = os.path.join(os.path.dirname(__file__), "golden")


def cond_weights():
    shapes = json.load(open(os.path.join(G, "gpt_cond_shapes.json")))
    shapes = {k: tuple(v) for k, v in shapes.items() if k.startswith(("conditioning_encoder.", "perceiver_encoder."))}
    return {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, synth.gpt_param).items()}


def conditioning(W, mel):
    x, mask = conformer_encode(W, mel.transpose(1, 2), torch.tensor([mel.shape[-1]]))
    cm = torch.nn.functional.pad(mask.squeeze(1), (32, 0), value=True)
    return perceiver_resample(W, x, cm)


def test_conditioner_matches_reference():
    g = np.load(os.path.join(G, "gpt_small.npz"))
    W = cond_weights()
    c = conditioning(W, torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)))
    np.testing.assert_allclose(c.numpy(), g["conds"], atol=2e-4)
    c2 = conditioning(W, torch.from_numpy(synth.uniform("in.cond_mel2", (1, 100, 301), -6.0, 2.0)))
    np.testing.assert_allclose(c2.numpy(), g["conds2"], atol=2e-4)


def test_conditioner_without_padding_skips_the_masks_and_changes_no_bit():
    """lengths = None (every row as long as the tensor: the one-prompt case) leaves the identity mask operations out and takes
    the position projections from the per-weights cache: same bits as the general form, also on the second (cached) call and
    for a batch of two prompts of one length."""
    W = cond_weights()
    for name, shape in (("in.cond_mel", (1, 100, 120)), ("in.cond_mel2", (1, 100, 301)), ("in.cond_mel_b2", (2, 100, 77))):
        mel = torch.from_numpy(synth.uniform(name, shape, -6.0, 2.0))
        want = conformer_encode(dict(W), mel.transpose(1, 2), torch.full((shape[0],), shape[2]))
        cm = torch.nn.functional.pad(want[1].squeeze(1), (32, 0), value=True)
        want_c = perceiver_resample(W, want[0], cm)
        for _ in range(2):
            x, mask = conformer_encode(W, mel.transpose(1, 2), None)
            assert mask is None and torch.equal(x, want[0])
            assert torch.equal(perceiver_resample(W, x, None), want_c)
    assert len(W[("const", "conditioning_encoder.")]) == 3      # one entry per subsampled length


def test_ecapa_matches_reference():
    g = np.load(os.path.join(G, "bigvgan.npz"))
    shapes = json.load(open(os.path.join(G, "bigvgan_shapes.json")))
    shapes = {k: tuple(v) for k, v in shapes.items() if k.startswith("speaker_encoder.")}
    W = {k: torch.from_numpy(v) for k, v in synth.fill_state_dict(shapes, synth.bigvgan_param).items()}
    e = ecapa_embed(W, torch.from_numpy(g["melref"]))
    np.testing.assert_allclose(e.numpy(), g["spk4"], atol=2e-5)
    e2 = ecapa_embed(W, torch.from_numpy(g["melref_b2"]))
    np.testing.assert_allclose(e2.numpy(), g["spk_b2"], atol=2e-5)
