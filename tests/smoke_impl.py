"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle.

GPT: 2-layer model at the real width (d=1280, 20 heads), batch of 3 left-padded rows -> prefill logits + 3 cached
decode steps vs oracle/gpt_ref.py.  Vocoder: 2 latent frames through the full-size BigVGAN vs oracle/bigvgan_ref.py."""
import numpy as np
import torch


def run_smoke():
    import synth
    import weights
    from indextts.BigVGAN.models import BigVGAN
    from indextts.gpt.model import UnifiedVoice
    from indextts.utils.config import Config
    from oracle import bigvgan_ref, gpt_ref, sampling_ref

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = "cuda:0"
    torch.set_grad_enabled(False)
    cfg = weights.reference_config()
    sd = weights.gpt_state_dict(2)
    m = UnifiedVoice(**dict(cfg["gpt"], layers=2))
    m.load_state_dict(sd)
    m.to(dev).to(torch.float32).post_init_gpt2_config(kv_cache=True)
    cond_mel = torch.from_numpy(synth.uniform("in.cond_mel", (1, 100, 120), -6.0, 2.0)).to(dev)
    text = torch.tensor([[11, 22, 33, 44, 55, 66], [77, 88, 99, 1, 1, 1], [5, 6, 7, 8, 1, 1]], device=dev)
    conds = m.get_conditioning(cond_mel, None)
    _, emb, mask = m.prepare_gpt_inputs(conds, text)
    codes, logits = m.inference_speech(cond_mel, text, do_sample=False, num_beams=1, repetition_penalty=10.0,
                                       max_generate_length=4, return_logits=True)
    W = {k: v.float() for k, v in sd.items()}
    emb_o, mask_o, _ = gpt_ref.prepare_gpt_inputs(conds.cpu(), text.cpu(), W)
    assert (emb_o - emb.cpu()).abs().max().item() < 1e-5
    lg, past = gpt_ref.decode_prefill(emb_o, mask_o, W)
    errs = [(lg - logits[0].cpu()).abs().max().item()]
    hist = np.ones((3, emb_o.shape[1] + 1), dtype=np.int64)
    hist[:, -1] = 8192
    for s in range(1, 4):
        tok = codes[:, s - 1].cpu()
        mask_o = torch.cat([mask_o, torch.ones(3, 1, dtype=torch.bool)], 1)
        lg, past = gpt_ref.decode_step(tok, s, mask_o, past, W)
        errs.append((lg - logits[s].cpu()).abs().max().item())
    assert max(errs) < 1e-3, f"GPT logits vs oracle: {errs}"
    print(f"[smoke] GPT decode logits max-abs err vs oracle: {max(errs):.2e}; codes {codes.cpu().tolist()}")

    bsd = weights.bigvgan_state_dict()
    v = BigVGAN(Config(cfg["bigvgan"]))
    v.load_state_dict(bsd)
    v.to(dev).to(torch.float32).remove_weight_norm()
    lat = torch.from_numpy(synth.uniform("smoke.latent", (1, 2, 1280), -1.7, 1.7))
    mel = torch.from_numpy(synth.uniform("in.melref", (1, 120, 100), -6.0, 2.0))
    wav, _ = v(lat.to(dev), mel.to(dev))
    spk = v.speaker_embedding(mel.to(dev)).cpu()
    ref = bigvgan_ref.forward(lat, spk.transpose(1, 2), bigvgan_ref.Weights({k: t.numpy() for k, t in bsd.items()}))
    rms = (wav.cpu() - ref).pow(2).mean().sqrt().item()
    assert rms < 1e-4, f"waveform RMS vs oracle {rms}"
    print(f"[smoke] BigVGAN waveform RMS err vs oracle: {rms:.2e} (signal rms {ref.pow(2).mean().sqrt().item():.3f})")
    print("[smoke] ok")
