"""Prompt front-end: sinc resampling + log-mel spectrogram, in plain torch (torchaudio is not required).

Restates the torchaudio transforms the reference calls: `torchaudio.transforms.Resample(sr, 24000)` (infer.py:795;
sinc_interp_hann, lowpass_filter_width=6, rolloff=0.99) and `MelSpectrogram(sr=24000, n_fft=1024, hop=256, n_mels=100,
power=1, center=True, pad_mode='reflect', hann window, HTK mel scale, no norm)` followed by log(clip(., 1e-7))
(indextts/utils/feature_extractors.py:43-67, utils/common.py:116-129).  Parity note: torchaudio is absent from this
image, so these two functions are pinned by self-consistency tests only."""
import math

import torch
import torch.nn.functional as F


def resample(wave: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """wave [..., N] -> [..., ceil(N*new/orig)]."""
    if orig_freq == new_freq:
        return wave
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kern = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    kern = kern.to(wave.dtype).to(wave.device)
    shape = wave.shape
    x = wave.reshape(-1, 1, shape[-1])
    x = F.pad(x, (width, width + orig))
    y = F.conv1d(x, kern, stride=orig)  # [n, new, frames]
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = int(math.ceil(new * shape[-1] / orig))
    return y[..., :target].reshape(*shape[:-1], target)


def mel_filterbank(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> torch.Tensor:
    """HTK mel triangles, no area normalisation -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]
    down = -slopes[:, :-2] / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0)


class MelSpectrogramFeatures:
    def __init__(self, sample_rate=24000, n_fft=1024, hop_length=256, win_length=None, n_mels=100, mel_fmin=0,
                 mel_fmax=None, normalize=False, padding="center"):
        if padding not in ("center", "same"):
            raise ValueError("padding must be 'center' or 'same'")
        self.padding = padding
        self.n_fft, self.hop = n_fft, hop_length
        self.win_length = win_length or n_fft
        self.normalized = normalize
        self.fb = mel_filterbank(n_fft // 2 + 1, float(mel_fmin), float(mel_fmax or sample_rate // 2), n_mels, sample_rate)

    def __call__(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [B, N] -> log-mel [B, n_mels, frames]."""
        if self.padding == "same":
            pad = self.win_length - self.hop
            audio = F.pad(audio[:, None], (pad // 2, pad // 2), mode="reflect")[:, 0]
        win = torch.hann_window(self.win_length, periodic=True, dtype=audio.dtype, device=audio.device)
        spec = torch.stft(audio, self.n_fft, hop_length=self.hop, win_length=self.win_length, window=win,
                          center=self.padding == "center", pad_mode="reflect", normalized=False, onesided=True,
                          return_complex=True).abs()
        if self.normalized:
            spec = spec / win.pow(2).sum().sqrt()
        mel = torch.matmul(self.fb.to(spec.device, spec.dtype).t(), spec)
        return torch.log(torch.clip(mel, min=1e-7))

    forward = __call__
