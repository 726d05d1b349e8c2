"""Minimal WAV I/O (PCM 8/16/24/32-bit and float32) so that the path does not depend on soundfile, which the reference
uses at infer.py:790 (sf.read) and :912 (sf.write PCM_16).  soundfile is used when it is installed."""
import struct
import wave

import numpy as np


def read_audio(path: str):
    """-> (float64 array [N] or [N, channels] in [-1,1), sample_rate), like soundfile.read."""
    try:
        import soundfile as sf  # noqa: WPS433
        return sf.read(path)
    except ImportError:
        pass
    try:
        with wave.open(path, "rb") as w:
            nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
            raw = w.readframes(n)
    except wave.Error:
        return _read_wav_float(path)
    if width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
    elif width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - 0x1000000, v)
        a = v.astype(np.float64) / 8388608.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    if nch > 1:
        a = a.reshape(-1, nch)
    return a, sr


def _read_wav_float(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file (compressed prompts need soundfile)")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif cid == b"data":
            payload = body
        pos += 8 + size + (size & 1)
    if fmt is None or payload is None or fmt[0] != 3 or fmt[5] != 32:
        raise ValueError(f"{path}: unsupported WAV encoding")
    a = np.frombuffer(payload, dtype="<f4").astype(np.float64)
    if fmt[1] > 1:
        a = a.reshape(-1, fmt[1])
    return a, fmt[2]


def write_pcm16(path: str, samples: np.ndarray, sample_rate: int):
    """samples int16 [N] or [N, channels] -> RIFF PCM_16 (what sf.write(..., subtype='PCM_16') produces)."""
    s = np.ascontiguousarray(samples, dtype="<i2")
    nch = 1 if s.ndim == 1 else s.shape[1]
    with wave.open(path, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(2)
        w.setframerate(int(sample_rate))
        w.writeframes(s.tobytes())
