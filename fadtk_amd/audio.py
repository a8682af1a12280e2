"""Audio I/O and normalisation -- the step in front of the front end (fadtk/fad.py:139-186).

fadtk normalises every input file to mono 16-bit PCM WAV at the model's sample rate with torchaudio
(`Resample(lowpass_filter_width=64, rolloff=0.9476, resampling_method="sinc_interp_kaiser",
beta=14.77)`) and caches it under <dir>/convert/<sr>/.  torchaudio / soundfile are optional here:
WAV files are read with the stdlib, anything else needs soundfile or torchaudio to be installed.
The Kaiser-windowed sinc resampler below is the same published algorithm, written against plain
torch ops (conv1d); it is host-side plumbing, not one of the measured kernels.
"""
from __future__ import annotations

import math
import wave
from pathlib import Path
from typing import Tuple

import numpy as np

LOWPASS_FILTER_WIDTH = 64            # fad.py:154-157
ROLLOFF = 0.9475937167399596
BETA = 14.769656459379492


def read_audio(path) -> Tuple[np.ndarray, int]:
    """-> (float32 [channels, samples] in [-1, 1], sample_rate)."""
    path = Path(path)
    if path.suffix.lower() == ".wav":
        try:
            return _read_wav_stdlib(path)
        except (wave.Error, ValueError):
            pass
    try:
        import soundfile
        data, sr = soundfile.read(str(path), dtype="float32", always_2d=True)
        return np.ascontiguousarray(data.T), int(sr)
    except ImportError:
        pass
    try:
        import torchaudio
        x, sr = torchaudio.load(str(path))
        return x.numpy().astype(np.float32), int(sr)
    except ImportError:
        raise RuntimeError(f"cannot decode {path}: only PCM WAV is readable without soundfile/torchaudio installed")


def _read_wav_stdlib(path: Path) -> Tuple[np.ndarray, int]:
    with wave.open(str(path), "rb") as w:
        ch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    return np.ascontiguousarray(x.reshape(-1, ch).T), int(sr)


def read_pcm16(path) -> Tuple[np.ndarray, int]:
    """int16 samples of a PCM16 WAV (what `soundfile.read(dtype='int16')` returns at model_loader.py:64)."""
    with wave.open(str(path), "rb") as w:
        if w.getsampwidth() != 2:
            x, sr = read_audio(path)
            return np.clip(np.rint(x.mean(axis=0) * 32768.0), -32768, 32767).astype(np.int16), sr
        ch, sr, n = w.getnchannels(), w.getframerate(), w.getnframes()
        data = np.frombuffer(w.readframes(n), dtype="<i2")
    return (data.reshape(-1, ch)[:, 0] if ch > 1 else data).copy(), int(sr)


def write_pcm16(path, mono: np.ndarray, sr: int):
    """Mono float in [-1, 1] -> 16-bit PCM WAV (torchaudio.save(..., encoding='PCM_S', bits_per_sample=16))."""
    q = np.clip(np.rint(np.asarray(mono, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(q.tobytes())


def resample_kaiser(x: np.ndarray, orig_sr: int, new_sr: int, device=None) -> np.ndarray:
    """Kaiser-windowed sinc interpolation of a mono signal, parameters of fad.py:151-158."""
    if orig_sr == new_sr:
        return np.asarray(x, dtype=np.float32)
    import torch
    g = math.gcd(int(orig_sr), int(new_sr))
    orig, new = int(orig_sr) // g, int(new_sr) // g
    base = min(orig, new) * ROLLOFF
    width = math.ceil(LOWPASS_FILTER_WIDTH * orig / base)
    dev = device if device is not None else ("cuda" if torch.cuda.is_available() else "cpu")
    idx = torch.arange(-width, width + orig, dtype=torch.float64, device=dev)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64, device=dev)[:, None, None] / new + idx
    t = (t * base).clamp(-LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    beta = torch.tensor(BETA, dtype=torch.float64, device=dev)
    window = torch.i0(beta * torch.sqrt(1 - (t / LOWPASS_FILTER_WIDTH) ** 2)) / torch.i0(beta)
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), torch.sin(t) / t) * window * (base / orig)
    wav = torch.as_tensor(np.asarray(x, dtype=np.float32), device=dev)[None, None]
    length = wav.shape[-1]
    wav = torch.nn.functional.pad(wav, (width, width + orig))
    out = torch.nn.functional.conv1d(wav, kernel.to(torch.float32), stride=orig)      # [1, new, frames]
    out = out.transpose(1, 2).reshape(-1)
    return out[: math.ceil(new * length / orig)].cpu().numpy()


def convert_to_model_rate(src, dst, sr: int):
    """Decode, mix to mono, resample to ``sr`` and store as PCM16 WAV (fad.py:148-160)."""
    x, fs = read_audio(src)
    mono = x.mean(axis=0)
    write_pcm16(dst, resample_kaiser(mono, fs, sr), sr)
