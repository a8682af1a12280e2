"""Audio I/O and normalisation -- the step in front of the front end (fadtk/fad.py:139-186).

fadtk normalises every input file to mono 16-bit PCM WAV at the model's sample rate with torchaudio
(`Resample(lowpass_filter_width=64, rolloff=0.9476, resampling_method="sinc_interp_kaiser",
beta=14.77)`) and caches it under <dir>/convert/<sr>/.  torchaudio / soundfile are optional here:
WAV files are read with the stdlib, anything else needs soundfile or torchaudio to be installed.
The Kaiser-windowed sinc resampler is the same published algorithm as a HIP kernel (csrc/resample.hip,
`fad_resample_kaiser`); this module is the file I/O around it.
"""
from __future__ import annotations

import wave
from pathlib import Path
from typing import Tuple

import numpy as np



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


def resample_kaiser(x: np.ndarray, orig_sr: int, new_sr: int, quantize_pcm16: bool = False, device: int = 0) -> np.ndarray:
    """Kaiser-windowed sinc interpolation of a mono signal with the parameters of fad.py:151-158, on the GPU
    (`fad_resample_kaiser`, csrc/resample.hip).  ``quantize_pcm16`` adds the 16-bit round trip of the cache file."""
    from . import hip
    return hip.resample_kaiser(np.asarray(x, dtype=np.float32), orig_sr, new_sr, quantize_pcm16=quantize_pcm16, device=device)


def convert_to_model_rate(src, dst, sr: int, device: int = 0):
    """Decode, mix to mono, resample to ``sr`` (on GPU ``device``) and store as PCM16 WAV (fad.py:148-160)."""
    src = Path(src)
    if src.suffix.lower() == ".wav":
        # a mono 16-bit PCM WAV already at the model's rate: decode -> float -> quantise gives back the very samples (x / 32768 * 32768 is exact
        # in float32), so the cache file is the input's frames under a fresh header -- no float round trip, no GPU call (config 2: every file)
        try:
            with wave.open(str(src), "rb") as w:
                if w.getsampwidth() == 2 and w.getnchannels() == 1 and w.getframerate() == int(sr) and w.getcomptype() == "NONE":
                    raw = w.readframes(w.getnframes())
                    dst = Path(dst)
                    dst.parent.mkdir(parents=True, exist_ok=True)
                    with wave.open(str(dst), "wb") as o:
                        o.setnchannels(1); o.setsampwidth(2); o.setframerate(int(sr))
                        o.writeframes(raw)
                    return
        except (wave.Error, ValueError, EOFError):
            pass
    x, fs = read_audio(src)
    mono = x.mean(axis=0)
    write_pcm16(dst, resample_kaiser(mono, fs, sr, device=device), sr)
