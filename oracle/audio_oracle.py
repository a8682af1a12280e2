"""CPU oracle for the audio normalisation step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 numpy restatement of the resampling that FrechetAudioDistance.load_audio delegates to torchaudio
(fadtk/fad.py:151-159):

    torchaudio.transforms.Resample(fs, model_sr, lowpass_filter_width=64, rolloff=0.9475937167399596,
                                   resampling_method="sinc_interp_kaiser", beta=14.769656459379492)

followed by the 16-bit PCM cache file (fad.py:160 `torchaudio.save(..., encoding="PCM_S", bits_per_sample=16)`,
read back as int16 / 32768 at model_loader.py:64).

PARITY UNPINNED: torchaudio (2.7.0 in the reference's lock file) is a third-party dependency that is not under
/root/reference and not installed here, and the reference holds no test vector at this boundary.  What follows is the
published algorithm of torchaudio.functional.resample (`_get_sinc_resample_kernel`, `_apply_sinc_resample_kernel`): the
filter table is evaluated in float64 and rounded to float32 as torchaudio does; the convolution is accumulated in
float64 here (torchaudio: float32 conv1d), which is what makes this the tighter side of a comparison.
"""
from __future__ import annotations

import math

import numpy as np

LOWPASS_FILTER_WIDTH = 64
ROLLOFF = 0.9475937167399596
BETA = 14.769656459379492


def sinc_kernel(orig_sr: int, new_sr: int):
    """-> (kernel float32 [new, 2*width + orig], width, orig, new) with orig/new the rates over their gcd."""
    g = math.gcd(int(orig_sr), int(new_sr))
    orig, new = int(orig_sr) // g, int(new_sr) // g
    base = min(orig, new) * ROLLOFF
    width = math.ceil(LOWPASS_FILTER_WIDTH * orig / base)
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -LOWPASS_FILTER_WIDTH, LOWPASS_FILTER_WIDTH)
    window = np.i0(BETA * np.sqrt(1.0 - (t / LOWPASS_FILTER_WIDTH) ** 2)) / np.i0(BETA)
    t = t * np.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(t == 0, 1.0, np.sin(t) / t)
    return (sinc * window * (base / orig)).astype(np.float32), width, orig, new


def resample_kaiser(x: np.ndarray, orig_sr: int, new_sr: int) -> np.ndarray:
    """Mono signal -> float64 samples at ``new_sr``; length ceil(new * n / orig)."""
    x = np.asarray(x, dtype=np.float32).astype(np.float64).reshape(-1)
    if int(orig_sr) == int(new_sr):
        return x
    k, width, orig, new = sinc_kernel(orig_sr, new_sr)
    n = len(x)
    padded = np.concatenate([np.zeros(width), x, np.zeros(width + orig)])
    taps = k.shape[1]
    frames = (len(padded) - taps) // orig + 1
    idx = np.arange(taps)[None, :] + orig * np.arange(frames)[:, None]
    out = padded[idx] @ k.astype(np.float64).T                 # [frames, new]
    return out.reshape(-1)[: math.ceil(new * n / orig)]


def pcm16_roundtrip(y: np.ndarray) -> np.ndarray:
    """What survives the reference's 16-bit cache file: clip(rint(y * 32768)) / 32768."""
    return np.clip(np.rint(np.asarray(y, dtype=np.float64) * 32768.0), -32768, 32767) / 32768.0
