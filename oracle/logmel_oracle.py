"""CPU oracle for the log-mel front ends -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

float64 numpy restatements of the feature extraction that fadtk's loaders call inside third-party
packages which are NOT vendored under the reference tree:

  vggish_examples   fadtk/model_loader.py:99,108 -> torch.hub 'harritaylor/torchvggish' (unpinned):
                    AudioSet `vggish_input.waveform_to_examples` / `mel_features.log_mel_spectrogram`.
                    PARITY UNPINNED: that package is not installed here and the reference holds no
                    test vector at this boundary; the parameters are the published AudioSet ones.
  whisper_features  fadtk/model_loader.py:661,666 -> transformers 4.52.3 WhisperFeatureExtractor
                    (`_np_extract_fbank_features`).  PINNED against the installed transformers
                    extractor: tests/golden/g9_frontend.npz (make_frontend_golden.py).
  htsat_logmel      fadtk/model_loader.py:385,406 -> laion-clap 1.1.7 HTSAT: torchlibrosa 0.1.0
                    Spectrogram(n_fft=1024, hop=480, hann, center, reflect, power=2) +
                    LogmelFilterBank(sr=48000, n_mels=64, fmin=50, fmax=14000, ref=1, amin=1e-10,
                    top_db=None).  PINNED (same arithmetic) against transformers' ClapFeatureExtractor
                    slaney path (`mel_filters_slaney`, log_mel="dB"): tests/golden/g9_frontend.npz.
"""
from __future__ import annotations

import numpy as np


def _periodic_hann(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _frames(x: np.ndarray, win: int, hop: int) -> np.ndarray:
    n = 1 + (len(x) - win) // hop
    idx = np.arange(win)[None, :] + hop * np.arange(n)[:, None]
    return x[idx]


# ------------------------------------------------------------------ mel filter banks
def _hz_to_mel_htk(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_htk_vggish(bins=257, nmel=64, sr=16000.0, fmin=125.0, fmax=7500.0) -> np.ndarray:
    """AudioSet mel_features.spectrogram_to_mel_matrix: triangles in the HTK-mel domain, DC bin zeroed."""
    spec_mel = _hz_to_mel_htk(np.linspace(0.0, sr / 2.0, bins))
    edges = np.linspace(_hz_to_mel_htk(fmin), _hz_to_mel_htk(fmax), nmel + 2)
    w = np.empty((bins, nmel))
    for i in range(nmel):
        lo, ce, up = edges[i:i + 3]
        w[:, i] = np.maximum(0.0, np.minimum((spec_mel - lo) / (ce - lo), (up - spec_mel) / (up - ce)))
    w[0, :] = 0.0
    return w


def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp(np.log(6.4) / 27.0 * (m - 15.0)), 200.0 * m / 3.0)


def mel_slaney(bins, nmel, sr, fmin, fmax) -> np.ndarray:
    """librosa.filters.mel(htk=False, norm='slaney') == transformers mel_filter_bank(norm/mel_scale='slaney')."""
    ff = _mel_to_hz_slaney(np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), nmel + 2))
    fft = np.linspace(0.0, sr / 2.0, bins)
    slopes = ff[None, :] - fft[:, None]
    down = -slopes[:, :-2] / np.diff(ff)[:-1]
    up = slopes[:, 2:] / np.diff(ff)[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    return fb * (2.0 / (ff[2:nmel + 2] - ff[:nmel]))


# ------------------------------------------------------------------ the three front ends
def vggish_num_examples(n_samples: int) -> int:
    if n_samples < 400:
        return 0
    frames = 1 + (n_samples - 400) // 160
    return 0 if frames < 96 else 1 + (frames - 96) // 96


def vggish_examples(wav: np.ndarray) -> np.ndarray:
    """16 kHz mono in [-1, 1] -> [n_examples, 96, 64] log-mel patches."""
    x = np.asarray(wav, dtype=np.float64)
    n_ex = vggish_num_examples(len(x))
    if n_ex == 0:
        return np.zeros((0, 96, 64))
    fr = _frames(x, 400, 160) * _periodic_hann(400)
    mag = np.abs(np.fft.rfft(fr, 512, axis=1))
    logmel = np.log(mag @ mel_htk_vggish() + 0.01)
    return logmel[:n_ex * 96].reshape(n_ex, 96, 64)


def _centered_power(x: np.ndarray, nfft: int, hop: int) -> np.ndarray:
    xp = np.pad(x, nfft // 2, mode="reflect")
    fr = _frames(xp, nfft, hop) * _periodic_hann(nfft)
    return np.abs(np.fft.rfft(fr, nfft, axis=1)) ** 2


def whisper_features(wav: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """16 kHz mono -> [n_mels, 3000]: pad/cut to 30 s, power STFT 400/160, Slaney mels, log10, clamp, scale."""
    x = np.asarray(wav, dtype=np.float64)[:480000]
    x = np.pad(x, (0, 480000 - len(x)))
    power = _centered_power(x, 400, 160)[:-1]                       # 3001 frames, last one dropped
    mel = power @ mel_slaney(201, n_mels, 16000.0, 0.0, 8000.0)
    logs = np.log10(np.maximum(mel, 1e-10))
    logs = np.maximum(logs, logs.max() - 8.0)
    return ((logs + 4.0) / 4.0).T


def htsat_logmel(wav: np.ndarray) -> np.ndarray:
    """48 kHz mono -> [1 + n/480, 64]: power STFT 1024/480 centred, 64 Slaney mels 50-14000 Hz, 10 log10."""
    x = np.asarray(wav, dtype=np.float64)
    power = _centered_power(x, 1024, 480)
    mel = power @ mel_slaney(513, 64, 48000.0, 50.0, 14000.0)
    return 10.0 * np.log10(np.maximum(mel, 1e-10))
