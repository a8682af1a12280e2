"""Deterministic synthetic inputs shared by the golden-vector generator and the tests.

Everything is drawn from ``numpy.random.default_rng(seed)`` (PCG64; the stream of
``standard_normal`` / ``random`` is stable across numpy releases), so a fixture only has to
store the *expected outputs* of the reference plus a checksum of the inputs.

Shapes follow BASELINE.json ``configs`` / SURVEY.md section 8(d):
  C1  N=1024   D=128   A ~ N(0,1),  B ~ 1.02 N(0,1) + 0.01   float16
  C3  N=100000 D=512   same recipe, seeds (10, 11)
  C5  D=768 baseline, songs of 2 frames each
"""
from __future__ import annotations

import numpy as np


def normal_rows(seed: int, n: int, d: int, scale: float = 1.0, shift=0.0, dtype=np.float16) -> np.ndarray:
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    if scale != 1.0:
        x = scale * x
    return (x + shift).astype(dtype)


def c1_pair(dtype=np.float16):
    return normal_rows(0, 1024, 128, dtype=dtype), normal_rows(1, 1024, 128, 1.02, 0.01, dtype=dtype)


def c3_pair(n: int = 100_000, d: int = 512, dtype=np.float16):
    return normal_rows(10, n, d, dtype=dtype), normal_rows(11, n, d, 1.02, 0.01, dtype=dtype)


def shifted_pair(n: int = 4096, d: int = 128, dtype=np.float16):
    """|mu|/sigma ~ 7 with a small FAD: the case where the float16 mean (Q1) matters most."""
    return (normal_rows(20, n, d, 1.0, 7.0, dtype=dtype),
            normal_rows(21, n, d, 1.01, 7.01, dtype=dtype))


def orthobasis(seed: int, d: int) -> np.ndarray:
    rng = np.random.default_rng(seed)
    q, r = np.linalg.qr(rng.standard_normal((d, d)))
    return q * np.sign(np.diag(r))          # sign-fixed so LAPACK version does not matter


def decaying_rows(seed: int, n: int, d: int, basis_seed: int, power: float = 1.5, gain: float = 1.0,
                  dtype=np.float32) -> np.ndarray:
    """Rows with covariance spectrum lambda_k ~ k^-power in a random orthonormal basis."""
    rng = np.random.default_rng(seed)
    lam = np.arange(1, d + 1, dtype=np.float64) ** (-power)
    x = rng.standard_normal((n, d)) * np.sqrt(lam) * gain
    return (x @ orthobasis(basis_seed, d).T).astype(dtype)


def ragged_files(seed: int, n_files: int, d: int, dtype=np.float16, min_rows: int = 2, max_rows: int = 40):
    """Unequal per-file frame matrices (always includes one 2-row file), with non-zero means."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(min_rows, max_rows + 1, size=n_files)
    sizes[n_files // 2] = 2
    shift = rng.standard_normal(d) * 0.5
    out = []
    for k, n in enumerate(sizes):
        gain = 0.5 + rng.random()
        out.append((gain * rng.standard_normal((int(n), d)) + shift + 0.1 * k / n_files).astype(dtype))
    return out


def shifted_files(seed: int, n_files: int, d: int, dtype=np.float16, min_rows: int = 600, max_rows: int = 3000, shift: float = 7.0):
    """Long per-file frame matrices with |mu| / sigma ~ 7: np.mean's float32 running sum (utils.py:16, fad.py:48) ends ~1e-6 off the exact
    column sums of a few thousand such rows, and the float16 mean differs from the rounded exact one in some dimensions."""
    rng = np.random.default_rng(seed)
    sizes = rng.integers(min_rows, max_rows + 1, size=n_files)
    out = []
    for k, n in enumerate(sizes):
        gain = 0.8 + 0.4 * rng.random()
        out.append((gain * rng.standard_normal((int(n), d)) + shift + 0.02 * rng.standard_normal(d)).astype(dtype))
    return out


def baseline_stats(seed: int, n: int, d: int):
    """A full-rank float64 baseline (mu, Sigma) -- stand-in for the missing fma_pop.npz."""
    rng = np.random.default_rng(seed)
    gains = 1.0 + 0.5 * rng.random(d)
    x = rng.standard_normal((n, d)) * gains + 0.05 * rng.standard_normal(d)
    return x.mean(axis=0), np.cov(x, rowvar=False)


def songs(seed: int, n_songs: int, rows_per_song, d: int, dtype=np.float16):
    """Per-song frame matrices; ``rows_per_song`` is an int or a sequence cycled over songs."""
    rng = np.random.default_rng(seed)
    if np.isscalar(rows_per_song):
        rows_per_song = [int(rows_per_song)]
    out = []
    for k in range(n_songs):
        n = int(rows_per_song[k % len(rows_per_song)])
        gain = 0.7 + 0.6 * rng.random()
        out.append((gain * rng.standard_normal((n, d)) + 0.1 * rng.standard_normal(d)).astype(dtype))
    return out


def checksum(a: np.ndarray) -> float:
    """Order-independent fingerprint of an input array (guards against RNG drift)."""
    a64 = np.asarray(a, dtype=np.float64)
    return float(a64.sum() + 3.0 * np.abs(a64).sum())


def audio_clip(seed: int, n: int, sr: int) -> np.ndarray:
    """Synthetic mono audio: white noise plus three sinusoids, float32 in about [-0.6, 0.6]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    x = 0.1 * rng.standard_normal(n)
    for f, a in ((220.0, 0.2), (1760.0, 0.1), (5200.0, 0.05)):
        x += a * np.sin(2 * np.pi * f * t + rng.random() * 6.28)
    return x.astype(np.float32)
