"""CPU oracle for the FAD hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy/scipy restatement of the reference algorithm (microsoft/fadtk v1.1.0) for the one
path this repository accelerates:

    embeddings [N x D]  ->  (n, sum x, sum x x^T)  ->  (mu, Sigma)  ->  Frechet distance

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this module, and only as the checker / reported CPU baseline.  The product package
(``fadtk_amd``) never imports it: the HIP library is the only compute path there.

Parity status: PINNED.  Every function below is checked bit-for-bit or to <=1e-12 against
outputs of the reference itself (imported in the authoring container by
``tests/golden/make_golden.py``; fixtures under ``tests/golden/*.npz|json``), see
``tests/test_oracle_golden.py``.

Each function cites the reference lines it follows (paths relative to the reference root).
The dtype behaviour of numpy is part of the algorithm here: ``np.mean`` of a
float16 matrix is float16, ``np.cov`` is float64, and ``diff.dot(diff)`` of float16 vectors is
rounded to float16.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Iterable, List, NamedTuple, Sequence, Tuple

import numpy as np
from numpy.lib.scimath import sqrt as _complex_sqrt
from scipy import linalg as _la

_log = logging.getLogger("fad_oracle")


# --------------------------------------------------------------------------------------
# a1  calc_embd_statistics            fadtk/fad.py:42-48
# --------------------------------------------------------------------------------------
def embd_statistics(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(mu, Sigma) of a frame matrix ``rows [N x D]`` (fad.py:42-48).

    mu keeps numpy's mean dtype rule (float16 in -> float16 out, summed in float32);
    Sigma is ``np.cov(rowvar=False)`` = float64, ddof=1.  N < 2 is an AssertionError.
    """
    if rows.shape[0] < 2:
        raise AssertionError(
            f"FAD requires at least two embedding window frames, you have {rows.shape}.")
    mu = np.mean(rows, axis=0)
    sigma = np.cov(rows, rowvar=False)
    return mu, sigma


# --------------------------------------------------------------------------------------
# a2  calc_frechet_distance           fadtk/fad.py:51-120
# --------------------------------------------------------------------------------------
class FrechetParts(NamedTuple):
    mean_term: float      # diff . diff                       (fad.py:83, 119)
    tr1: float            # trace(cov1)                       (fad.py:119)
    tr2: float            # trace(cov2)                       (fad.py:120)
    tr_sqrt_eig: float    # trace of V sqrt(D) V^-1           (fad.py:91-92, 108)  <- returned root
    tr_sqrt_schur: float  # trace of scipy.linalg.sqrtm       (fad.py:88, 109)     <- diagnostic
    used_eps: bool        # eps fallback taken                (fad.py:94-99)


def frechet_parts(mu1, cov1, mu2, cov2, eps: float = 1e-6, run_sqrtm: bool = True) -> FrechetParts:
    """All the scalars that fad.py:51-120 combines; see ``frechet_distance``."""
    mu1 = np.atleast_1d(mu1)
    mu2 = np.atleast_1d(mu2)
    cov1 = np.atleast_2d(cov1)
    cov2 = np.atleast_2d(cov2)
    if mu1.shape != mu2.shape:                      # fad.py:78-79
        raise AssertionError(
            f"Training and test mean vectors have different lengths ({mu1.shape} vs {mu2.shape})")
    if cov1.shape != cov2.shape:                    # fad.py:80-81
        raise AssertionError(
            f"Training and test covariances have different dimensions ({cov1.shape} vs {cov2.shape})")

    delta = mu1 - mu2                               # fad.py:83 (dtype of the inputs is kept)
    product = cov1.dot(cov2)

    # fad.py:88 -- the Schur square root is computed by the reference on every call, but it
    # only feeds the "high error" warning (fad.py:108-117), never the returned value.
    tr_schur = float("nan")
    if run_sqrtm:
        root_schur = _la.sqrtm(product)
        tr_schur = np.trace(root_schur)
        if np.iscomplexobj(tr_schur) and abs(tr_schur.imag) < 1e-3:
            tr_schur = tr_schur.real

    # fad.py:91-92 -- the root that IS returned: eigendecomposition, complex sqrt of the
    # eigenvalues (negative ones become imaginary), V sqrt(D) V^-1.
    evals, evecs = _la.eig(product)
    root = (evecs * _complex_sqrt(evals)) @ _la.inv(evecs)

    used_eps = False
    if not np.isfinite(root).all():                 # fad.py:94-99
        used_eps = True
        _log.info("fid calculation produces singular product; adding %s to diagonal of cov estimates", eps)
        bump = np.eye(cov1.shape[0]) * eps
        root = _la.sqrtm((cov1 + bump).dot(cov2 + bump))

    if np.iscomplexobj(root):                       # fad.py:102-106
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(root.imag))))
        root = root.real

    tr_eig = np.trace(root)
    if run_sqrtm and not np.iscomplexobj(tr_schur):  # fad.py:114-117
        gap = np.abs(tr_eig - tr_schur)
        if gap > 1e-3:
            _log.warning("Detected high error in sqrtm calculation: %s", gap)

    return FrechetParts(delta.dot(delta), np.trace(cov1), np.trace(cov2), tr_eig,
                        tr_schur, used_eps)


def frechet_distance(mu1, cov1, mu2, cov2, eps: float = 1e-6, run_sqrtm: bool = True):
    """||mu1-mu2||^2 + tr(C1) + tr(C2) - 2 tr sqrt(C1 C2)   (fad.py:119-120).

    ``run_sqrtm=False`` skips the diagnostic-only Schur root (fad.py:88); the returned value is
    identical, it only saves time in tests.  The timed CPU baseline keeps it on, as the
    reference pays for both roots on every call.
    """
    p = frechet_parts(mu1, cov1, mu2, cov2, eps=eps, run_sqrtm=run_sqrtm)
    return p.mean_term + p.tr1 + p.tr2 - 2 * p.tr_sqrt_eig


# --------------------------------------------------------------------------------------
# a3  _process_file                   fadtk/utils.py:13-16
# a4  calculate_embd_statistics_online fadtk/utils.py:19-46
# --------------------------------------------------------------------------------------
def file_moments(rows: np.ndarray) -> Tuple[np.ndarray, np.ndarray, int]:
    """Per-file (mean, centred scatter, n)  (utils.py:13-16).

    mean has numpy's dtype rule (float16 stays float16); scatter = cov * (n-1) in float64.
    A one-row file yields a NaN scatter (0/0 in np.cov) -- SURVEY.md Q5.
    """
    n = rows.shape[0]
    with np.errstate(all="ignore"):
        scatter = np.cov(rows, rowvar=False) * (n - 1)
    return np.mean(rows, axis=0), scatter, n


def statistics_online(blocks: Sequence[np.ndarray]) -> Tuple[np.ndarray, np.ndarray]:
    """Sequential pairwise merge of per-file moments into dataset (mu, Sigma) (utils.py:19-46).

    ``blocks`` are the per-file frame matrices in file order (the reference np.load()s them).
    """
    if len(blocks) == 0:
        raise AssertionError("No files provided")
    d = blocks[0].shape[-1]                         # utils.py:28
    mu = np.zeros(d)
    scatter = np.zeros((d, d))
    count = 0
    for rows in blocks:                             # utils.py:35-40
        m_f, s_f, n_f = file_moments(rows)
        step = m_f - mu
        mu += n_f / (count + n_f) * step
        scatter += s_f + step[:, None] * step[None, :] * count * n_f / (count + n_f)
        count += n_f
    if count < 2:                                   # utils.py:42-43
        return mu, np.zeros_like(scatter)
    return mu, scatter / (count - 1)                # utils.py:45


# --------------------------------------------------------------------------------------
# a7  score_individual                fadtk/fad.py:353-395   (per-song scores + CSV text)
# --------------------------------------------------------------------------------------
def individual_scores(mu_base, cov_base, songs: Sequence[np.ndarray], run_sqrtm: bool = True) -> List:
    """Per-song FAD against one baseline (fad.py:373-378); failures become None (fad.py:380-383)."""
    out = []
    for rows in songs:
        try:
            mu_s, cov_s = embd_statistics(rows)
            out.append(frechet_distance(mu_base, cov_base, mu_s, cov_s, run_sqrtm=run_sqrtm))
        except Exception as exc:                    # noqa: BLE001 - the reference swallows everything
            _log.error("individual FAD failed: %s", exc)
            out.append(None)
    return out


def individual_csv_text(paths: Sequence, scores: Sequence) -> str:
    """CSV body written by fad.py:390-393: drop failures, sort by |score|, ',' -> '_'."""
    pairs = [(p, s) for p, s in zip(paths, scores) if s is not None]
    pairs.sort(key=lambda ps: np.abs(ps[1]))        # stable, like sorted()
    return "\n".join(",".join(str(x).replace(",", "_") for x in row) for row in pairs)


# --------------------------------------------------------------------------------------
# f1  score_inf                       fadtk/fad.py:304-351
# --------------------------------------------------------------------------------------
class InfResult(NamedTuple):
    score: float
    slope: float
    r2: float
    points: list


def score_inf(mu_base, cov_base, rows: np.ndarray, steps: int = 25, min_n: int = 500,
              run_sqrtm: bool = True) -> InfResult:
    """FAD-infinity extrapolation (fad.py:325-351).  Uses the GLOBAL numpy RNG exactly like the
    reference (``np.random.choice`` at fad.py:333): seed it before calling to reproduce."""
    max_n = len(rows)
    ns = [int(n) for n in np.linspace(min_n, max_n, steps)]      # fad.py:328
    points = []
    for n in ns:
        pick = np.random.choice(rows.shape[0], size=n, replace=True)   # fad.py:333
        mu_e, cov_e = embd_statistics(rows[pick])
        points.append([n, frechet_distance(mu_base, cov_base, mu_e, cov_e, run_sqrtm=run_sqrtm)])
    ys = np.array(points)
    xs = 1 / np.array(ns)
    slope, intercept = np.polyfit(xs, ys[:, 1], 1)               # fad.py:345
    fit = slope * xs + intercept
    r2 = 1 - np.sum((ys[:, 1] - fit) ** 2) / np.sum((ys[:, 1] - np.mean(ys[:, 1])) ** 2)
    return InfResult(intercept, slope, r2, points)


# --------------------------------------------------------------------------------------
# Whole-job helper used by bench.py's cpu_baseline leg and by smoke()
# --------------------------------------------------------------------------------------
def fad_between(rows_a: np.ndarray, rows_b: np.ndarray):
    """One 'FAD score' exactly as the reference computes it from two frame matrices:
    2 x embd_statistics (fad.py:42-48) + frechet_distance with BOTH roots (fad.py:88-92)."""
    mu_a, cov_a = embd_statistics(rows_a)
    mu_b, cov_b = embd_statistics(rows_b)
    return frechet_distance(mu_a, cov_a, mu_b, cov_b, run_sqrtm=True)
