#!/usr/bin/env python3
"""Eigenvalue-level emulation of the ADAPTIVE scaled Newton-Schulz steps of the batched per-song chain (csrc/ns_check.h:
ns_l0_from_participation, ns_step_scale, ns_step_scale_with, ns_scale_cap; csrc/ns_fast.h: nsf_check).  Every eigenvalue x of
sqrt(Z Y) follows the scalar map x -> mu x (3 - mu^2 x^2) / 2; the check of iteration k sets mu_{k+1} from a lower bound l of x that it
advances through the cubic, raises to sqrt(1 - r_k) once the residual r_k = ||I - Z Y||_F is below 1, and caps at 1 / sqrt(1 - r_k / sqrt(d)).
Prints the iterations to the low-precision floor for starts l0 = estimate x {0.03 .. 3}, with and without the cap, on the spectra of
bench.py's per-song extras (variances that differ by dimension x a sample covariance of a few D frames).
    python scripts/ns_emulate_adaptive.py          (tests/test_host_logic.py imports `run` and `song_spectrum`)"""
import numpy as np


def song_spectrum(rng, d, n, lo, hi):
    """x = sqrt(lambda / u) of A = Sigma_b Sigma_s for one synthetic song, u = min(Frobenius, 1-, inf-norm) >= rho(A); and PR_F"""
    s = lo + (hi - lo) * rng.random(d)
    x = rng.standard_normal((n, d)) * s
    a = np.diag(s ** 2 * 1.05 ** 2) @ np.cov(x, rowvar=False)
    lam = np.linalg.eigvals(a).real
    u = min(np.linalg.norm(a, "fro"), np.linalg.norm(a, 1), np.linalg.norm(a, np.inf))
    return np.sqrt(np.maximum(lam, 1e-300) / u), np.trace(a) ** 2 / np.linalg.norm(a, "fro") ** 2


def l0_estimate(pr, d):
    """ns_l0_from_participation: power-law model of the spectrum, a third of its x_min"""
    def s(p):
        return 0.5 * (1 + d ** -p) + ((d ** (1 - p) - 1) / (1 - p) if abs(p - 1) > 1e-6 else np.log(d))
    lo, hi = 0.0, 8.0
    for _ in range(40):
        p = 0.5 * (lo + hi)
        if s(p) ** 2 / s(2 * p) > pr:
            lo = p
        else:
            hi = p
    return min(max(d ** (-p / 2) / 3, 1e-5), 0.5)


def step_scale(l):
    """ns_step_scale: -> (mu, bound after the step)"""
    if l >= 0.9:
        return 1.0, l * (3 - l * l) / 2
    m = np.sqrt(3 / (1 + l + l * l))
    return m, m * l * (3 - m * m * l * l) / 2


def run(x, l0, cap=True, floor=1e-3, thr_pred=3e-4, max_iter=40, tolerant=True):
    """-> iterations until the chain's check closes the problem (negative: the 'residual grows' rule gave up at that iteration)"""
    d = len(x)
    x = x.copy()
    mu, l = step_scale(l0)                       # iteration 0
    x = mu * x * (3 - mu * mu * x * x) / 2
    k, prev, grew = 1, 1e300, False
    mu, _ = step_scale(l)
    while k < max_iter:
        res = float(np.sqrt(np.sum((1 - x * x) ** 2)))
        grows = k >= 4 and res > prev and res > 1e-3
        if grows and (grew or not tolerant):                     # (nsf_check: two growing residuals in a row)
            return -k
        grew = grows
        if res <= floor:
            return k
        if 0.75 * res ** 2 + 0.25 * res ** 3 <= thr_pred and mu == 1.0:
            return k + 1
        prev = res
        if res < 1:
            l = max(l, np.sqrt(1 - res))
        x = mu * x * (3 - mu * mu * x * x) / 2                     # the update of iteration k, with the scale decided before
        l = min(mu * l * (3 - mu * mu * l * l) / 2, 1.0)
        mu, _ = step_scale(l)
        if cap:
            mu = min(mu, np.sqrt(1 / (1 - min(res / np.sqrt(d), 0.66))))
        k += 1
    return max_iter


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for name, (d, n, lo, hi) in {"[1500 x 768]": (768, 1500, 0.5, 1.5), "[1200 x 512]": (512, 1200, 0.5, 1.5), "[2250 x 128]": (128, 2250, 0.6, 1.4)}.items():
        x, pr = song_spectrum(rng, d, n, lo, hi)
        e = l0_estimate(pr, d)
        print(f"{name}: x in [{x.min():.4f}, {x.max():.3f}], estimate {e:.4f}, participation ratio / d {pr / d:.2f}")
        for cap, tol in ((False, False), (True, False), (True, True)):
            print(("   cap" if cap else "no cap") + (", gives up after TWO growing residuals" if tol else ", gives up at the first growing residual"),
                  {sc: run(x, min(e * sc, 0.5), cap, tolerant=tol) for sc in (0.03, 0.1, 0.25, 0.5, 1, 2, 3)})
