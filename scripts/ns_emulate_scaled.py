#!/usr/bin/env python3
"""numpy emulation of the SCALED coupled Newton-Schulz iteration for tr sqrt(S1 S2) on decaying spectra:
(Y, Z) <- mu_k (Y, Z) before every step, mu_k^2 = 3 / (1 + l_k + l_k^2), l_{k+1} = mu_k l_k (3 - mu_k^2 l_k^2) / 2 (Chen & Chow's
optimal scaling of the cubic on [l_k, 1]); folded into T: T = 1.5 mu I - 0.5 mu^3 Z Y.  The start needs every eigenvalue of A / c in
(0, 1]: c = an upper bound of the spectral radius.  l_0 comes from the participation ratio (tr A)^2 / ||A||_F^2 through a power-law
model of the spectrum, divided by a safety factor; the schedule stays valid for any l_0 (too small: slower; too large: the
eigenvalues below it take plain-step growth).      python scripts/ns_emulate_scaled.py"""
import numpy as np
import scipy.linalg as sl


def power_law_cov(rng, d, p, n=None):
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.arange(1, d + 1, dtype=np.float64) ** (-p)
    c = (q * lam) @ q.T
    if n:                                   # a sample covariance of n rows drawn from it
        x = rng.standard_normal((n, d)) @ (q * np.sqrt(lam)).T
        c = np.cov(x, rowvar=False)
    return c


def _s(p, d):
    """sum_{k=1..d} k^-p by the trapezoid rule on the integral (what the device code uses: a few pow() per evaluation)"""
    if abs(p - 1.0) < 1e-9:
        return 0.5 * (1.0 + 1.0 / d) + np.log(d)
    return 0.5 * (1.0 + d ** -p) + (d ** (1.0 - p) - 1.0) / (1.0 - p)


def l0_from_participation(pr, d, safety=3.0):
    """x_min estimate: invert PR(p) = (sum k^-p)^2 / sum k^-2p for the exponent p of A's spectrum, then x_min = d^(-p/2)."""
    lo, hi = 0.0, 8.0
    for _ in range(40):
        p = 0.5 * (lo + hi)
        val = _s(p, d) ** 2 / _s(2 * p, d)
        if val > pr: lo = p
        else: hi = p
    return max(min(d ** (-p / 2) / safety, 0.5), 1e-5), p


def run(c1, c2, scaled, l0=None, max_iter=80):
    a = c1 @ c2
    d = a.shape[0]
    u = min(np.linalg.norm(a, "fro"), np.linalg.norm(a, 1), np.linalg.norm(a, np.inf))
    if scaled:
        c = u
    else:                                   # the shipped rule (frechet_f64.hip ns_prepare)
        c = max(u / 2.5, np.linalg.norm(a, "fro") ** 2 / np.trace(a))
    y, z = a / c, np.eye(d)
    l = l0
    res_hist = []
    for k in range(max_iter):
        m = z @ y
        r = np.linalg.norm(np.eye(d) - m, "fro")
        res_hist.append(r)
        if r <= 1e-13 * d or (k > 3 and 0.75 * r * r + 0.25 * r ** 3 <= 1e-13 * d):
            if r > 1e-13 * d:               # predicted: one last update
                t = 1.5 * np.eye(d) - 0.5 * m
                y = y @ t
            break
        mu = 1.0
        if scaled and l is not None and l < 0.9:
            mu = np.sqrt(3.0 / (1.0 + l + l * l))
            l = mu * l * (3.0 - mu * mu * l * l) / 2.0
        t = 1.5 * mu * np.eye(d) - 0.5 * mu ** 3 * m
        y, z = y @ t, t @ z
    return np.sqrt(c) * np.trace(y), k + 1, res_hist


def probe_pair(rng, d, n, decay):
    """the pairs of scripts/probe_illcond.py (VERDICT r03's table): both sets share the eigenvectors, std_k = k^(-decay / 2)"""
    lam = np.arange(1, d + 1) ** (-decay / 2.0)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    a = ((rng.standard_normal((n, d)) * lam) @ q.T).astype(np.float16).astype(np.float64)
    b = ((1.05 * rng.standard_normal((n, d)) * lam) @ q.T + 0.01).astype(np.float16).astype(np.float64)
    return np.cov(a, rowvar=False), np.cov(b, rowvar=False)


def main():
    rng = np.random.default_rng(0)
    d = 512
    for p, n in ((0.5, 100000), (1.0, 100000), (2.0, 100000), (3.0, 20000)):
        c1, c2 = probe_pair(rng, d, n, p)
        a = c1 @ c2
        lam = np.linalg.eigvals(a).real
        want = np.sqrt(np.maximum(lam, 0)).sum()
        pr = np.trace(a) ** 2 / np.trace(a @ a)            # (tr A)^2 / tr(A^2) = (sum lambda)^2 / sum lambda^2, exact also for a non-normal A
        l0, pexp = l0_from_participation(pr, d)
        u = min(np.linalg.norm(a, "fro"), np.linalg.norm(a, 1), np.linalg.norm(a, np.inf))
        x_min = np.sqrt(max(lam.min(), 0) / u)
        t0, it0, _ = run(c1, c2, False)
        t1, it1, h1 = run(c1, c2, True, l0)
        t2, it2, _ = run(c1, c2, True, x_min)          # with the true lower bound
        print(f"k^-{p} (N={n}): PR {pr:7.1f} -> p_hat {pexp:.2f}, l0 {l0:.2e} (true x_min {x_min:.2e}, cond(A) {lam.max() / max(lam.min(), 1e-300):.1e}) | "
              f"plain {it0} it err {abs(t0 - want) / want:.1e} | scaled {it1} it err {abs(t1 - want) / want:.1e} | scaled with true l0 {it2} it err {abs(t2 - want) / want:.1e}")


if __name__ == "__main__":
    main()
