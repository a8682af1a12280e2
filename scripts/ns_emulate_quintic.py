#!/usr/bin/env python3
"""numpy emulation: DEGREE-5 steps for the growth phase of the coupled Newton-Schulz square root (float64 route).
A step is X = Z Y, T = a I + b X + c X^2, Y <- Y T, Z <- T Z: four products in three dependent launches, where the scaled cubic step
(T = 1.5 mu I - 0.5 mu^3 X) is three products in two.  (a, b, c) = the odd quintic p(x) = a x + b x^3 + c x^5 closest to 1 on [l, 1] in
the maximum norm (equioscillation, Remez), divided by 1 + E so that the image is [(1 - E) / (1 + E), 1] = the next step's interval --
the composition of Amsel et al.'s "Polar Express", used here on the eigenvalues x of sqrt(A / c)'s iterate.  For small l the quintic lifts
the bottom of the spectrum by up to ~8x per step, the scaled cubic by <= 2.6x.
    python scripts/ns_emulate_quintic.py            -> iterations / products / launches / error against eig for the probe pairs
    python scripts/ns_emulate_quintic.py table      -> the coefficient table shipped in fadtk_amd/csrc/ns_quintic.h"""
import sys
import numpy as np
sys.path.insert(0, __file__.rsplit("/", 1)[0])
from ns_emulate_scaled import probe_pair, l0_from_participation


def minimax_quintic(l):
    """(a, b, c, E): p(x) = a x + b x^3 + c x^5 with max |1 - p| on [l, 1] minimal (= E)."""
    if l >= 1.0 - 1e-12:
        return 15 / 8, -10 / 8, 3 / 8, 0.0
    # reference points l = x0 < x1 < x2 < x3 = 1, errors alternate:  p(x_i) = 1 - (-1)^i E
    xs = np.array([l, l + (1 - l) * 0.25, l + (1 - l) * 0.65, 1.0])
    for _ in range(200):
        m = np.array([[x, x ** 3, x ** 5, (-1.0) ** i] for i, x in enumerate(xs)])
        a, b, c, e = np.linalg.solve(m, np.ones(4))
        # interior extrema of p: a + 3 b s + 5 c s^2 = 0, s = x^2
        disc = 9 * b * b - 20 * a * c
        if disc <= 0: break
        s = np.sort([(-3 * b - np.sqrt(disc)) / (10 * c), (-3 * b + np.sqrt(disc)) / (10 * c)])
        if s[0] <= 0: break
        new = np.array([l, np.sqrt(s[0]), np.sqrt(s[1]), 1.0])
        if not (l < new[1] < new[2] < 1.0): break
        if np.max(np.abs(new - xs)) < 1e-15: xs = new; break
        xs = new
    return a, b, c, abs(e)


def quintic_schedule(l, until=0.9, max_steps=12):
    """[(a, b, c)] normalised to image (.., 1], and the lower bounds l_k after each step"""
    out, ls = [], []
    while l < until and len(out) < max_steps:
        a, b, c, e = minimax_quintic(l)
        out.append((a / (1 + e), b / (1 + e), c / (1 + e)))
        l = (1 - e) / (1 + e)
        ls.append(l)
    return out, ls


def run_quintic(c1, c2, l0, until=0.9, tol=1e-13, max_iter=60, dtype=np.float64):
    a = (c1 @ c2).astype(dtype)
    d = a.shape[0]
    u = min(np.linalg.norm(a, "fro"), np.linalg.norm(a, 1), np.linalg.norm(a, np.inf))
    y, z = a / u, np.eye(d, dtype=dtype)
    sched, _ = quintic_schedule(l0, until)
    eye = np.eye(d, dtype=dtype)
    products = launches = 0
    hist = []
    for k in range(max_iter):
        x = z @ y; products += 1; launches += 1
        r = np.linalg.norm(eye - x, "fro"); hist.append(r)
        if r <= tol * d or (k >= len(sched) and 0.75 * r * r + 0.25 * r ** 3 <= tol * d):
            if r > tol * d:
                y = y @ (1.5 * eye - 0.5 * x); products += 1; launches += 1
            break
        if k < len(sched):
            qa, qb, qc = sched[k]
            t = qa * eye + qb * x + qc * (x @ x); products += 1; launches += 1
        else:
            t = 1.5 * eye - 0.5 * x                # (the cubic's T comes out of the X launch's epilogue)
        y, z = y @ t, t @ z; products += 2; launches += 1
    return np.sqrt(u) * np.trace(y), k + 1, products, launches, hist


def run_cubic(c1, c2, l0, tol=1e-13, max_iter=80):
    a = c1 @ c2
    d = a.shape[0]
    u = min(np.linalg.norm(a, "fro"), np.linalg.norm(a, 1), np.linalg.norm(a, np.inf))
    y, z, eye, l = a / u, np.eye(d), np.eye(d), l0
    products = launches = 0
    for k in range(max_iter):
        x = z @ y; products += 1; launches += 1
        r = np.linalg.norm(eye - x, "fro")
        if r <= tol * d or (k > 3 and 0.75 * r * r + 0.25 * r ** 3 <= tol * d):
            if r > tol * d: y = y @ (1.5 * eye - 0.5 * x); products += 1; launches += 1
            break
        mu = 1.0
        if l < 0.9:
            mu = np.sqrt(3.0 / (1.0 + l + l * l)); l = mu * l * (3.0 - mu * mu * l * l) / 2.0
        t = 1.5 * mu * eye - 0.5 * mu ** 3 * x
        y, z = y @ t, t @ z; products += 2; launches += 1
    return np.sqrt(u) * np.trace(y), k + 1, products, launches


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "table":
        for e2 in range(2, 36):
            l = 2.0 ** (-e2 / 2)
            sched, ls = quintic_schedule(l)
            print(f"l0 = 2^-{e2 / 2:4.1f} = {l:.3e}: {len(sched)} steps; " + " ".join(f"({a:.4f},{b:.4f},{c:.4f})->{x:.3f}" for (a, b, c), x in zip(sched, ls)))
        return
    rng = np.random.default_rng(0)
    d = 512
    for p, n in ((0.5, 100000), (1.0, 100000), (2.0, 100000), (3.0, 20000)):
        c1, c2 = probe_pair(rng, d, n, p)
        a = c1 @ c2
        lam = np.linalg.eigvals(a).real
        want = np.sqrt(np.maximum(lam, 0)).sum()
        pr = np.trace(a) ** 2 / np.trace(a @ a)
        l0, _ = l0_from_participation(pr, d)
        tc, ic, pc, lc = run_cubic(c1, c2, l0)
        print(f"k^-{p}: l0 {l0:.2e} | scaled cubic: {ic} it, {pc} products, {lc} launches, err {abs(tc - want) / want:.1e}")
        for until in (0.9, 0.99, 0.999):
            for scale in (1.0, 0.3):
                tq, iq, pq, lq, h = run_quintic(c1, c2, l0 * scale, until)
                print(f"        quintic until l >= {until}, l0 x {scale}: {iq} it ({len(quintic_schedule(l0 * scale, until)[0])} quintic), {pq} products, {lq} launches, err {abs(tq - want) / want:.1e}")


if __name__ == "__main__":
    main()
