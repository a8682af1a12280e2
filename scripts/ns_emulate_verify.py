#!/usr/bin/env python3
"""CPU emulation (numpy) of round 5's WIDE low-precision Frechet chain (csrc/ns_fast.h, frechet.hip: fast_decide_one): pairs whose
product Sigma_1 Sigma_2 has a DECAYING spectrum run the split-float16 Newton-Schulz iteration with scaled steps (device rules of
nsf_split<SP_FIRST> / nsf_check), then the exact correction, then -- where the norm bound ||Z||^3 ||R||^2 / 8 says nothing -- the
verification products  P = Z R,  E = I - Z Y,  Q = Z P:

    tr sqrt(A/c) = tr Y + 1/2 tr(Z R) + 1/2 tr(E P) - [second order],     [second order] estimated by 1/8 |tr(Q P)|,

accepted when 4 x that estimate + ||E||_F^2 ||P||_F moves the distance by less than 4e-6 of itself.  Prints, per spectrum k^-p, the
iterations, the true error of the corrected trace against eig (as a fraction of the FAD), both estimates and the decision.

    python scripts/ns_emulate_verify.py [p ...]              (tests/test_host_logic.py imports `emulate`)
"""
import sys

import numpy as np

f32 = np.float32
K_VER_SCALE = 4096.0
L0_SCALE, L0_MIN, MAX_LOW = 3.0, 9e-4, 22


def split16(x):
    x = x.astype(f32); hi = x.astype(np.float16); lo = ((x - hi.astype(f32)) * f32(2048)).astype(np.float16)
    return hi, lo


def used(s):
    return s[0].astype(f32) + s[1].astype(f32) / f32(2048)


def mm(a, b):
    """split-float16 product: hi hi + (hi lo + lo hi) / 2048, float32 accumulation"""
    ah, al = (t.astype(f32) for t in a); bh, bl = (t.astype(f32) for t in b)
    return (ah @ bh + (ah @ bl + al @ bh) / f32(2048)).astype(f32)


def grid(x):
    return np.rint(x * 2.0 ** 40) / 2.0 ** 40


def step_scale(l):
    if l >= 0.9:
        return 1.0, l * (3 - l * l) / 2
    m = np.sqrt(3 / (1 + l + l * l))
    return m, m * l * (3 - m * m * l * l) / 2


def l0_estimate(pr, d):
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


def scale_cap(res, d):
    return np.sqrt(1 / (1 - min(res / np.sqrt(d), 0.66)))


def emulate(C1, C2, thr=None, verbose=False):
    """-> dict(route, iters, fad_rel_err, est_norm_bound, est_verify, accepted_by, ...) for covariances C1, C2 (float64)"""
    D = C1.shape[0]
    thr = 2.5e-3 * D / 512 if thr is None else thr
    s1 = 2.0 ** -np.ceil(np.log2(np.abs(np.diag(C1)).max())); s2 = 2.0 ** -np.ceil(np.log2(np.abs(np.diag(C2)).max()))
    ev = np.linalg.eigvals(C1 @ C2).real
    tr_true = np.sqrt(np.clip(ev, 0, None)).sum()
    A = grid(C1 * s1) @ grid(C2 * s2); ss = s1 * s2
    tsum = np.trace(C1) + np.trace(C2)
    fad = tsum - 2 * tr_true                                   # (no mean term: the worst case for the cancellation)
    fro = np.linalg.norm(A); u = min(fro, np.abs(A).sum(0).max(), np.abs(A).sum(1).max())
    pr = np.trace(A) ** 2 / fro ** 2
    out = {"cond": ev.max() / max(ev.min(), 1e-300), "pr_over_d": pr / D, "kappa": tsum / fad}
    scaled = pr < 0.8 * D
    if scaled:
        c = u; l = min(l0_estimate(pr, D) * L0_SCALE, 0.5)
        if l < L0_MIN:
            out.update(route="declined", l0=l); return out
    else:
        c = u / 2.9; wm = fro ** 2 / np.trace(A)
        if c < wm <= u:
            c = wm
        l = 1.0
    out["l0"] = l
    I = np.eye(D, dtype=f32)
    Y = split16(A / c); Z = split16(I)
    mu, l = step_scale(l) if scaled else (1.0, 1.0)
    M = mm(Z, Y); T = split16(f32(1.5 * mu) * I - f32(0.5 * mu ** 3) * M); Y, Z = split16(mm(Y, T)), split16(mm(T, Z))
    mu_next = step_scale(l)[0] if scaled else 1.0
    prev, grew, final, k = 1e300, False, None, 1
    while k < MAX_LOW:
        mu = mu_next
        M = mm(Z, Y)
        Tm = f32(1.5 * mu) * I - f32(0.5 * mu ** 3) * M
        res = 2 * np.linalg.norm(Tm.astype(np.float64) - (1.5 * mu - 0.5 * mu ** 3) * np.eye(D)) / mu ** 3
        T = split16(Tm)
        Yn, Zn = split16(mm(Y, T)), split16(mm(T, Z))
        if scaled:
            ll = l
            if res < 1:
                ll = max(ll, np.sqrt(1 - res))
            ll = min(mu * ll * (3 - mu * mu * ll * ll) / 2, 1.0)
            mu_next = min(step_scale(ll)[0], scale_cap(res, D)); l = ll
        grows = k >= 4 and res > prev and res > 1e-3
        if (grows and (not scaled or grew)) or not np.isfinite(res):
            out.update(route="gave up", iters=k); return out
        grew = grows
        if res <= 1e-3 and (res > 0.3 * prev or res <= 1e-6):
            final = (k, Y, Z); break
        if 0.75 * res ** 2 + 0.25 * res ** 3 <= thr and mu == 1.0:
            final = (k + 1, Yn, Zn); break
        prev = res; Y, Z = Yn, Zn; k += 1
    if final is None:
        out.update(route="not finished", iters=MAX_LOW); return out
    fi, Y, Z = final
    Yd = used(Y).astype(np.float64); Zd = used(Z).astype(np.float64)
    R = A / c - grid(Yd) @ grid(Yd)
    t1 = np.trace(Yd) + 0.5 * np.sum(Zd * R.T)
    zn = np.sqrt(np.abs(Zd).sum(0).max() * np.abs(Zd).sum(1).max()); rn = np.linalg.norm(R)
    est_old = zn ** 3 * rn ** 2 / 8                              # (+ the residual term, small here)
    sc = np.sqrt(c / ss)
    rel = lambda t: abs(2 * (sc * t - tr_true)) / abs(fad)        # error of the corrected trace as a fraction of the distance
    out.update(route="chain", iters=fi, scaled=scaled, norm_bound_fad=2 * sc * est_old / abs(fad))
    if est_old <= 1e-9 * abs(t1) or 2 * sc * est_old <= 1e-5 * abs(fad):
        out.update(accepted_by="norm bound", fad_rel_err=rel(t1)); return out
    Rs = split16(R * K_VER_SCALE)
    P = mm(Z, Rs).astype(np.float64); E = (np.eye(D) - mm(Z, Y).astype(np.float64)) * K_VER_SCALE
    Ps, Es = split16(P), split16(E)
    Q = mm(Z, Ps).astype(np.float64)
    inv = 1.0 / K_VER_SCALE ** 2
    Pu, Eu = used(Ps).astype(np.float64), used(Es).astype(np.float64)
    qp = np.sum(Q * Pu.T) * inv; ep = np.sum(Eu * Pu.T) * inv; pp = np.sum(Pu * Pu) * inv; ee = np.sum(Eu * Eu) * inv
    t2 = t1 + 0.5 * ep
    est_v = 4 * abs(qp) / 8 + ee * np.sqrt(pp)
    ok = np.isfinite(est_v) and (est_v <= 1e-9 * abs(t2) or 2 * sc * est_v <= 4e-6 * abs(tsum - 2 * sc * t2))
    out.update(accepted_by="verification" if ok else None, verify_est_fad=2 * sc * est_v / abs(fad), fad_rel_err=rel(t2),
               fad_rel_err_without_e_term=rel(t1), second_order_fad=2 * sc * abs(qp) / 8 / abs(fad))
    if not ok:
        out["route"] = "rejected"
    return out


def bench_pair(p, d=512, n=100000, seed=77):
    """bench.py: extra_decaying's recipe -- both sets share the eigenvectors, Sigma ~ k^-p"""
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.arange(1, d + 1) ** (-p / 2.0)
    a = ((rng.standard_normal((n, d)) * lam) @ q.T).astype(np.float16).astype(np.float64)
    b = ((1.05 * rng.standard_normal((n, d)) * lam) @ q.T + 0.01).astype(np.float16).astype(np.float64)
    return np.cov(a, rowvar=False), np.cov(b, rowvar=False)


if __name__ == "__main__":
    ps = [float(a) for a in sys.argv[1:]] or [0.0, 0.5, 1.0, 1.25, 2.0]
    for p in ps:
        r = emulate(*bench_pair(p))
        print(f"k^-{p:g}:", {k: (f"{v:.3g}" if isinstance(v, float) else v) for k, v in r.items()})
