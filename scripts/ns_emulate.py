#!/usr/bin/env python3
"""CPU emulation (numpy) of the float32 Newton-Schulz leg + the float64 correction of frechet.hip: residual per iteration, the
bound on the next one, and what the correction would return from EACH iterate (true error against eig, error estimate).  Used to
choose the stop rule of the float32 leg."""
import numpy as np, scipy.linalg as sl
rng = np.random.default_rng(0)
def run(N, D, decay=0.0, label=""):
    lam = np.arange(1, D + 1) ** (-decay / 2.0)
    a = (rng.standard_normal((N, D)) * lam).astype(np.float16).astype(np.float64)
    b = ((1.02 * rng.standard_normal((N, D)) + 0.01) * lam).astype(np.float16).astype(np.float64)
    C1 = np.cov(a, rowvar=False); C2 = np.cov(b, rowvar=False)
    A = C1 @ C2
    ev = np.linalg.eigvals(A).real
    tr_true = np.sqrt(np.clip(ev, 0, None)).sum()
    fro = np.linalg.norm(A); one = np.abs(A).sum(0).max(); inf = np.abs(A).sum(1).max()
    u = min(fro, one, inf); c = u / 2.5
    wmean = np.trace(A @ A) / np.trace(A)
    if c < wmean <= u: c = wmean
    Y = (A / c).astype(np.float32); I = np.eye(D, dtype=np.float32)
    T = (1.5 * I - 0.5 * Y).astype(np.float32); Z = T.copy()
    r0 = np.linalg.norm(I - Y)
    Y = (Y @ T).astype(np.float32)
    print(label, "D", D, "cond-ish", ev.max() / ev.min(), "c/lmax", c / ev.max(), "res0 %.3e" % r0)
    for k in range(1, 9):
        M = (Z @ Y).astype(np.float32)
        T = (1.5 * I - 0.5 * M).astype(np.float32)
        res = np.linalg.norm(I.astype(np.float64) - M.astype(np.float64))
        # corrected trace from (Y, Z) = iterate k
        Y64 = Y.astype(np.float64); Z64 = Z.astype(np.float64)
        R = A / c - Y64 @ Y64
        trs = np.trace(Y64) + 0.5 * np.sum(Z64 * R.T)
        zn = np.sqrt(np.abs(Z64).sum(0).max() * np.abs(Z64).sum(1).max()); rn = np.linalg.norm(R)
        est = zn**3 * rn**2 / 8 + zn * res * rn / 2
        err = abs(np.sqrt(c) * trs - tr_true) / tr_true
        print("  k=%d res=%.3e pred_next=%.3e  corr-at-Y_k: relerr=%.2e est/|tr|=%.2e zn=%.2f rn=%.2e" % (k, res, .75*res*res+.25*res**3, err, est / abs(trs), zn, rn))
        Y = (Y @ T).astype(np.float32); Z = (T @ Z).astype(np.float32)
run(100000, 512, 0.0, "C3")
run(20000, 512, 0.0, "N20k")
run(100000, 768, 0.0, "D768")
run(100000, 128, 0.0, "D128")
run(50000, 512, 0.5, "decay.5")
run(50000, 512, 1.0, "decay1")
