#!/usr/bin/env python3
"""CPU emulation (numpy) of round 3's Frechet chain: Newton-Schulz on SPLIT-float16 operands (x = hi + lo / 2048, three MFMA terms
hi hi + hi lo + lo hi with float32 accumulation) and the two exact products (A = C1 C2, G = Y Y) through base-128 digit planes
on the int8 MFMA (fixed point, 2^-40).  Prints, per iterate, what the float64 correction returns against eig."""
import numpy as np
rng = np.random.default_rng(0)

def split16(x):
    x = x.astype(np.float32)
    hi = x.astype(np.float16)
    lo = ((x - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo

def used(hi, lo):
    return hi.astype(np.float32) + lo.astype(np.float32) / np.float32(2048)

def mm_split(a, b):
    """a, b = (hi, lo) pairs; float32 accumulation emulated by sgemm on the exact float16 pieces"""
    ah, al = (t.astype(np.float32) for t in a); bh, bl = (t.astype(np.float32) for t in b)
    main = ah @ bh
    cross = ah @ bl + al @ bh
    return (main + cross / np.float32(2048)).astype(np.float32)

def digits(x, nd=6, frac=40):
    """balanced base-128 digits of rint(x 2^frac), most significant first; returns (list of int8 arrays, value represented)"""
    t = x.astype(np.float64) * 2.0 ** (frac - 7 * (nd - 1))
    ds = []
    for _ in range(nd):
        d = np.rint(t)
        ds.append(d.astype(np.int64))
        t = (t - d) * 128.0
    val = sum(d.astype(np.float64) * 2.0 ** (7 * (nd - 1 - i) - frac) for i, d in enumerate(ds))
    assert all(np.abs(d).max() <= 64 for d in ds), [np.abs(d).max() for d in ds]
    return ds, val

def mm_digits(xa, xb, nd=6, frac=40, umin=3):
    da, va = digits(xa, nd, frac); db, vb = digits(xb, nd, frac)
    out = np.zeros((xa.shape[0], xb.shape[1]))
    for s in range(nd):          # index 0 = most significant: weight 128^(nd-1-s)
        for t in range(nd):
            u = (nd - 1 - s) + (nd - 1 - t)
            if u < umin: continue
            out += (da[s] @ db[t]).astype(np.float64) * 2.0 ** (7 * u - 2 * frac)
    return out, va, vb

def run(N, D, decay=0.0, label="", thr=2.5e-3):
    lam = np.arange(1, D + 1) ** (-decay / 2.0)
    a = (rng.standard_normal((N, D)) * lam).astype(np.float16).astype(np.float64)
    b = ((1.02 * rng.standard_normal((N, D)) + 0.01) * lam).astype(np.float16).astype(np.float64)
    C1 = np.cov(a, rowvar=False); C2 = np.cov(b, rowvar=False)
    ev = np.linalg.eigvals(C1 @ C2).real
    tr_true = np.sqrt(np.clip(ev, 0, None)).sum()
    # normalisation by powers of two so that the fixed point has range 2
    s1 = 2.0 ** -np.ceil(np.log2(np.abs(C1).max())); s2 = 2.0 ** -np.ceil(np.log2(np.abs(C2).max()))
    A, v1, v2 = mm_digits(C1 * s1, C2 * s2)
    A_exact = (C1 * s1) @ (C2 * s2)
    print(label, "D", D, "A digits err", np.abs(A - A_exact).max(), "max|A|", np.abs(A).max())
    fro = np.linalg.norm(A); one = np.abs(A).sum(0).max(); inf = np.abs(A).sum(1).max()
    u = min(fro, one, inf); c = u / 2.5
    wmean = (A * A).sum() / np.trace(A)
    if c < wmean <= u: c = wmean
    I = np.eye(D, dtype=np.float32)
    Y0 = (A / c).astype(np.float32)
    T0 = (1.5 * I - 0.5 * Y0).astype(np.float32)
    Ys, Ts = split16(Y0), split16(T0)
    Y = split16(mm_split(Ys, Ts)); Z = Ts
    for k in range(1, 7):
        M = mm_split(Z, Y)
        T = split16((1.5 * I - 0.5 * M).astype(np.float32))
        res = 2 * np.linalg.norm(used(*T).astype(np.float64) - I)
        Yu = used(*Y).astype(np.float64); Zu = used(*Z).astype(np.float64)
        G, _, _ = mm_digits(Yu, Yu)
        G_exact = Yu @ Yu
        R = A / c - G
        trs = np.trace(Yu) + 0.5 * np.sum(Zu * R.T)
        trs_exact = np.trace(Yu) + 0.5 * np.sum(Zu * (A_exact / c - G_exact).T)
        zn = np.sqrt(np.abs(Zu).sum(0).max() * np.abs(Zu).sum(1).max()); rn = np.linalg.norm(R)
        est = zn**3 * rn**2 / 8 + zn * res * rn / 2
        scale = np.sqrt(c / (s1 * s2))
        err = abs(scale * trs - tr_true) / tr_true
        err_x = abs(scale * trs_exact - tr_true) / tr_true
        print("  k=%d res=%.3e pred_next=%.3e corr-at-Y_k: relerr=%.2e (exact products %.2e) est/|tr|=%.2e zn=%.2f rn=%.2e  G digit err %.1e"
              % (k, res, .75*res*res+.25*res**3, err, err_x, est / abs(trs), zn, rn, np.abs(G - G_exact).max()))
        Y, Z = split16(mm_split(Y, T)), split16(mm_split(T, Z))

if __name__ == "__main__":
    run(100000, 512, 0.0, "C3")
    run(20000, 512, 0.0, "N20k")
    run(100000, 128, 0.0, "D128")
    run(50000, 512, 0.5, "decay.5")
