#!/usr/bin/env python3
"""What would a QUINTIC Newton-Schulz step buy?  Best one-sided odd polynomials p of degree 3 and 5 on [l, 1] -- maximise
min p subject to p <= 1 (linear programme on a grid) -- and the factor by which they lift the lower end.  A quintic step is three dependent
GEMM levels (M = Z Y; M^2 -> T = a I + b M + c M^2; Y T and T Z), a cubic step two.     python scripts/ns_quintic_lp.py"""
import numpy as np
from scipy.optimize import linprog


def best(l, deg):
    xs = np.unique(np.concatenate([np.geomspace(l, 1, 400), np.linspace(l, 1, 400)]))
    nco = (deg + 1) // 2
    v = np.stack([xs ** (2 * j + 1) for j in range(nco)], 1)
    c = np.zeros(nco + 1); c[-1] = -1
    a = np.vstack([np.hstack([v, np.zeros((len(xs), 1))]), np.hstack([-v, np.ones((len(xs), 1))])])
    b = np.concatenate([np.ones(len(xs)), np.zeros(len(xs))])
    r = linprog(c, A_ub=a, b_ub=b, bounds=[(None, None)] * (nco + 1), method="highs")
    return r.x[:nco], r.x[-1]


if __name__ == "__main__":
    for l in (1e-6, 1e-4, 1e-3, 1e-2, 0.05, 0.1, 0.2, 0.3, 0.5, 0.7, 0.9):
        c3, t3 = best(l, 3); c5, t5 = best(l, 5)
        print(f"l = {l:8.1e}: cubic lifts {t3 / l:5.2f} x (coefficients {np.round(c3, 4)}), quintic {t5 / l:5.2f} x ({np.round(c5, 4)}); "
              f"per microsecond at 30 / 43 us a step: {np.log(t3 / l) / 30:.4f} / {np.log(t5 / l) / 43:.4f}")
