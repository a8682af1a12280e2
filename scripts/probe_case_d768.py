#!/usr/bin/env python3
"""GPU diagnostic for one pair of tests/test_gpu_fuzz.py::test_fuzz_wide_chain_decaying_pairs_against_oracle (case 0: D = 768, spectra
k^-1.26 / k^-1.16, 15360 and 1152 frames -- the second set barely full rank): where does the distance differ from the reference's eig
formula -- in the moments or in the root?  Prints the distance from (a) the library end to end, (b) the oracle on the frames, (c) the oracle
on the library's own (mu, Sigma), (d) the library's root on the oracle's float64 (mu, Sigma)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fadtk_amd
from fadtk_amd import hip
from oracle import fad_oracle as O

rng = np.random.default_rng(505)
d = int(rng.choice([256, 384, 512, 512, 768]))
p1 = float(rng.uniform(0.2, 1.7)); p2 = p1 + float(rng.choice([0.0, 0.0, 0.1, -0.1]))
n1 = int(d * rng.choice([1.5, 4, 20, 60])); n2 = int(d * rng.choice([1.5, 4, 20]))
scale = float(rng.choice([1e-3, 1.0, 1.0, 30.0]))
q1, _ = np.linalg.qr(rng.standard_normal((d, d)))
q2 = q1 if rng.random() < 0.7 else np.linalg.qr(q1 + 0.05 * rng.standard_normal((d, d)))[0]
lam1 = np.arange(1, d + 1) ** (-p1 / 2.0); lam2 = np.arange(1, d + 1) ** (-p2 / 2.0)
off = float(rng.choice([0.0, 0.01, 0.5]))
a = (((rng.standard_normal((n1, d)) * lam1) @ q1.T) * scale).astype(np.float16)
b = (((1.05 * rng.standard_normal((n2, d)) * lam2) @ q2.T + off * lam2.mean()) * scale).astype(np.float16)
print(f"d={d} p=({p1:.2f},{p2:.2f}) n=({n1},{n2}) scale={scale} off={off}")
a64, b64 = a.astype(np.float64), b.astype(np.float64)
m1, c1, m2, c2 = a64.mean(0), np.cov(a64, rowvar=False), b64.mean(0), np.cov(b64, rowvar=False)
want = O.frechet_distance(m1, c1, m2, c2, run_sqrtm=False)
with hip.Moments(d) as ma, hip.Moments(d) as mb:
    ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
    got, dg = hip.frechet_from_moments(ma, mb, mean_dtype=0)
    g1, k1, _ = ma.finalize(); g2, k2, _ = mb.finalize()
print(f"(a) library end to end   {got:.9f}  rel {abs(got - want) / want:.2e}  diag {dg}")
print(f"(b) oracle on the frames {want:.9f}")
onlib = O.frechet_distance(g1, k1, g2, k2, run_sqrtm=False)
print(f"(c) oracle on the library's (mu, Sigma) {onlib:.9f}  rel to (b) {abs(onlib - want) / want:.2e}   [moments]   rel (a) to (c) {abs(got - onlib) / onlib:.2e}   [root]")
for nm, x, y in (("mu1", g1, m1), ("mu2", g2, m2), ("cov1", k1, c1), ("cov2", k2, c2)):
    print(f"     {nm}: max |diff| {np.abs(x - y).max():.3e} (max |ref| {np.abs(y).max():.3e}); diagonal {np.abs(np.diag(x - y)).max() if x.ndim == 2 else 0:.3e}")
root = float(fadtk_amd.calc_frechet_distance(m1, c1, m2, c2))
print(f"(d) library root on the oracle's float64 (mu, Sigma) {root:.9f}  rel {abs(root - want) / want:.2e}")
root2 = float(fadtk_amd.calc_frechet_distance(g1, k1, g2, k2))
print(f"(e) library root (host route) on the library's (mu, Sigma) {root2:.9f}  rel to (c) {abs(root2 - onlib) / onlib:.2e}")
