import os, sys, time
sys.path.insert(0, os.environ.get("FAD_TREE", os.environ.get("GRAFT_REPO_ROOT", "/root/repo")))
import numpy as np, torch
from fadtk_amd import hip, _capi as K
rng = np.random.default_rng(0)
d, n, decay = 512, 100000, 1.0
lam = np.arange(1, d + 1) ** (-decay / 2.0)
q, _ = np.linalg.qr(rng.standard_normal((d, d)))
a = torch.from_numpy(((rng.standard_normal((n, d)) * lam) @ q.T).astype(np.float16)).cuda()
b = torch.from_numpy(((1.05 * rng.standard_normal((n, d)) * lam) @ q.T + 0.01).astype(np.float16)).cuda()
with hip.Moments(d) as ma, hip.Moments(d) as mb:
    hip.Moments.update_multi([ma, mb], [a, b])
    for _ in range(3): fad, diag = hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fad, diag = hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)
        ts.append((time.perf_counter() - t0) * 1e3)
print(f"FAD {fad:.6g} iters {diag['iters']}: " + " ".join(f"{t:.3f}" for t in ts))
