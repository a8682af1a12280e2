"""Round-3 probe of the batched per-song call: `python scripts/songs_probe.py c5|c4 [calls]` builds the synthetic songs of
bench.py's extra (32 x [1500 x 768] or 2000 x [2250 x 128]) and runs the call a few times (for rocprofv3 --kernel-trace --stats);
`python scripts/songs_probe.py steep` re-runs the mixed case of tests/test_gpu_parity.py with the fast chain on and off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

dev = torch.device("cuda", 0)
what = sys.argv[1]
if what in ("c5", "c4", "gen"):
    # gen <d> <frames> <songs> [calls]: any shape
    if what == "gen":
        d, frames, nsongs = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        sys.argv = sys.argv[:2] + sys.argv[5:]
    else:
        nsongs, frames, d = (32, 1500, 768) if what == "c5" else (2000, 2250, 128)
    g = torch.Generator(device=dev); g.manual_seed(55)
    scale = 0.5 + torch.rand((d,), generator=g, device=dev)
    songs = (torch.randn((nsongs * frames, d), generator=g, device=dev) * scale).to(torch.float16)
    base = torch.randn((20000, d), generator=g, device=dev, dtype=torch.float64) * scale.double() * 1.05 + 0.01
    mu, cov = base.mean(0).cpu().numpy(), torch.cov(base.T).cpu().numpy()
    offs = np.arange(0, nsongs * frames + 1, frames)
    for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc, st = hip.frechet_batched(mu, cov, songs, offs)
        torch.cuda.synchronize(); print(what, "call", i, "%.3f ms" % ((time.perf_counter() - t0) * 1e3), "ok", int((st == 0).sum()))
elif what in ("acc4", "acc5"):
    # accuracy of the batched call on bench.py's per-song extras (a few songs against the oracle), under whatever FAD_SONG_* is set
    from oracle import fad_oracle as O
    nsongs, frames, d = (24, 2250, 128) if what == "acc4" else (6, 1500, 768)
    g = torch.Generator(device=dev); g.manual_seed(44 if what == "acc4" else 55)
    if what == "acc4":
        scale = 0.6 + 0.8 * torch.rand((d,), generator=g, device=dev)
        songs = (torch.randn((nsongs * frames, d), generator=g, device=dev) * scale).to(torch.float16)
        base = torch.randn((50000, d), generator=g, device=dev, dtype=torch.float64) * scale.double() * 1.03 + 0.02
    else:
        scale = 0.5 + torch.rand((d,), generator=g, device=dev)
        songs = (torch.randn((nsongs * frames, d), generator=g, device=dev) * scale).to(torch.float16)
        base = torch.randn((20000, d), generator=g, device=dev, dtype=torch.float64) * scale.double() * 1.05 + 0.01
    mu, cov = base.mean(0).cpu().numpy(), torch.cov(base.T).cpu().numpy()
    offs = np.arange(0, nsongs * frames + 1, frames)
    sc, st = hip.frechet_batched(mu, cov, songs, offs)
    host = songs.cpu().numpy()
    want = np.array(O.individual_scores(mu, cov, [host[i * frames:(i + 1) * frames] for i in range(nsongs)], run_sqrtm=False), dtype=np.float64)
    rel = np.abs(sc - want) / np.abs(want)
    print(what, {k: v for k, v in os.environ.items() if k.startswith("FAD_SONG")}, "max rel %.3e  median %.3e  (scores %.4f .. %.4f)" % (rel.max(), np.median(rel), want.min(), want.max()))
else:
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    from oracle import fad_oracle as O
    import recipes as R
    for d, frames in ((768, [1500, 900]), (512, [1100]), (1024, [2100])):
        rng = np.random.default_rng(d)
        mu_b, cov_b = R.baseline_stats(700 + d, 6 * d, d)
        sg = [(rng.standard_normal((n, d)) * (0.7 + 0.6 * rng.random(d)) + 0.1 * rng.standard_normal(d)).astype(np.float16) for n in frames]
        flat = np.tile(sg[0][:1], (d + 5, 1))
        steep = (rng.standard_normal((2 * d, d)) * np.arange(1, d + 1) ** -1.5).astype(np.float16)
        sg = sg + [flat, steep]
        rows = np.concatenate(sg)
        offs = np.concatenate([[0], np.cumsum([s.shape[0] for s in sg])])
        want = np.array(O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False), dtype=np.float64)
        want_s = np.array(O.individual_scores(mu_b, cov_b, sg[-1:], run_sqrtm=True), dtype=np.float64)
        print("d", d, "oracle eig", want, "sqrtm(steep)", want_s)
        sc, st = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
        print("  fast", os.environ.get("FAD_SONG_FAST"), "sym", os.environ.get("FAD_SONG_SYM"), sc, st, "rel", np.abs(sc - want) / np.abs(want))
