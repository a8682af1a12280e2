#!/usr/bin/env python3
"""Per-song batched chain, plain against scaled Newton-Schulz steps (FAD_SONG_SCALED=0 / 1): bench.py's two full-rank shapes, time per call,
iterations per song (FAD_FAST_TRACE), scores against each other and against the float64 routes."""
import os, sys, time, re, subprocess
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from fadtk_amd import hip
dev = torch.device("cuda", 0)

def make(nsongs, frames, d, seed, lo):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    scale = lo + (1.0 if lo == 0.5 else 0.8) * torch.rand((d,), generator=g, device=dev)
    songs = (torch.randn((nsongs * frames, d), generator=g, device=dev) * scale).to(torch.float16)
    base = torch.randn((20000 if d == 768 else 50000, d), generator=g, device=dev, dtype=torch.float64) * scale.double() * 1.05 + 0.01
    return base.mean(0).cpu().numpy(), torch.cov(base.T).cpu().numpy(), songs, np.arange(0, nsongs * frames + 1, frames)

if len(sys.argv) > 1 and sys.argv[1] == "trace":          # child: one call with the trace on
    shape = {"768": (32, 1500, 768, 55, 0.5), "512": (32, 1200, 512, 56, 0.5), "128": (2000, 2250, 128, 44, 0.6)}[sys.argv[2]]
    mu, cov, songs, offs = make(*shape)
    hip.frechet_batched(mu, cov, songs, offs)
    sys.exit(0)

for name, shape in (("32 x [1500 x 768]", (32, 1500, 768, 55, 0.5)), ("32 x [1200 x 512]", (32, 1200, 512, 56, 0.5)), ("2000 x [2250 x 128]", (2000, 2250, 128, 44, 0.6))):
    mu, cov, songs, offs = make(*shape)
    res = {}
    for knob in ("0", "1"):
        os.environ["FAD_SONG_SCALED"] = knob
        for _ in range(2): hip.frechet_batched(mu, cov, songs, offs)
        ms = []
        for _ in range(5):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sc, st = hip.frechet_batched(mu, cov, songs, offs)
            torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
        res[knob] = (float(np.median(ms)), sc, st)
    os.environ["FAD_SONG_FAST"] = "0"
    sc64, st64 = hip.frechet_batched(mu, cov, songs[: 8 * shape[1]], offs[:9])
    del os.environ["FAD_SONG_FAST"]
    d01 = float(np.max(np.abs(res["0"][1] - res["1"][1]) / np.abs(res["0"][1])))
    d64 = float(np.max(np.abs(res["1"][1][:8] - sc64) / np.abs(sc64)))
    print(f"{name}: plain {res['0'][0]:.3f} ms, scaled {res['1'][0]:.3f} ms; ok {int((res['1'][2] == 0).sum())}/{shape[0]}; max rel diff plain/scaled {d01:.2e}, scaled/float64 routes {d64:.2e}", flush=True)
    key = str(shape[2])
    for knob in ("0", "1"):
        env = dict(os.environ, FAD_SONG_SCALED=knob, FAD_FAST_TRACE="1")
        r = subprocess.run([sys.executable, __file__, "trace", key], env=env, capture_output=True, text=True)
        its = [int(m) for m in re.findall(r"status 1 iters (\d+)", r.stderr)]
        bad = len(re.findall(r"status [^1] ", r.stderr))
        if its: print(f"    FAD_SONG_SCALED={knob}: iterations per accepted song min {min(its)} median {int(np.median(its))} max {max(its)} ({len(its)} songs, {bad} not accepted)", flush=True)
