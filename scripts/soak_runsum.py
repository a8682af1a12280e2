#!/usr/bin/env python3
"""GPU soak: np.mean's float32 running column sums (fad_moments_set_reference_mean) bit for bit over random shapes -- rows 1 .. 60000 (below, at and
above the walk's 192-row tiles), D 8 .. 1024 (multiples of 8 and not), float16 / bfloat16 / float32 frames, pitched and misaligned views (the
fallback kernels), one to four updates per handle, device and host rows, several matrices per launch (update_multi).  Prints the failures."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from fadtk_amd import hip

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 120
rng = np.random.default_rng(seed)
tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
bad = 0
for case in range(cases):
    d = int(rng.choice([8, 16, 24, 96, 100, 128, 136, 256, 384, 512, 520, 768, 1024]))
    n = int(rng.choice([1, 2, 15, 16, 17, 191, 192, 193, 383, 384, 385, 1000, 4097, 20000, 60000]))
    dt = str(rng.choice(["f16", "f16", "f16", "bf16", "f32"]))
    pitch = d + int(rng.choice([0, 0, 0, 8, 3]))
    off = float(rng.choice([0.0, 0.5, 3.0, -7.0]))
    x = torch.from_numpy(rng.standard_normal((n + 1, pitch)) * (0.2 + rng.random()) + off).to(tdt[dt])
    skew = int(rng.choice([0, 0, 0, 1]))                    # a view that starts one element in: rows no longer 16-byte aligned
    view = x[:n, skew:skew + d] if skew + d <= pitch else x[:n, :d]
    host = bool(rng.random() < 0.25) and dt != "bf16"
    want = np.mean(view.float().numpy() if dt == "bf16" else view.numpy(), axis=0, dtype=np.float32) if dt != "f16" else None
    if dt == "f16":
        want16 = np.mean(view.numpy(), axis=0)              # numpy: float32 running sum, float32 quotient, cast
    elif dt == "f32":
        want32 = np.mean(view.numpy(), axis=0)
    else:                                                   # bfloat16 has no numpy twin: the float32 running sum of the widened values
        acc = np.zeros(d, np.float32)
        for r in view.float().numpy():
            acc += r
        want32 = (acc.astype(np.float64) / n).astype(np.float32)
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=int(rng.choice([0, 0, 1, 3])))]))
    multi = (not host) and rng.random() < 0.3 and len(cuts) == 2
    with hip.Moments(d) as m, hip.Moments(d) as m2:
        m.set_reference_mean(True); m2.set_reference_mean(True)
        src = view.numpy() if host else view.cuda()
        if multi:
            hip.Moments.update_multi([m, m2], [src, src])
        else:
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi > lo:
                    m.update(src[lo:hi])
        mu, _, cnt = m.finalize() if n >= 2 else (None, None, n)
        if n < 2:
            continue
        got32 = mu.astype(np.float32)
        ok = np.array_equal(got32.astype(np.float16), want16) if dt == "f16" else np.array_equal(got32, want32)
        if multi:
            mu2, _, _ = m2.finalize()
            ok = ok and np.array_equal(mu2, mu)
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} d={d} pitch={pitch} skew={skew} dtype={dt} off={off} host={host} cuts={cuts} multi={multi}")
print(f"seed {seed}: {cases} cases, {bad} mismatches")
