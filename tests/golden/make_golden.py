#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Run in the authoring container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's functions (fadtk/fad.py:42-120, 304-395; fadtk/utils.py:13-46) are imported
unmodified through ``_ref_import`` and fed the synthetic inputs of ``recipes.py``.  Only
inputs' checksums and the reference's OUTPUTS are written -- no reference source.

Fixture groups:
  G1 calc_embd_statistics: mu (dtype + values) and cov for fp16/fp32/fp64, 3 shapes
  G2 calc_frechet_distance scalars: iid, identical, decaying spectrum, rank-deficient, shifted
  G3 calculate_embd_statistics_online over ragged fp16 files (+ the 1-row-file NaN quirk)
  G4 score_individual: exact CSV text (ordering, ',' -> '_')
  G5 load_stats: resolution order and dtypes of the cache it writes
  G6 score_inf with np.random.seed(0)
  G7 config-3 scalar (N=100000, D=512, seeds 10/11)
  G8 config-5 shape: D=768 baseline, 64 two-row songs
  G10 hard spectra for the square root: eval sets of N < D frames with power-law spectra (k^-2 .. k^-4), full-rank
      power-law pairs, covariances that went through float32 (near-singular products with eigenvalues at roundoff level)
"""
from __future__ import annotations

import json
import shutil
import sys
import tempfile
import time
import types
from pathlib import Path

import numpy as np
import scipy

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import recipes as R                      # noqa: E402
from _ref_import import import_reference  # noqa: E402

ref_fad, ref_utils = import_reference()
import logging                            # noqa: E402
logging.disable(logging.CRITICAL)

OUT_JSON = {}
meta = {"numpy": np.__version__, "scipy": scipy.__version__,
        "reference": "microsoft/fadtk v1.1.0 (mounted 2025-09-19)",
        "generated_by": "tests/golden/make_golden.py"}


def f(x):
    return float(np.real(x))


# ---------------------------------------------------------------- G1
def g1():
    arrays = {}
    cases = []
    k = 0
    for (n, d) in ((2, 8), (257, 16), (1024, 128)):
        for dt in ("float16", "float32", "float64"):
            x = R.normal_rows(100 + k, n, d, 1.3, 0.25 * (1 + (k % 3)), dtype=np.dtype(dt))
            mu, cov = ref_fad.calc_embd_statistics(x)
            arrays[f"mu{k}"] = mu
            arrays[f"cov{k}"] = cov
            cases.append({"id": k, "seed": 100 + k, "n": n, "d": d, "dtype": dt, "scale": 1.3,
                          "shift": 0.25 * (1 + (k % 3)), "mu_dtype": str(mu.dtype),
                          "cov_dtype": str(cov.dtype), "in_checksum": R.checksum(x)})
            k += 1
    np.savez_compressed(HERE / "g1_stats.npz", **arrays)
    OUT_JSON["g1"] = cases


# ---------------------------------------------------------------- G2
def _fd(a, b):
    m1, c1 = ref_fad.calc_embd_statistics(a)
    m2, c2 = ref_fad.calc_embd_statistics(b)
    return f(ref_fad.calc_frechet_distance(m1, c1, m2, c2)), f(np.trace(c1)), f(np.trace(c2))


def g2():
    out = {}
    a, b = R.c1_pair()
    out["c1_iid"] = dict(zip(("fad", "tr1", "tr2"), _fd(a, b)))
    out["c1_iid"]["in_checksum"] = [R.checksum(a), R.checksum(b)]
    out["identical"] = dict(zip(("fad", "tr1", "tr2"), _fd(a, a)))
    a32, b32 = R.c1_pair(np.float32)
    out["c1_iid_f32"] = dict(zip(("fad", "tr1", "tr2"), _fd(a32, b32)))
    a64, b64 = R.c1_pair(np.float64)
    out["c1_iid_f64"] = dict(zip(("fad", "tr1", "tr2"), _fd(a64, b64)))
    sa, sb = R.shifted_pair()
    out["shifted"] = dict(zip(("fad", "tr1", "tr2"), _fd(sa, sb)))
    # the same at 60000 rows: np.mean's float32 running sum (fad.py:48) is then 1e-5 off the exact column sums and the float16 means
    # differ from the rounded exact means in some dimensions (round 4: fad_moments_set_reference_mean reproduces numpy's)
    la, lb = R.shifted_pair(n=60000)
    out["shifted_long"] = dict(zip(("fad", "tr1", "tr2"), _fd(la, lb)))
    out["shifted_long"]["in_checksum"] = [R.checksum(la), R.checksum(lb)]
    for d in (64, 512):
        x1 = R.decaying_rows(30, 4 * d, d, basis_seed=40)
        x2 = R.decaying_rows(31, 4 * d, d, basis_seed=40, gain=1.1)
        x3 = R.decaying_rows(32, 4 * d, d, basis_seed=41)
        out[f"decay_same_basis_d{d}"] = dict(zip(("fad", "tr1", "tr2"), _fd(x1, x2)))
        out[f"decay_diff_basis_d{d}"] = dict(zip(("fad", "tr1", "tr2"), _fd(x1, x3)))
    # rank-deficient eval against a full-rank float64 baseline (per-song shape)
    for d, rows in ((128, 2), (128, 10), (128, 50), (768, 2), (768, 10)):
        mu_b, cov_b = R.baseline_stats(50 + d, 4 * d, d)
        s = R.songs(60 + rows, 1, rows, d)[0]
        mu_s, cov_s = ref_fad.calc_embd_statistics(s)
        out[f"rankdef_d{d}_n{rows}"] = {"fad": f(ref_fad.calc_frechet_distance(mu_b, cov_b, mu_s, cov_s)),
                                         "in_checksum": [R.checksum(cov_b), R.checksum(s)]}
    OUT_JSON["g2"] = out


# ---------------------------------------------------------------- G3
def g3():
    tmp = Path(tempfile.mkdtemp(prefix="fad_g3_"))
    try:
        blocks = R.ragged_files(70, 37, 24)
        files = []
        for i, blk in enumerate(blocks):
            p = tmp / f"f{i:03d}.npy"
            np.save(p, blk)
            files.append(p)
        mu, cov = ref_utils.calculate_embd_statistics_online(files)
        # the one-row-file quirk (Q5)
        one = tmp / "one.npy"
        np.save(one, blocks[0][:1])
        with np.errstate(all="ignore"):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mu_nan, cov_nan = ref_utils.calculate_embd_statistics_online(files[:3] + [one])
        # round 5: long files with |mu| / sigma ~ 7 -- the per-file float16 means (utils.py:16) are numpy's float32 running-sum means
        sblocks = R.shifted_files(71, 12, 32, min_rows=9000, max_rows=40000)      # (float16 frames near 7 are multiples of 2^-8: the float32 sum is exact below 65536, i.e. ~9400 rows)
        assert sum(int((b.mean(axis=0) != b.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float16)).sum()) for b in sblocks) > 0
        sfiles = []
        for i, blk in enumerate(sblocks):
            p = tmp / f"s{i:03d}.npy"
            np.save(p, blk)
            sfiles.append(p)
        mu_s, cov_s = ref_utils.calculate_embd_statistics_online(sfiles)
        OUT_JSON["g3_shifted"] = {"seed": 71, "n_files": 12, "d": 32, "min_rows": 9000, "max_rows": 40000, "sizes": [int(b.shape[0]) for b in sblocks],
                                  "in_checksum": float(sum(R.checksum(b) for b in sblocks))}
        np.savez_compressed(HERE / "g3_online.npz", mu=mu, cov=cov, mu_nan=mu_nan,
                            cov_nan_isnan=np.isnan(cov_nan), mu_shifted=mu_s, cov_shifted=cov_s)
        OUT_JSON["g3"] = {"seed": 70, "n_files": 37, "d": 24,
                          "sizes": [int(b.shape[0]) for b in blocks],
                          "in_checksum": float(sum(R.checksum(b) for b in blocks)),
                          "mu_dtype": str(mu.dtype), "cov_dtype": str(cov.dtype),
                          "cov_nan_all": bool(np.isnan(cov_nan).all())}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------- G4 / G5
def _fake_loader(name):
    return types.SimpleNamespace(name=name, sr=16000, num_features=32, load_model=lambda: None)


def g4_g5():
    root = Path(tempfile.mkdtemp(prefix="fad_g4_"))
    try:
        model = "toy-model"
        d = 32
        evald = root / "evalset"
        (evald / "embeddings" / model).mkdir(parents=True)
        names = ["alpha.wav", "be,ta.wav", "gamma.flac", "delta.mp3", "eps.wav", "zeta.ogg", "short.wav"]
        song_rows = R.songs(80, len(names), [5, 9, 2, 33, 12, 7, 1], d)
        for nm, rows in zip(names, song_rows):
            (evald / nm).write_bytes(b"")                    # the "audio file" only has to exist
            np.save(evald / "embeddings" / model / (Path(nm).stem + ".npy"), rows)
        mu_b, cov_b = R.baseline_stats(81, 400, d)
        npz = root / "base.npz"
        np.savez(npz, **{f"{model}.mu": mu_b, f"{model}.cov": cov_b})

        fad = ref_fad.FrechetAudioDistance(_fake_loader(model), audio_load_worker=2, load_model=False)
        csv = root / "indiv.csv"
        import io
        import contextlib
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fad.score_individual(str(npz), evald, csv)
        text = csv.read_text().replace(str(root), "{ROOT}")
        order = sorted(p.name for p in evald.glob("*.*"))
        OUT_JSON["g4"] = {"model": model, "d": d, "names": names, "rows": [5, 9, 2, 33, 12, 7, 1],
                          "songs_seed": 80, "base_seed": 81, "base_n": 400, "csv": text,
                          "glob_sorted": order}

        # round 5: songs of many frames with |mu| / sigma ~ 7 -- np.mean of a float16 song (fad.py:377 -> :48) is the float32 running-sum mean
        evs = root / "evalshift"
        (evs / "embeddings" / model).mkdir(parents=True)
        srows = R.shifted_files(83, 6, d, min_rows=12000, max_rows=30000)
        assert sum(int((b.mean(axis=0) != b.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float16)).sum()) for b in srows) > 0
        snames = [f"s{i}.wav" for i in range(len(srows))]
        for nm, rows in zip(snames, srows):
            (evs / nm).write_bytes(b"")
            np.save(evs / "embeddings" / model / (Path(nm).stem + ".npy"), rows)
        rngb = np.random.default_rng(84)
        xb = rngb.standard_normal((4000, d)) * (1.0 + 0.3 * rngb.random(d)) + 7.0
        mu_sb, cov_sb = xb.mean(axis=0), np.cov(xb, rowvar=False)
        npz_s = root / "base_shift.npz"
        np.savez(npz_s, **{f"{model}.mu": mu_sb, f"{model}.cov": cov_sb})
        csv_s = root / "indiv_shift.csv"
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            fad.score_individual(str(npz_s), evs, csv_s)
        OUT_JSON["g4_shifted"] = {"model": model, "d": d, "names": snames, "rows": [int(r.shape[0]) for r in srows], "songs_seed": 83, "min_rows": 12000, "max_rows": 30000,
                                  "base_seed": 84, "base_n": 4000, "csv": csv_s.read_text().replace(str(root), "{ROOT}")}

        # G5: load_stats on a directory computes + caches; dtypes of what it wrote
        with contextlib.redirect_stdout(io.StringIO()):
            mu_d, cov_d = fad.load_stats(evald) if False else (None, None)
        # (the eval dir holds a 1-row file -> NaN cov; use a clean dir for G5)
        clean = root / "clean"
        (clean / "embeddings" / model).mkdir(parents=True)
        blocks = R.ragged_files(82, 9, d)
        for i, blk in enumerate(blocks):
            np.save(clean / "embeddings" / model / f"c{i}.npy", blk)
        with contextlib.redirect_stdout(io.StringIO()):
            mu_c, cov_c = fad.load_stats(clean)
            mu_again, cov_again = fad.load_stats(clean)           # now from the cache
            mu_n, cov_n = fad.load_stats(str(npz))
        wrote = sorted(p.name for p in (clean / "stats" / model).glob("*"))
        OUT_JSON["g5"] = {"seed": 82, "n_files": 9, "d": d, "cache_files": wrote,
                          "mu_dtype": str(np.load(clean / "stats" / model / "mu.npy").dtype),
                          "cov_dtype": str(np.load(clean / "stats" / model / "cov.npy").dtype),
                          "cache_roundtrip_equal": bool(np.array_equal(mu_c, mu_again) and np.array_equal(cov_c, cov_again)),
                          "npz_returns_stored": bool(np.array_equal(mu_n, mu_b) and np.array_equal(cov_n, cov_b)),
                          "fad_clean_vs_npz": f(ref_fad.calc_frechet_distance(mu_n, cov_n, mu_c, cov_c)),
                          "glob_order_dependent": True}
        # the reference globs in filesystem order; store the order it saw so the test can mirror it
        OUT_JSON["g5"]["glob_order"] = [p.name for p in (clean / "embeddings" / model).glob("*.npy")]
        np.savez_compressed(HERE / "g5_stats.npz", mu=mu_c, cov=cov_c)
    finally:
        shutil.rmtree(root, ignore_errors=True)


# ---------------------------------------------------------------- G6
def g6():
    d = 32
    mu_b, cov_b = R.baseline_stats(90, 600, d)
    rows = R.normal_rows(91, 2000, d, 1.1, 0.05)
    tmp = Path(tempfile.mkdtemp(prefix="fad_g6_"))
    try:
        npz = tmp / "b.npz"
        np.savez(npz, **{"toy.mu": mu_b, "toy.cov": cov_b})
        files = []
        for i in range(4):
            p = tmp / f"e{i}.npy"
            np.save(p, rows[i * 500:(i + 1) * 500])
            files.append(p)
        fad = ref_fad.FrechetAudioDistance(_fake_loader("toy"), load_model=False)
        import io
        import contextlib
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()):
            res = fad.score_inf(str(npz), files)
        OUT_JSON["g6"] = {"d": d, "base_seed": 90, "base_n": 600, "rows_seed": 91, "n": 2000,
                          "score": f(res.score), "slope": f(res.slope), "r2": f(res.r2),
                          "points": [[int(n), f(v)] for n, v in res.points]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ---------------------------------------------------------------- G7
def g7():
    a, b = R.c3_pair()
    t0 = time.time()
    m1, c1 = ref_fad.calc_embd_statistics(a)
    m2, c2 = ref_fad.calc_embd_statistics(b)
    t1 = time.time()
    fad = ref_fad.calc_frechet_distance(m1, c1, m2, c2)
    t2 = time.time()
    d = (m1 - m2)
    OUT_JSON["g7"] = {"n": 100000, "d": 512, "seeds": [10, 11], "fad": f(fad), "tr1": f(np.trace(c1)),
                      "tr2": f(np.trace(c2)), "mean_term": f(d.dot(d)), "mean_term_dtype": str(d.dot(d).dtype),
                      "tr_sqrt": f((d.dot(d) + np.trace(c1) + np.trace(c2) - fad) / 2),
                      "in_checksum": [R.checksum(a), R.checksum(b)],
                      "ref_seconds_here": {"stats_x2": t1 - t0, "frechet": t2 - t1}}


# ---------------------------------------------------------------- G8
def g8():
    d = 768
    mu_b, cov_b = R.baseline_stats(95, 3 * d, d)
    sg = R.songs(96, 64, 2, d)
    scores = []
    for s in sg:
        mu_s, cov_s = ref_fad.calc_embd_statistics(s)
        scores.append(f(ref_fad.calc_frechet_distance(mu_b, cov_b, mu_s, cov_s)))
    # a few multi-frame songs at D=128 (VGGish-like 10 s clips -> 10 frames; and n > D)
    d2 = 128
    mu_b2, cov_b2 = R.baseline_stats(97, 4 * d2, d2)
    sg2 = R.songs(98, 12, [10, 3, 50, 200, 2, 129], d2)
    scores2 = []
    for s in sg2:
        mu_s, cov_s = ref_fad.calc_embd_statistics(s)
        scores2.append(f(ref_fad.calc_frechet_distance(mu_b2, cov_b2, mu_s, cov_s)))
    OUT_JSON["g8"] = {"d": d, "base_seed": 95, "base_n": 3 * d, "songs_seed": 96, "n_songs": 64,
                      "rows": 2, "scores": scores,
                      "multi": {"d": d2, "base_seed": 97, "base_n": 4 * d2, "songs_seed": 98,
                                "rows": [10, 3, 50, 200, 2, 129], "n_songs": 12, "scores": scores2}}


# ---------------------------------------------------------------- G10
def g10():
    out = {}
    for p in (2.0, 3.0, 4.0):
        base = R.decaying_rows(50, 4096, 256, 52, power=p)
        song = R.decaying_rows(51, 150, 256, 53, power=p)
        mb, cb = ref_fad.calc_embd_statistics(base)
        ms, cs = ref_fad.calc_embd_statistics(song)
        out[f"short_eval_d256_n150_p{p:g}"] = {"fad": f(ref_fad.calc_frechet_distance(mb, cb, ms, cs)), "power": p,
                                               "in_checksum": [R.checksum(base), R.checksum(song)]}
    for p in (3.0, 4.0):
        a = R.decaying_rows(30, 4096, 128, 32, power=p)
        b = R.decaying_rows(31, 4096, 128, 33, power=p, gain=1.05)
        out[f"fullrank_d128_p{p:g}"] = dict(zip(("fad", "tr1", "tr2"), _fd(a, b)))
        m1, c1 = ref_fad.calc_embd_statistics(a)
        m2, c2 = ref_fad.calc_embd_statistics(b)
        c1f, c2f = c1.astype(np.float32).astype(np.float64), c2.astype(np.float32).astype(np.float64)
        out[f"f32cov_d128_p{p:g}"] = {"fad": f(ref_fad.calc_frechet_distance(m1, c1f, m2, c2f)), "power": p}
    OUT_JSON["g10"] = out


if __name__ == "__main__":
    only = set(sys.argv[1:])
    prev = {}
    jpath = HERE / "golden.json"
    if only and jpath.exists():
        prev = json.loads(jpath.read_text())
    for name, fn in (("g1", g1), ("g2", g2), ("g3", g3), ("g4", g4_g5), ("g6", g6), ("g7", g7), ("g8", g8), ("g10", g10)):
        if only and name not in only:
            continue
        t = time.time()
        fn()
        print(f"{name}: {time.time() - t:.1f}s", flush=True)
    prev.update(OUT_JSON)
    prev["meta"] = meta
    jpath.write_text(json.dumps(prev, indent=1))
    print("wrote", jpath)
