"""Randomised sweeps of the HIP path against float64 numpy / the oracle: shapes, dtypes, pitches, chunking, kernel
generations for the moments; dimensions, conditioning, rank and scale for the Frechet distance.  Fixed seeds -- these
are regression nets around the curated cases of test_gpu_parity.py, not new semantics."""
import logging

import numpy as np
import pytest

from oracle import fad_oracle as O

pytestmark = pytest.mark.gpu
logging.getLogger("fad_oracle").setLevel(logging.CRITICAL)


@pytest.mark.parametrize("seed", [1, 7])
def test_fuzz_moments_against_numpy(seed, monkeypatch):
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(seed)
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32, "f64": torch.float64}
    for case in range(60):
        d = int(rng.choice([1, 3, 8, 24, 64, 100, 128, 136, 200, 256, 384, 512, 640, 768]))
        n = int(rng.choice([0, 1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 255, 257, 1000, 4097, 20000, 70001]))
        dt = str(rng.choice(list(tdt)))
        pitch = d + int(rng.choice([0, 0, 8, 3]))
        variant = str(rng.choice(["", "", "4", "8"]))      # (drawn to keep the sequence of cases; the knob itself is gone)
        shift = float(rng.choice([0.0, 0.0, 0.5, 5.0]))
        x64 = rng.standard_normal((n, pitch)) * (0.3 + rng.random()) + shift
        view = torch.from_numpy(x64).to(tdt[dt]).cuda()[:, :d]
        ref = view.double().cpu().numpy()
        with Moments(d) as m:
            chunks = int(rng.choice([1, 1, 2, 3]))
            cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, size=chunks - 1)]))
            if rng.random() < 0.3:
                m.reset()
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                m.update(view[lo:hi])
            p = m.export()
        what = f"case {case}: n={n} d={d} pitch={pitch} dtype={dt} variant={variant!r} shift={shift} cuts={cuts}"
        S, s1 = ref.T @ ref, ref.sum(0)
        half = dt in ("f16", "bf16")                 # fp32-accumulating MFMA kernel vs the exact fp64 kernel
        M = p[1 + d:].reshape(d, d)
        assert p[0] == n, what
        assert np.array_equal(M, M.T), what
        assert np.abs(M - S).max() <= (1e-6 if half else 1e-11) * max(np.abs(S).max(), 1e-30), what
        np.testing.assert_allclose(p[1:1 + d], s1, rtol=2e-7 if half else 1e-12, atol=1e-6 * max(np.abs(s1).max(), 1e-30),
                                   err_msg=what)


def test_fuzz_moments_guard_second_pass(monkeypatch):
    """Columns with |mean| >> std (the shift guard) on float16 rows long enough for the second pass (>= 16 rows per column):
    random shapes, chunkings and column patterns; the covariance must come out to float32-sum accuracy IN UNITS OF
    sigma_i sigma_j for every entry -- raw second moments are 1e2..1e4 times larger here."""
    import torch
    from fadtk_amd.hip import Moments
    import os
    rng = np.random.default_rng(int(os.environ.get("FAD_FUZZ_SEED", "2024")))
    for case in range(18):
        d = int(rng.choice([64, 128, 200, 256, 512, 640]))
        n = int(rng.choice([16 * d, 16 * d + 7, 30000, 70001]))
        sig = 0.2 + rng.random(d)
        x64 = rng.standard_normal((n, d)) * sig
        k = int(rng.choice([1, 3, d // 8, d // 2]))
        cols = rng.choice(d, size=k, replace=False)
        x64[:, cols] += rng.choice([-1.0, 1.0], size=k) * sig[cols] * rng.uniform(10.0, 60.0, size=k)
        if rng.random() < 0.5:
            x64[:, cols[0]] = 7.25                                          # a constant column
        x = torch.from_numpy(x64).to(torch.float16).cuda()
        ref = x.double().cpu().numpy()
        with Moments(d) as m:
            cuts = sorted(set([0, n] + [int(c) for c in rng.integers(16 * d, n + 1, size=int(rng.choice([0, 1, 2])))]))
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi - lo > 0:
                    m.update(x[lo:hi])
            mu, cov, cnt = m.finalize()
        what = f"case {case}: n={n} d={d} outlier columns {k} cuts={cuts}"
        want = np.cov(ref, rowvar=False)
        sd = np.sqrt(np.diag(want))
        scale = np.outer(sd, sd)
        live = sd > 0
        assert cnt == n, what
        np.testing.assert_allclose(mu, ref.mean(0), rtol=1e-6, atol=1e-7, err_msg=what)
        err = np.abs(cov - want)
        assert (err[np.ix_(live, live)] / scale[np.ix_(live, live)]).max() <= 5e-6, what
        assert err[~live].max(initial=0.0) <= 1e-9 and err[:, ~live].max(initial=0.0) <= 1e-9, what


def _random_cov(rng, d, n, decay, scale):
    """Sample covariance of n rows with spectrum ~ k^-decay (n <= d gives a rank-deficient matrix)."""
    lam = np.arange(1, d + 1, dtype=np.float64) ** (-decay)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    x = (rng.standard_normal((n, d)) * np.sqrt(lam)) @ q.T * scale
    return x.mean(0), np.cov(x, rowvar=False)


def test_fuzz_frechet_against_oracle(F=None):
    import fadtk_amd, os
    rng = np.random.default_rng(int(os.environ.get("FAD_FUZZ_SEED", "11")))
    worst = 0.0
    for case in range(28):
        d = int(rng.choice([2, 7, 32, 64, 96, 128, 200, 256, 384]))
        decay = float(rng.choice([0.0, 0.5, 1.0, 1.5]))
        n1 = int(rng.choice([4 * d + 3, 2 * d, d + 5]))
        n2 = int(rng.choice([4 * d + 3, d + 5, max(2, d // 2), 3]))          # the last two: rank-deficient evaluation sets
        s1, s2 = float(rng.choice([1.0, 1e-3, 30.0])), float(rng.choice([1.0, 1.0, 0.7, 1.3]))
        mu1, c1 = _random_cov(rng, d, n1, decay, s1)
        mu2, c2 = _random_cov(rng, d, n2, decay, s1 * s2)
        want = O.frechet_distance(mu1, c1, mu2, c2, run_sqrtm=False)
        got = fadtk_amd.calc_frechet_distance(mu1, c1, mu2, c2)
        what = f"case {case}: d={d} decay={decay} n1={n1} n2={n2} scale={s1}x{s2} want={want:.6e} got={got:.6e}"
        # FAD is a cancellation: compare relative to the terms that are summed.  With a rank-deficient set the
        # reference itself is only good to ~sqrt(machine eps): eig returns the zero eigenvalues of C1 C2 as +-1e-16
        # and takes their square roots (fad.py:91-92).
        terms = np.trace(c1) + np.trace(c2)
        full_rank = min(n1, n2) > d
        err = abs(got - want) / terms
        print(f"{what} err/terms={err:.2e}")
        # (full rank: 1e-10 of the terms on the float64 route; since round 5 barely-full-rank products -- n = d + 5 rows -- may stay on the
        #  low-precision chain, whose verification accepts what it estimates at < 4e-6 of the DISTANCE: seen 3e-9 of the terms)
        assert err <= (2e-8 if full_rank else 1e-6), what
        assert abs(got - want) <= 1e-4 * abs(want) + 1e-10 * terms, what
        worst = max(worst, err if full_rank else 0.0)
    assert worst < 2e-8


def test_fuzz_per_song_scores_fast_chain_against_float64_routes_and_oracle(monkeypatch):
    """The batched per-song call on songs of at least D + 1 float16 frames, D in {128, 256, 384}: the low-precision chain (default)
    against the float64 routes (FAD_SONG_FAST=0) on the same call and against the oracle (fad.py:373-378) -- with overall scales
    from 1e-2 to 30, columns whose mean is far above their spread (the shift of the float16 covariances), a few columns carrying
    most of the variance, songs barely above D frames and long ones, and batches on either side of the big-tile threshold."""
    from fadtk_amd import hip
    import os
    rng = np.random.default_rng(int(os.environ.get("FAD_FUZZ_SEED", "23")))
    for case in range(10):
        d = int(rng.choice([128, 256, 384]))
        nsongs = int(rng.choice([3, 9, 17]))
        scale = float(rng.choice([1e-2, 1.0, 30.0]))
        col = (0.5 + rng.random(d)) * scale
        if case % 3 == 0:
            col[: d // 16] *= 6.0                                       # a few dominant columns
        mean = np.where(rng.random(d) < 0.15, 8.0, 0.1) * col * rng.standard_normal(d)     # |mean| >> std on ~15 % of the columns
        base = rng.standard_normal((6 * d, d)) * col * 1.1 + mean
        mu_b, cov_b = base.mean(0), np.cov(base, rowvar=False)
        frames = [int(rng.choice([d + 1, d + 7, 2 * d, 5 * d, 4100 if d == 128 else 3 * d])) for _ in range(nsongs)]
        # (every fifth case in float32: those frames keep the float64-MFMA covariances and the two-pass statistics in front of the chain)
        dt = np.float32 if case % 5 == 4 else np.float16
        songs = [(rng.standard_normal((n, d)) * col * (0.8 + 0.4 * rng.random()) + mean * (1.0 + 0.05 * rng.standard_normal())).astype(dt)
                 for n in frames]
        rows = np.concatenate(songs)
        offs = np.concatenate([[0], np.cumsum(frames)])
        want = np.array(O.individual_scores(mu_b, cov_b, songs, run_sqrtm=False), dtype=np.float64)
        monkeypatch.delenv("FAD_SONG_FAST", raising=False)
        fast, st_fast = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
        monkeypatch.setenv("FAD_SONG_FAST", "0")
        f64, st_f64 = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
        what = f"case {case}: d={d} songs={nsongs} scale={scale} dtype={np.dtype(dt).name} frames={frames}"
        assert (st_fast == 0).all() and (st_f64 == 0).all(), what
        # a song of D + 1 frames has a (numerically) singular covariance: the reference's eig is good to ~sqrt(eps) there
        tol = np.where(np.array(frames) < d + 16, 2e-5, 2e-6)
        rel_o = np.abs(fast - want) / np.abs(want)
        rel_f = np.abs(fast - f64) / np.abs(f64)
        print(f"{what} max rel vs oracle {rel_o.max():.2e} vs float64 routes {rel_f.max():.2e}")
        # float32 frames: the reference's np.mean (fad.py:48) adds the rows one after the other in float32 -- 1e-6 off the rounded exact
        # mean, which cost up to 5e-5 of a small score until round 4; the statistics kernels now walk the column in numpy's order
        # (frechet_songs.hip: mean_like_reference), so float32 frames meet the same bound as float16 ones
        assert (rel_o <= tol).all(), (what, rel_o)
        assert (rel_f <= tol).all(), (what, rel_f)


def test_fuzz_wide_chain_decaying_pairs_against_oracle():
    """Round 5's wide chain (scaled low-precision steps + a-posteriori verification) on the spectra it was built for, drawn at random:
    D in {256, 384, 512, 768}, covariance spectra k^-p with p in [0.2, 1.7] (different exponents for the two sets, different random bases
    for some pairs), sample sizes from 1.5 D to 60 D, overall scales 1e-3 .. 30, mean offsets -- single pairs through the blocking call,
    then the same pairs as batches of up to eight through fad_frechet_from_moments_multi_begin.  Whatever route a pair takes (chain,
    verified chain, float64), the distance must agree with the reference's eig formula (fad.py:91-92) to 1e-6 -- a hundredth of the
    north star's tolerance -- and the batch with the single scores."""
    import torch
    from fadtk_amd import hip, _capi as K
    import os
    rng = np.random.default_rng(int(os.environ.get("FAD_FUZZ_SEED", "505")))      # (soaks: FAD_FUZZ_SEED=<n> pytest -k wide_chain)
    pairs, info = [], []
    for case in range(24):
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
        pairs.append((a, b)); info.append(f"case {case}: d={d} p=({p1:.2f},{p2:.2f}) n=({n1},{n2}) scale={scale} off={off}")
    worst, routes, by_d = 0.0, [], {}
    for k, ((a, b), what) in enumerate(zip(pairs, info)):
        d = a.shape[1]
        ma, mb = hip.Moments(d), hip.Moments(d)
        ma.set_reference_mean(True); mb.set_reference_mean(True)      # numpy's own float16 means (fad.py:48), as calc_embd_statistics carries them
        ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
        got, dg = hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)   # ... and the reference's float16 mean term
        mu_a, cov_a = O.embd_statistics(a); mu_b, cov_b = O.embd_statistics(b)
        want = O.frechet_distance(mu_a, cov_a, mu_b, cov_b, run_sqrtm=False)
        err = abs(got - want) / abs(want)
        routes.append(int(dg["route"]))
        print(f"{what} want={want:.6e} got={got:.6e} rel={err:.1e} route={dg['route']} iterations={dg['iters']}")
        assert err <= 1e-6, what
        worst = max(worst, err)
        by_d.setdefault(d, []).append((ma, mb, got))
    assert 2 in routes, "none of the pairs stayed on the chain: the case list no longer exercises it"
    for d, lst in by_d.items():                                       # the same pairs, batched
        for lo in range(0, len(lst), 8):
            grp = lst[lo:lo + 8]
            res = hip.FrechetMultiJob([(ma, mb) for ma, mb, _ in grp], mean_dtype=K.FAD_F16).result()
            for (f, dg), (_, _, single) in zip(res, grp):
                assert abs(f - single) <= 2e-6 * abs(single), (d, f, single, dg)
    for lst in by_d.values():
        for ma, mb, _ in lst:
            ma.close(); mb.close()
    print("worst relative error", worst)


def test_fuzz_running_sum_mean_on_hostile_columns():
    """np.mean's float32 running sum (fadtk/fad.py:48) bit for bit on columns built to break a re-ordered or wider sum: sums that hover
    around a power of two (the rounding unit changes back and forth), exact ties (x an odd multiple of half the unit: round-to-even
    depends on the running sum's last bit), cancellation of large values of both signs, a column that overflows float16's range in the
    sum but not float32's, denormal-sized frames, and Inf / NaN entries (numpy propagates them: so must the walk).  Device rows and
    host rows (the latter in pieces), one update and three."""
    import torch
    from fadtk_amd import hip
    rng = np.random.default_rng(77)
    n, d = 50000, 512
    x = (rng.standard_normal((n, d)) * 0.5 + 0.25).astype(np.float32)
    x[:, 0] = np.where(np.arange(n) % 2 == 0, 1024.0, -1023.5)                      # hovers around 2^k for ever larger k ... slowly upward
    x[:, 1] = 0.0009765625 * (2 * rng.integers(0, 8, n) + 1)                        # odd multiples of 2^-10: ties once the sum passes 2^14
    x[:, 2] = rng.choice([60000.0, -60000.0], n)                                    # cancellation at the top of float16's range
    x[:, 3] = 60000.0                                                               # the sum reaches 3e9
    x[:, 4] = 6e-8 * rng.integers(1, 5, n)                                          # float16 denormals
    x[:, 5] = rng.standard_normal(n) * 1e-3 + 2048.0                                # |mean| / std = 2e6
    x[:, 6] = np.where(rng.random(n) < 0.5, 0.5, -0.5) + 4096.0 * (np.arange(n) == 17)
    x16 = x.astype(np.float16)
    x16[40000, 7] = np.inf
    x16[123, 8] = np.nan
    x16[30000, 9] = np.inf; x16[30001, 9] = -np.inf                                 # Inf - Inf = NaN from then on
    with np.errstate(all="ignore"):
        want = np.mean(x16, axis=0)
    assert want.dtype == np.float16 and np.isnan(want[8]) and np.isnan(want[9]) and np.isinf(want[7])
    for rows, cuts in ((torch.from_numpy(x16).cuda(), (n,)), (torch.from_numpy(x16).cuda(), (7, 20001, n)), (x16, (n,)), (x16, (33333, n))):
        with hip.Moments(d) as acc:
            acc.set_reference_mean(True)
            lo = 0
            for hi in cuts:
                acc.update(rows[lo:hi]); lo = hi
            mu, _, cnt = acc.finalize()
        assert cnt == n
        with np.errstate(all="ignore"):
            got = mu.astype(np.float32).astype(np.float16)
        np.testing.assert_array_equal(got, want)                                    # (assert_array_equal treats NaN == NaN, Inf == Inf)
