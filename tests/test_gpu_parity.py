"""Parity of the HIP path (through the C ABI) against the oracle and the reference's golden vectors.

Bar (BASELINE.json north_star): |FAD_gpu - FAD_ref| / |FAD_ref| <= 1e-4.  The checks below are
tighter wherever the arithmetic allows, so a regression shows long before the bar is hit.
"""
import logging

import contextlib

import numpy as np
import pytest

import recipes as R
from oracle import fad_oracle as O

pytestmark = pytest.mark.gpu
logging.getLogger("fad_oracle").setLevel(logging.CRITICAL)

FAD_BAR = 1e-4


@pytest.fixture(scope="module")
def F():
    import fadtk_amd
    from fadtk_amd import _capi
    _capi.require_gpu(0)
    return fadtk_amd


def structured_rows(seed, n, d, dtype):
    """Column-dependent gains, shifts and cross-correlations: any column permutation or transposition
    in the kernel's fragment/epilogue maps changes the expected covariance."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((n, d))
    gains = 0.5 + np.arange(d) / d
    x = x * gains + 0.3 * np.sin(np.arange(d))
    x[:, 1:] += 0.25 * x[:, :-1]                    # asymmetric neighbour coupling
    return x.astype(dtype)


# --------------------------------------------------------------------------------- moments
@pytest.mark.parametrize("n,d", [(2, 8), (257, 16), (1024, 128), (3000, 512), (777, 200), (4099, 136), (50, 768)])
@pytest.mark.parametrize("dtype", [np.float16, np.float32, np.float64])
def test_moments_match_np_cov(F, n, d, dtype):
    x = structured_rows(n * 7 + d, n, d, dtype)
    mu, cov = F.calc_embd_statistics(x)
    mu_o, cov_o = O.embd_statistics(x)
    assert mu.dtype == mu_o.dtype and cov.dtype == np.float64
    scale = np.abs(cov_o).max()
    tol = 2e-6 if dtype == np.float16 else 1e-11            # fp16 path sums bounded runs in fp32
    np.testing.assert_allclose(cov, cov_o, rtol=0, atol=tol * scale)
    np.testing.assert_array_equal(cov, cov.T)                # exactly symmetric, like dsyrk
    if dtype == np.float64:
        np.testing.assert_allclose(mu, mu_o, rtol=1e-13, atol=1e-15)
    else:                                                    # numpy's own mean, bit for bit: the rows added one after the other in float32
        np.testing.assert_array_equal(mu, mu_o)              # (round 4: fad_moments_set_reference_mean; before: the rounded exact mean,
                                                             #  within one float16 ulp of numpy's and equal to it in > 95 % of the columns)


def test_moments_golden_g1(F, golden, golden_dir):
    z = np.load(golden_dir / "g1_stats.npz")
    for c in golden["g1"]:
        x = R.normal_rows(c["seed"], c["n"], c["d"], c["scale"], c["shift"], dtype=np.dtype(c["dtype"]))
        mu, cov = F.calc_embd_statistics(x)
        assert str(mu.dtype) == c["mu_dtype"]
        ref = z[f"cov{c['id']}"]
        np.testing.assert_allclose(cov, ref, rtol=0, atol=(2e-6 if c["dtype"] == "float16" else 1e-11) * np.abs(ref).max())


def test_moments_alignment_and_pitch_fallbacks(F):
    from fadtk_amd.hip import Moments
    big = structured_rows(5, 600, 136, np.float16)
    views = {"pitched": big[:, :128],            # ld=136 > d=128, 16-B aligned rows -> MFMA kernel
             "odd_offset": big[:, 1:129],        # rows start 2 B off a 16-B boundary -> generic kernel
             "odd_width": big[:, :131]}          # d % 8 != 0 -> generic kernel
    for name, v in views.items():
        with Moments(v.shape[1]) as m:
            m.update(v)
            mu, cov, n = m.finalize()
        mu_o, cov_o = O.embd_statistics(np.ascontiguousarray(v))
        assert n == 600
        np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max(), err_msg=name)
        np.testing.assert_allclose(mu, mu_o.astype(np.float64), rtol=0, atol=1e-3, err_msg=name)


def test_moments_mfma_and_generic_kernels_agree(F, monkeypatch):
    from fadtk_amd.hip import Moments
    x = structured_rows(9, 5000, 256, np.float16)
    with Moments(256) as m:
        m.set_timing(True)
        m.update(x)
        _, _, variant = m.last_timing()
        p_mfma = m.export()
    assert variant == 0
    monkeypatch.setenv("FAD_MOMENTS_FORCE_GENERIC", "1")
    with Moments(256) as m:
        m.set_timing(True)
        m.update(x)
        _, _, variant = m.last_timing()
        p_gen = m.export()
    assert variant == 1
    x64 = x.astype(np.float64)
    exact = x64.T @ x64
    d = 256
    np.testing.assert_allclose(p_gen[1 + d:].reshape(d, d), exact, rtol=1e-12)
    np.testing.assert_allclose(p_mfma[1 + d:].reshape(d, d), exact, rtol=0, atol=1e-6 * np.abs(exact).max())
    np.testing.assert_allclose(p_mfma[1:1 + d], x64.sum(0), rtol=1e-7, atol=1e-6)   # 8-term fp32 sums, then fp64
    assert p_mfma[0] == 5000 and p_gen[0] == 5000


@pytest.mark.parametrize("n,d", [(8200, 512), (20011, 512), (12300, 768), (16500, 1024), (10250, 640), (20500, 1280)])
def test_moments_tile256_kernel_matches_float64(F, monkeypatch, n, d):
    """D >= 512, float16, at least 16 rows per column: the 256-column-slab kernel (moments_tile256.h) -- P/Q pairs (512, 1024),
    a leftover Z triangle (768, 1280), a ragged last superblock (640) -- against exact float64 raw moments, and against the
    128 x 128 kernel (FAD_MOMENTS_TILE256=0) on the same rows."""
    from fadtk_amd.hip import Moments
    x = structured_rows(n + d, n, d, np.float16)
    x64 = x.astype(np.float64)
    exact, sums = x64.T @ x64, x64.sum(0)
    got = {}
    for knob, want_variant in (("1", 2), ("0", 0)):
        monkeypatch.setenv("FAD_MOMENTS_TILE256", knob)
        with Moments(d) as m:
            m.set_timing(True)
            m.update(x[: n // 2]); m.update(x[n // 2:])            # two updates accumulate (halves below 16 d rows: the float64 kernel)
            variant = m.last_timing()[2]
            p = m.export()
        if n // 2 >= 16 * d:
            assert variant == want_variant
        got[knob] = p
    monkeypatch.setenv("FAD_MOMENTS_TILE256", "1")
    with Moments(d) as m:                                    # the whole matrix in one update: always eligible here
        m.set_timing(True); m.update(x)
        assert m.last_timing()[2] == 2
        p = m.export()
    M = p[1 + d:].reshape(d, d)
    assert p[0] == n
    np.testing.assert_allclose(M, exact, rtol=0, atol=1e-6 * np.abs(exact).max())
    np.testing.assert_array_equal(M, M.T)
    np.testing.assert_allclose(p[1:1 + d], sums, rtol=1e-7, atol=1e-5)
    for knob in got:
        np.testing.assert_allclose(got[knob][1 + d:].reshape(d, d), exact, rtol=0, atol=1e-6 * np.abs(exact).max(), err_msg=knob)
    mu, cov = F.calc_embd_statistics(x)
    mu_o, cov_o = O.embd_statistics(x)
    np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())


def test_moments_tile256_two_sets_one_launch_and_guard_second_pass(F):
    """update_multi on the 256-column-slab kernel: one benign set and one whose outlier columns trip the shift guard (second
    pass over x - c, un-shifted in the reduce), D = 768 so that a Z item carries its two column-sum rows through the un-shift."""
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(123)
    n, d = 13000, 768
    a = structured_rows(1, n, d, np.float16)
    b = rng.standard_normal((n + 700, d))
    b[:, 5:70] = 30.0 + 0.04 * b[:, 5:70]           # mean / std ~ 750 in superblock 0 ...
    b[:, 600:640] = -20.0 + 0.03 * b[:, 600:640]    # ... and in the Z superblock
    b[:, 300] = 2.5
    b = b.astype(np.float16)
    import torch
    with Moments(d) as ma, Moments(d) as mb:
        ma.set_timing(True)
        Moments.update_multi([ma, mb], [torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()])
        assert ma.last_timing()[2] == 2
        _, cov_a, na = ma.finalize()
        _, cov_b, nb = mb.finalize()
    assert na == n and nb == n + 700
    _, cov_ao = O.embd_statistics(a)
    _, cov_bo = O.embd_statistics(b)
    np.testing.assert_allclose(cov_a, cov_ao, rtol=0, atol=2e-6 * np.abs(cov_ao).max())
    assert np.abs(cov_b - cov_bo).max() <= 1e-6 * np.abs(cov_bo).max()
    blk = np.ix_(range(5, 70), range(5, 70))
    assert np.abs(cov_b[blk] - cov_bo[blk]).max() < 1e-6 * np.abs(cov_bo[blk]).max()      # accuracy relative to the VARIANCES
    blk = np.ix_(range(600, 640), range(600, 640))
    assert np.abs(cov_b[blk] - cov_bo[blk]).max() < 1e-6 * np.abs(cov_bo[blk]).max()
    assert abs(cov_b[300, 300]) < 1e-12


def test_moments_shift_guard_large_mean_small_std(F, monkeypatch):
    """|mean| >> std: the covariance is a tiny difference of huge raw moments.  The guard must notice; float16 rows then get a
    second pass over x - c (c = the column's mean on the float16 grid, exact by Sterbenz) and the raw moments are restored in
    float64 -- the covariance must come out as accurately as for well-centred data, i.e. to float32-sum accuracy RELATIVE TO
    THE VARIANCES, not to the raw moments."""
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(77)
    n, d = 6000, 256
    x = rng.standard_normal((n, d))
    x[:, :64] = 40.0 + 0.05 * x[:, :64]            # outlier dimensions: mean/std ~ 800 (fp16 grid is 0.03 there)
    x[:, 64] = 3.0                                 # a constant column
    x = x.astype(np.float16)
    _, cov_o = O.embd_statistics(x)
    mu, cov = F.calc_embd_statistics(x)
    assert np.abs(cov - cov_o).max() <= 1e-6 * np.abs(cov_o).max()
    assert abs(cov[64, 64]) < 1e-12 and np.abs(cov[:64, :64] - cov_o[:64, :64]).max() < 1e-6 * np.abs(cov_o[:64, :64]).max()
    np.testing.assert_allclose(mu.astype(np.float64), x.astype(np.float64).mean(0), rtol=1e-3)
    with Moments(d) as m:                          # guard decisions are per update: a benign block stays on the MFMA path
        m.set_timing(True)
        m.update(x[:, 128:].repeat(2, axis=1)[:, :d].copy())
        assert m.last_timing()[2] == 0
    monkeypatch.setenv("FAD_MOMENTS_SHIFT_GUARD", "0")     # without the guard the fp32 partial sums show (the knob is read when a
    with Moments(d) as m:                                  # handle is created; calc_embd_statistics keeps its handle per thread)
        m.update(x)
        _, cov_fast, _ = m.finalize()
    assert np.abs(cov_fast[:64, :64] - cov_o[:64, :64]).max() > 1e-7


@pytest.mark.parametrize("d,sizes,outliers", [(512, (100000, 70001), (0, 5, 300, 511)), (128, (50013,), (7, 100)),
                                              (768, (9000, 33, 8193), (1, 2, 3, 700)), (200, (4099,), (0, 199))])
def test_moments_guard_second_pass_raw_moments(F, d, sizes, outliers):
    """The second pass (x - c through the fp16 MFMA kernel, un-shifted in the reduce) must give the RAW moments of the rows:
    several sets in one launch of which only some are flagged, ragged last stages (rows not a multiple of 32), columns past a
    partial last tile, accumulation over two updates, a column that is constant and one whose mean is negative."""
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(d + len(sizes))
    blocks, refs = [], []
    for k, n in enumerate(sizes):
        x = rng.standard_normal((n, d)) * (0.5 + rng.random(d))
        if k != 1:                                       # the second set (if any) stays benign
            for c in outliers:
                x[:, c] = (-1) ** c * (20.0 + 3.0 * c / d) + 0.02 * x[:, c]
            x[:, outliers[0]] = 12.5                     # constant
        x = x.astype(np.float16)
        blocks.append(torch.from_numpy(x).cuda()); refs.append(x.astype(np.float64))
    accs = [Moments(d) for _ in sizes]
    try:
        Moments.update_multi(accs, blocks)
        Moments.update_multi(accs, [b[: b.shape[0] // 3] for b in blocks])       # on top: a second, shorter update
        for a, x in zip(accs, refs):
            p = a.export()
            xx = np.concatenate([x, x[: x.shape[0] // 3]])
            nn = xx.shape[0]
            assert p[0] == nn
            np.testing.assert_allclose(p[1:1 + d], xx.sum(0), rtol=1e-6, atol=1e-6 * nn)
            mu_, cov_ = p[1:1 + d] / nn, None
            cov = (p[1 + d:].reshape(d, d) - np.outer(p[1:1 + d], p[1:1 + d]) / nn) / (nn - 1)
            ref = np.cov(xx, rowvar=False)
            scale = np.sqrt(np.outer(np.diag(ref), np.diag(ref))) + 1e-30
            assert np.abs(cov - ref).max() <= 1e-6 * np.abs(ref).max()
            ok = np.diag(ref) > 0
            assert (np.abs(cov - ref)[np.ix_(ok, ok)] / scale[np.ix_(ok, ok)]).max() <= 2e-4     # entry by entry, in units of sigma_i sigma_j
    finally:
        for a in accs:
            a.close()


def test_moments_tile_kernel_structured_rows(F):
    """The fp16 tile kernel (four waves x 64 x 64, transpose reads) on rows with structure (ramps, sign patterns, a few
    large entries): raw moments against float64 arithmetic."""
    from fadtk_amd.hip import Moments
    x = structured_rows(31, 7001, 384, np.float16)
    with Moments(384) as m:
        m.update(x)
        p = m.export()
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(p[1 + 384:].reshape(384, 384), x64.T @ x64, rtol=0, atol=1e-6 * np.abs(x64.T @ x64).max())
    np.testing.assert_allclose(p[1:385], x64.sum(0), rtol=1e-7, atol=1e-6)


@pytest.mark.parametrize("d,sizes,dtype", [(512, (7001, 3000), "f16"), (128, (100, 0, 5000, 33, 2), "f16"),
                                           (384, (9000, 9000, 4097), "bf16"), (200, (700, 64), "f16"),
                                           (64, (1000, 10), "f32"), (768, (20000, 20000, 333, 5, 1, 8191, 8193, 64), "f16")])
def test_moments_update_multi_matches_separate_updates(F, d, sizes, dtype):
    """fad_moments_update_multi: up to 8 frame matrices -> 8 handles in one launch of each kernel.  Every handle must
    end up with the statistics of ITS matrix (and only those), on top of what it held before."""
    import torch
    from fadtk_amd.hip import Moments
    tdt = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}[dtype]
    rng = np.random.default_rng(d + len(sizes))
    blocks = [torch.from_numpy(rng.standard_normal((n, d)) * (0.5 + k) + 0.1 * k).to(tdt).cuda() for k, n in enumerate(sizes)]
    accs = [Moments(d) for _ in sizes]
    prior = torch.from_numpy(rng.standard_normal((50, d))).to(tdt).cuda()
    accs[0].update(prior)                                  # handle 0 is not fresh: the multi update must ADD
    Moments.update_multi(accs, blocks)
    half = dtype in ("f16", "bf16")
    for k, (m, b) in enumerate(zip(accs, blocks)):
        ref = b.double().cpu().numpy()
        if k == 0:
            ref = np.concatenate([prior.double().cpu().numpy(), ref])
        p = m.export()
        S = ref.T @ ref
        assert p[0] == ref.shape[0], k
        M = p[1 + d:].reshape(d, d)
        assert np.array_equal(M, M.T)
        assert np.abs(M - S).max() <= (1e-6 if half else 1e-11) * max(np.abs(S).max(), 1e-30), k
        np.testing.assert_allclose(p[1:1 + d], ref.sum(0), rtol=2e-7 if half else 1e-12, atol=1e-6 * max(np.abs(ref.sum(0)).max(), 1e-30))
        m.close()


def test_moments_update_multi_shift_guard_is_per_set(F):
    """One of the sets of a multi launch has outlier dimensions: it alone is redone in fp64, the others keep the
    fp16 MFMA result; both must match float64 numpy."""
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(5)
    d = 256
    good = rng.standard_normal((5000, d)).astype(np.float16)
    bad = rng.standard_normal((6000, d))
    bad[:, :64] = 40.0 + 0.05 * bad[:, :64]
    bad = bad.astype(np.float16)
    ma, mb = Moments(d), Moments(d)
    Moments.update_multi([ma, mb], [torch.from_numpy(good).cuda(), torch.from_numpy(bad).cuda()])
    for m, x in ((ma, good), (mb, bad)):
        mu, cov, n = m.finalize()
        _, cov_o = O.embd_statistics(x)
        tol = 2e-6             # (the flagged set takes the second pass: float32-sum accuracy relative to the variances, like the others)
        assert np.abs(cov - cov_o).max() <= tol * np.abs(cov_o).max() + 1e-12
        m.close()


def test_moments_update_multi_argument_errors(F):
    import torch
    from fadtk_amd.hip import Moments
    x = torch.randn((64, 128), device="cuda", dtype=torch.float16)
    with Moments(128) as a, Moments(64) as b:
        with pytest.raises(AssertionError):
            Moments.update_multi([a, b], [x, x])           # dimension mismatch
        with pytest.raises(RuntimeError):
            Moments.update_multi([a, a], [x, x])           # one handle twice
        with pytest.raises(AssertionError):
            Moments.update_multi([a], [x.float().cpu().numpy()])       # host rows


@pytest.mark.parametrize("d,sizes,dtype", [(128, (2250, 2250, 1, 0, 700, 256, 257, 5000), np.float16), (768, (2, 2, 2, 2), np.float16),
                                           (200, (31, 1000, 3), np.float32), (512, (300, 0, 0, 12), np.float64),
                                           (512, (9000, 20000, 300, 8192, 8193, 0, 33), np.float16),      # runs longer than the fp32 cap
                                           (128,) + ((tuple([2250] * 300),) + (np.float16,))])            # config-4 shape, 300 files
@pytest.mark.parametrize("where", ["host", "device"])
def test_moments_segmented_sums_long_and_short_segments(F, d, sizes, dtype, where):
    """Per-segment column sums (the per-file means of utils.py:16 / per-song means of fad.py:377 come from them):
    two-stage deterministic kernel, segments from 0 rows to many 256-row pieces."""
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(len(sizes) * d)
    x = (rng.standard_normal((sum(sizes), d)) + 0.25).astype(dtype)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    with Moments(d) as m:
        rows = torch.from_numpy(x).cuda() if where == "device" else x
        sums = m.update_segmented(rows, offs)
        p = m.export()
    x64 = x.astype(np.float64)
    want = np.stack([x64[a:b].sum(0) for a, b in zip(offs[:-1], offs[1:])])
    # aligned float16 rows in long segments take the fused route: the sums are those of the MFMA pass (8-term fp32 sums, then
    # fp64); everything else goes through the exact fp64 two-stage kernel
    np.testing.assert_allclose(sums, want, rtol=2e-7 if dtype == np.float16 else 1e-13, atol=2e-6 if dtype == np.float16 else 1e-12)
    np.testing.assert_allclose(p[1:1 + d], x64.sum(0), rtol=2e-7 if dtype == np.float16 else 1e-12, atol=1e-6)
    assert p[0] == sum(sizes)


@pytest.mark.parametrize("d,sizes,dtype", [(128, tuple([2250] * 40) + (700, 256, 8192, 1025, 5000, 2250), np.float16),      # config-4 shape: one run per file
                                           (256, (2250, 3000, 4100, 600, 8000, 257), np.float16),                            # three tiles: the diagonal ones walk
                                           (128, (2250, 900, 4097, 256, 3000), "bfloat16"),
                                           (128, (2250, 9000, 2250, 300), np.float16)])                                      # one file of two runs: the separate walk
@pytest.mark.parametrize("where", ["host", "device"])
def test_segmented_ref_running_sums_are_numpys_per_file_sums(F, d, sizes, dtype, where):
    """``fad_moments_update_segmented_ref``: the per-file float32 running column sums are np.mean's own (utils.py:16: the rows of a
    float16 file added one after the other in float32) BIT FOR BIT -- round 6: walked by the tile kernel's diagonal workgroups in the
    same pass when every file is one run (frames with an offset of 7 sigma, where the order of the adds shows in the last bits)."""
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(sum(sizes) + d)
    x32 = (rng.standard_normal((sum(sizes), d)) + 7.0).astype(np.float32)
    offs = np.concatenate([[0], np.cumsum(sizes)])
    if dtype == "bfloat16":
        xt = torch.from_numpy(x32).to(torch.bfloat16)
        xf = xt.to(torch.float32).numpy()
        rows = xt.cuda() if where == "device" else None
        if where == "host":
            pytest.skip("numpy has no bfloat16 host arrays")
    else:
        x = x32.astype(dtype)
        xf = x.astype(np.float32)
        rows = torch.from_numpy(x).cuda() if where == "device" else x
    want = np.zeros((len(sizes), d), dtype=np.float32)
    for f, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
        acc = np.zeros(d, dtype=np.float32)
        for r in range(a, b):
            acc = acc + xf[r]                       # float32, one row after the other
        want[f] = acc
    with Moments(d) as m:
        sums, runs = m.update_segmented(rows, offs, want_runsums=True)
        p = m.export()
    runs = runs.cpu().numpy() if hasattr(runs, "cpu") else np.asarray(runs)
    np.testing.assert_array_equal(runs, want)
    if dtype != "bfloat16":                         # ... which is what np.mean divides: the float16 file means, bit for bit
        for f, (a, b) in enumerate(zip(offs[:-1], offs[1:])):
            np.testing.assert_array_equal((runs[f].astype(np.float64) / (b - a)).astype(np.float32).astype(np.float16), x[a:b].mean(axis=0))
    x64 = xf.astype(np.float64)
    np.testing.assert_allclose(np.asarray(sums.cpu() if hasattr(sums, "cpu") else sums), np.stack([x64[a:b].sum(0) for a, b in zip(offs[:-1], offs[1:])]), rtol=2e-7, atol=1e-4)
    M = x64.T @ x64
    np.testing.assert_allclose(p[1 + d:].reshape(d, d), M, rtol=0, atol=2e-6 * np.abs(M).max())


@pytest.mark.parametrize("where", ["host", "device"])
def test_moments_file_mean_terms_match_numpy(F, where):
    """fad_moments_update_file_means: rows sqrt(n_f) m_f / sqrt(n_f) m~_f / n_f m~_f from per-file sums, with the
    float16 rounding of np.mean (utils.py:16) applied on the device."""
    import torch
    from fadtk_amd import _capi as K
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(12)
    d, sizes = 96, np.array([5, 1, 0, 40, 2250, 3], dtype=np.int64)
    sums = rng.standard_normal((len(sizes), d)) * sizes[:, None]
    sums[2] = 0.0
    ok = sizes > 0
    means = np.zeros_like(sums); means[ok] = sums[ok] / sizes[ok, None]
    mref = means.astype(np.float32).astype(np.float16).astype(np.float64)
    w = sizes.astype(np.float64)
    with Moments(d) as e, Moments(d) as r, Moments(d) as wt:
        if where == "device":
            Moments.update_file_means(e, r, wt, torch.from_numpy(sums).cuda(), torch.from_numpy(sizes).cuda(), K.FAD_F16)
        else:
            Moments.update_file_means(e, r, wt, sums, sizes, K.FAD_F16)
        pe, pr, pw = e.export(), r.export(), wt.export()
    np.testing.assert_allclose(pe[1 + d:].reshape(d, d), (means * w[:, None]).T @ means, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pr[1 + d:].reshape(d, d), (mref * w[:, None]).T @ mref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(pw[1:1 + d], (mref * w[:, None]).sum(0), rtol=1e-12, atol=1e-12)


def test_moments_streaming_merge_export_import(F):
    """Sufficient statistics are additive: chunked updates, merged handles and an export/import
    round trip (the multi-GPU reduce) all give the statistics of the whole set."""
    from fadtk_amd.hip import Moments
    x = structured_rows(11, 9000, 128, np.float16)
    with Moments(128) as whole, Moments(128) as parts, Moments(128) as a, Moments(128) as b, Moments(128) as c:
        whole.update(x)
        for lo, hi in ((0, 1), (1, 4000), (4000, 4000), (4000, 9000)):      # includes an empty block
            parts.update(x[lo:hi])
        a.update(x[:2500]); b.update(x[2500:])
        a.merge(b)
        c.import_(a.export())
        pw, pp, pa, pc = whole.export(), parts.export(), a.export(), c.export()
        assert whole.count == 9000 and parts.count == 9000 and c.count == 9000
    np.testing.assert_array_equal(pa, pc)
    scale = np.abs(pw).max()
    np.testing.assert_allclose(pp, pw, rtol=0, atol=1e-6 * scale)
    np.testing.assert_allclose(pa, pw, rtol=0, atol=1e-6 * scale)


def test_moments_more_than_2_31_elements(F):
    """16.8M x 128 fp16 = 2.15e9 elements (4.3 GB): 64-bit row offsets, 512 row-splits, HBM-bound shape."""
    import torch
    from fadtk_amd.hip import Moments
    n, d = (1 << 24) + 999, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.empty((n, d), dtype=torch.float16, device="cuda")
    for lo in range(0, n, 1 << 22):                       # fill in slabs to keep the fp32 temporary small
        hi = min(n, lo + (1 << 22))
        x[lo:hi] = (torch.randn((hi - lo, d), generator=g, device="cuda") * 0.5 + 0.25).to(torch.float16)
    with Moments(d) as m:
        m.update(x)
        packed = m.export()
    assert packed[0] == n
    x64_sum = torch.zeros(d, dtype=torch.float64, device="cuda")
    gram = torch.zeros((d, d), dtype=torch.float64, device="cuda")
    for lo in range(0, n, 1 << 21):
        blk = x[lo:lo + (1 << 21)].to(torch.float64)
        x64_sum += blk.sum(0); gram += blk.T @ blk
    np.testing.assert_allclose(packed[1:1 + d], x64_sum.cpu().numpy(), rtol=1e-9)
    np.testing.assert_allclose(packed[1 + d:].reshape(d, d), gram.cpu().numpy(), rtol=2e-7)
    last = x[-3:].to(torch.float64).cpu().numpy()        # the tail rows are in: drop them and the count/sum move
    with Moments(d) as m2:
        m2.update(x[:-3]); p2 = m2.export()
    np.testing.assert_allclose(packed[1:1 + d] - p2[1:1 + d], last.sum(0), rtol=0, atol=1e-4)


def test_moments_torch_device_tensors(F):
    import torch
    from fadtk_amd.hip import Moments
    x = structured_rows(13, 2048, 256, np.float32)
    for tdt in (torch.float16, torch.bfloat16, torch.float32, torch.float64):
        xt = torch.from_numpy(x).to("cuda").to(tdt)
        with Moments(256) as m:
            m.update(xt)
            torch.cuda.synchronize()
            mu, cov, n = m.finalize()
        ref = xt.to(torch.float64).cpu().numpy()
        mu_o, cov_o = ref.mean(0), np.cov(ref, rowvar=False)
        np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
        np.testing.assert_allclose(mu, mu_o, rtol=0, atol=1e-9)


def test_moments_torch_views_and_strides(F):
    """Device tensors that are not plain contiguous matrices: row-pitched views stay in place (ld > d), transposed or
    column-strided ones are made contiguous by the binding."""
    import torch
    from fadtk_amd.hip import Moments
    base = torch.from_numpy(structured_rows(17, 1500, 160, np.float16)).cuda()
    cases = {"pitched": base[:, :128], "rows_sliced": base[100:1300, :128], "col_strided": base[:, ::2][:, :64],
             "transposed": base[:160, :160].T}
    for name, v in cases.items():
        with Moments(v.shape[1]) as m:
            m.update(v)
            torch.cuda.synchronize()
            mu, cov, n = m.finalize()
        ref = v.to(torch.float64).cpu().numpy()
        assert n == ref.shape[0], name
        np.testing.assert_allclose(mu, ref.mean(0), rtol=1e-7, atol=1e-7, err_msg=name)
        np.testing.assert_allclose(cov, np.cov(ref, rowvar=False), rtol=0, atol=2e-6 * np.abs(np.cov(ref, rowvar=False)).max(),
                                   err_msg=name)


def test_moments_edge_cases(F):
    from fadtk_amd.hip import Moments
    with pytest.raises(AssertionError):
        F.calc_embd_statistics(np.zeros((1, 8), np.float16))
    with Moments(8) as m:
        m.update(np.zeros((0, 8), np.float16))
        assert m.count == 0
        with pytest.raises(AssertionError):
            m.finalize()
        with pytest.raises(AssertionError):
            m.update(np.zeros((4, 9), np.float16))
        m.update(np.ones((1, 8), np.float32))
        with pytest.raises(AssertionError):
            m.finalize()
        m.update(3 * np.ones((1, 8), np.float32))
        mu, cov, n = m.finalize()
        assert n == 2 and np.allclose(mu, 2.0) and np.allclose(cov, 2.0)
        m.reset()
        assert m.count == 0


# --------------------------------------------------------------------------------- frechet
def _pair_stats(a, b):
    m1, c1 = O.embd_statistics(a)
    m2, c2 = O.embd_statistics(b)
    return m1, c1, m2, c2


@pytest.mark.parametrize("case", ["c1_iid", "c1_iid_f32", "c1_iid_f64", "shifted", "shifted_long"])
def test_frechet_golden_pairs(F, golden, case):
    # (shifted_long: 60000 float16 rows at |mu| / sigma ~ 7 -- the reference's np.mean is a float32 running sum there, five of its
    #  float16 means are not the rounded exact ones, and with those the FAD would be 1.4e-3 off this fixture: fad_moments_set_reference_mean)
    g = golden["g2"][case]
    a, b = {"c1_iid": R.c1_pair, "c1_iid_f32": lambda: R.c1_pair(np.float32),
            "c1_iid_f64": lambda: R.c1_pair(np.float64), "shifted": R.shifted_pair,
            "shifted_long": lambda: R.shifted_pair(n=60000)}[case]()
    fad = F.calc_frechet_distance(*_pair_stats(a, b))
    assert isinstance(fad, np.float64)
    assert abs(fad - g["fad"]) / abs(g["fad"]) < 1e-9
    # and with the statistics from the GPU as well (the whole path)
    m1, c1 = F.calc_embd_statistics(a)
    m2, c2 = F.calc_embd_statistics(b)
    fad2 = F.calc_frechet_distance(m1, c1, m2, c2)
    assert abs(fad2 - g["fad"]) / abs(g["fad"]) < FAD_BAR / 10


@pytest.mark.parametrize("d", [128, 256])
def test_frechet_from_moments_takes_numpys_mean_from_handles_that_carry_it(F, golden, d):
    """The chains that start from packed moments (single pair and the batch of pairs): with fad_moments_set_reference_mean on both handles
    their mean term uses numpy's float32 running-sum mean (rounded to float16 by mean_dtype) and the distance meets the reference's value
    for the shifted pair at 60000 rows (golden g2.shifted_long: 1.4e-3 away with the rounded exact means -- which the same call returns
    with the switch off)."""
    import torch
    from fadtk_amd import hip
    # (d = 128: the float64 route, against the reference's own value; d = 256: the eight-launch chain and its batched form, against the oracle)
    a, b = R.shifted_pair(n=60000, d=d)
    want = golden["g2"]["shifted_long"]["fad"] if d == 128 else O.fad_between(a, b)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    with hip.Moments(d) as ma, hip.Moments(d) as mb, hip.Moments(d) as pa, hip.Moments(d) as pb:
        ma.set_reference_mean(True); mb.set_reference_mean(True)
        hip.Moments.update_multi([ma, mb], [ta, tb])
        hip.Moments.update_multi([pa, pb], [ta, tb])
        fad, _ = hip.frechet_from_moments(ma, mb, mean_dtype=hip.K.FAD_F16)
        assert abs(fad - want) / want < 1e-5, (fad, want)
        fad_plain, _ = hip.frechet_from_moments(pa, pb, mean_dtype=hip.K.FAD_F16)
        assert 1e-4 < abs(fad_plain - want) / want < 5e-3, (fad_plain, want)       # (d = 256: 2.2e-4; d = 128: 1.4e-3)
        both = hip.FrechetMultiJob([(ma, mb), (pa, pb), (ma, mb)], mean_dtype=hip.K.FAD_F16).result()
        assert abs(both[0][0] - want) / want < 1e-5 and abs(both[2][0] - want) / want < 1e-5, both
        assert abs(both[1][0] - fad_plain) <= 1e-9 * abs(fad_plain)
        # round 5, detached walk (set_reference_mean(.., detached=True): the frames are resident and stay as they are): fed in two pieces
        # while other work is queued on the stream -- the readers wait for the walk, the value is the attached one's bit for bit
        mu_att = ma.finalize()[0]
        ma.reset(); mb.reset()
        ma.set_reference_mean(True, detached=True); mb.set_reference_mean(True, detached=True)
        busy = torch.randn((4096, 4096), device="cuda")
        for _ in range(4):
            busy = busy @ busy * 1e-4                                   # something for the caller's stream to chew on meanwhile
        hip.Moments.update_multi([ma, mb], [ta[:30001], tb[:30001]])
        hip.Moments.update_multi([ma, mb], [ta[30001:], tb[30001:]])
        fad_det, _ = hip.frechet_from_moments(ma, mb, mean_dtype=hip.K.FAD_F16)
        assert np.array_equal(ma.finalize()[0], mu_att)                 # the running sums carried across the two updates: numpy's mean bit for bit
        assert abs(fad_det - fad) <= 2e-6 * abs(fad), (fad_det, fad)    # (two updates sum the partial tiles in another order: 1e-9 .. 1e-6 through the root)
        # detached handles fed HOST rows -- those go through the handle's staging area, which a walk that waits for nothing
        # would read before the copy has landed (and while the next block overwrites it): staged updates take the attached walk.  Twice
        # over, a reset in between (the second walk must not overtake the first chain's read of the running sums), readers on two streams.
        for rep in range(2):
            ma.reset(); mb.reset()
            ma.update(a[:30001]); ma.update(a[30001:])
            mb.update(b[:30001]); mb.update(b[30001:])
            fad_host, _ = hip.frechet_from_moments(ma, mb, mean_dtype=hip.K.FAD_F16)
            other = torch.cuda.Stream()
            with torch.cuda.stream(other):
                mu_other = ma.finalize()[0]
            assert np.array_equal(mu_other, mu_att) and np.array_equal(ma.finalize()[0], mu_att), rep
            assert abs(fad_host - fad) <= 2e-6 * abs(fad), (rep, fad_host, fad)
        # a merge gives the row order up -- the handle falls back to the exact mean instead of dividing dst's running sums by
        # the merged count
        with hip.Moments(d) as half1, hip.Moments(d) as half2:
            half1.set_reference_mean(True); half2.set_reference_mean(True)
            half1.update(ta[:30000]); half2.update(ta[30000:])
            half1.merge(half2)
            mu_merged = half1.finalize()[0]
        exact = a.astype(np.float64).mean(axis=0)
        np.testing.assert_allclose(mu_merged, exact, rtol=1e-12)
        del busy


def test_frechet_identical_sets_is_zero(F, golden):
    a, _ = R.c1_pair()
    fad = F.calc_frechet_distance(*_pair_stats(a, a))
    assert abs(fad) < 1e-8 * golden["g2"]["identical"]["tr1"]


@pytest.mark.parametrize("d", [64, 512])
def test_frechet_decaying_spectrum(F, golden, d):
    x1 = R.decaying_rows(30, 4 * d, d, basis_seed=40)
    x2 = R.decaying_rows(31, 4 * d, d, basis_seed=40, gain=1.1)
    x3 = R.decaying_rows(32, 4 * d, d, basis_seed=41)
    for other, key in ((x2, f"decay_same_basis_d{d}"), (x3, f"decay_diff_basis_d{d}")):
        fad = F.calc_frechet_distance(*_pair_stats(x1, other))
        assert abs(fad - golden["g2"][key]["fad"]) / abs(golden["g2"][key]["fad"]) < 1e-6, key


@pytest.mark.parametrize("d,rows", [(128, 2), (128, 10), (128, 50), (768, 2), (768, 10)])
def test_frechet_rank_deficient(F, golden, d, rows):
    g = golden["g2"][f"rankdef_d{d}_n{rows}"]
    mu_b, cov_b = R.baseline_stats(50 + d, 4 * d, d)
    s = R.songs(60 + rows, 1, rows, d)[0]
    mu_s, cov_s = O.embd_statistics(s)
    fad = F.calc_frechet_distance(mu_b, cov_b, mu_s, cov_s)
    assert abs(fad - g["fad"]) / abs(g["fad"]) < 1e-6


@pytest.mark.parametrize("key", ["short_eval_d256_n150_p2", "short_eval_d256_n150_p3", "short_eval_d256_n150_p4",
                                 "fullrank_d128_p3", "fullrank_d128_p4", "f32cov_d128_p3", "f32cov_d128_p4"])
def test_frechet_hard_spectra_g10(F, golden, key):
    """Near-singular products: eigenvalues of C1 C2 at roundoff level (some slightly negative) used to run away after
    ~50 iterations and end in NaN / the eps fallback; the divergence guard returns the converged trace instead."""
    from test_oracle_golden import _g10_case
    g = golden["g10"][key]
    m1, c1, m2, c2 = _g10_case(key)
    fad = F.calc_frechet_distance(m1, c1, m2, c2)
    assert np.isfinite(fad)
    assert abs(fad - g["fad"]) / abs(g["fad"]) < 2e-6, (fad, g["fad"])


def test_per_song_short_eval_power_law_is_scored(F, golden):
    """--indiv on songs of 65..D frames with power-law spectra (batched D x D Newton-Schulz route, no eps fallback
    there): every song must get a finite score close to the reference's, none may be dropped."""
    from fadtk_amd import hip
    songs, want = [], []
    for p in (2.0, 3.0, 4.0):
        base = R.decaying_rows(50, 4096, 256, 52, power=p)
        songs.append(R.decaying_rows(51, 150, 256, 53, power=p))
        want.append(golden["g10"][f"short_eval_d256_n150_p{p:g}"]["fad"])
    for k, (song, ref) in enumerate(zip(songs, want)):
        p = (2.0, 3.0, 4.0)[k]
        mu_b, cov_b = O.embd_statistics(R.decaying_rows(50, 4096, 256, 52, power=p))
        scores, status = hip.frechet_batched(mu_b.astype(np.float64), cov_b, song, [0, song.shape[0]])
        assert status[0] == 0 and np.isfinite(scores[0]), (p, status, scores)
        assert abs(scores[0] - ref) / abs(ref) < 2e-6, (p, scores[0], ref)


@pytest.mark.parametrize("d,n,scale,expect_mixed", [(64, 2000, 1.0, True), (128, 1024, 1.0, True), (256, 3000, 30.0, True),
                                                    (512, 4000, 1e-3, True), (768, 6000, 1.0, None)])
def test_frechet_mixed_precision_matches_float64_iteration(F, monkeypatch, d, n, scale, expect_mixed):
    """Well-conditioned products take the float32 Newton-Schulz + float64 correction route (diag: converged == 3);
    the result must agree with the all-float64 iteration (FAD_FRECHET_MIXED=0, new thread = new workspace) and the
    oracle far below the 1e-4 bar."""
    import threading
    from fadtk_amd import hip
    rng = np.random.default_rng(d + n)
    a = (rng.standard_normal((n, d)) * scale * (0.5 + rng.random(d))).astype(np.float32)
    b = (1.05 * rng.standard_normal((n, d)) * scale * (0.5 + rng.random(d)) + 0.02 * scale).astype(np.float32)
    m1, c1, m2, c2 = _pair_stats(a, b)
    fad_mixed, diag = hip.frechet(m1.astype(np.float64), c1, m2.astype(np.float64), c2)
    if expect_mixed:
        assert diag["converged"] == 3, diag
    else:                                    # the spread of this spectrum sits at the edge: either route is legitimate
        assert diag["converged"] in (1, 2, 3), diag
    fad_again, diag_again = hip.frechet(m1.astype(np.float64), c1, m2.astype(np.float64), c2)     # batch sized by the first call
    assert fad_again == fad_mixed and diag_again["iters"] == diag["iters"]
    out = {}

    def run64():
        out["v"] = hip.frechet(m1.astype(np.float64), c1, m2.astype(np.float64), c2)
    monkeypatch.setenv("FAD_FRECHET_MIXED", "0")
    t = threading.Thread(target=run64); t.start(); t.join()
    fad64, diag64 = out["v"]
    assert diag64["converged"] in (1, 2)
    assert abs(diag["tr_sqrt"] - diag64["tr_sqrt"]) <= 2e-10 * abs(diag64["tr_sqrt"])
    ref = O.frechet_distance(m1.astype(np.float64), c1, m2.astype(np.float64), c2, run_sqrtm=False)
    assert abs(fad_mixed - ref) <= 1e-7 * abs(ref)


def test_frechet_rejected_prediction_iterates_on(F, monkeypatch):
    """The float32 leg takes Y_{k+1} as final when the predicted residual is small enough for the float64 correction to
    absorb.  With an absurdly generous threshold (FAD_FRECHET_PRED_THR, read once per thread) the prediction fires too
    early, the error estimate rejects the iterate, and the iteration must go on from it -- same answer, still on the
    float32 route, and the thread predicts from the float32 floor from then on."""
    import threading
    from fadtk_amd import hip
    rng = np.random.default_rng(77)
    d, n = 256, 4000
    a = (rng.standard_normal((n, d)) * (0.5 + rng.random(d))).astype(np.float32)
    b = (1.05 * rng.standard_normal((n, d)) * (0.5 + rng.random(d)) + 0.02).astype(np.float32)
    m1, c1, m2, c2 = _pair_stats(a, b)
    args = (m1.astype(np.float64), c1, m2.astype(np.float64), c2)
    fad0, diag0 = hip.frechet(*args)
    assert diag0["converged"] == 3
    out = {}

    def run():
        out["first"] = hip.frechet(*args)
        out["second"] = hip.frechet(*args)
    monkeypatch.setenv("FAD_FRECHET_PRED_THR", "0.9")
    t = threading.Thread(target=run); t.start(); t.join()
    for key in ("first", "second"):
        fad, diag = out[key]
        assert diag["converged"] == 3, diag
        assert abs(diag["tr_sqrt"] - diag0["tr_sqrt"]) <= 1e-9 * abs(diag0["tr_sqrt"])
        assert abs(fad - fad0) <= 1e-7 * abs(fad0)
    assert out["first"][1]["iters"] >= diag0["iters"] - 1
    ref = O.frechet_distance(*args, run_sqrtm=False)
    assert abs(out["first"][0] - ref) <= 1e-7 * abs(ref)


def test_frechet_mixed_precision_falls_back_when_ill_conditioned(F):
    """Decaying spectrum (cond ~ 1e7): the float32 leg either does not converge or its error estimate is too large;
    the float64 iteration must take over transparently."""
    from fadtk_amd import hip
    x1 = R.decaying_rows(30, 2048, 512, basis_seed=40)
    x2 = R.decaying_rows(31, 2048, 512, basis_seed=40, gain=1.1)
    m1, c1, m2, c2 = _pair_stats(x1, x2)
    fad, diag = hip.frechet(m1.astype(np.float64), c1, m2.astype(np.float64), c2)
    assert diag["converged"] in (1, 2)           # (x_min estimate ~1e-5: the wide chain of round 5 declines it before it starts)
    ref = O.frechet_distance(m1.astype(np.float64), c1, m2.astype(np.float64), c2, run_sqrtm=False)
    assert abs(fad - ref) <= 1e-6 * abs(ref)


@pytest.mark.gpu
@pytest.mark.parametrize("power,on_chain", [(0.5, True), (1.0, True), (2.0, False)])
def test_frechet_wide_chain_takes_decaying_spectra(F, monkeypatch, power, on_chain):
    """Round 5: pairs whose covariances decay like k^-power (both sets share the eigenvectors: bench.py's extra_decaying recipe at
    D = 512, 20 000 frames per set) stay on the eight-launch chain -- scaled Newton-Schulz steps on split-float16 operands, the exact
    correction, and the verification products (csrc/ns_fast.h: SP_V2 / SP_V3) where the norm bound says nothing -- up to k^-1 (condition
    3e5 of the product); k^-2 is declined on the device and takes the float64 route as before.  Single call, second call (launch counts
    follow the thread's history, the value must not), the batch of pairs, FAD_FRECHET_WIDE=0 (round 4's routing), all against the oracle."""
    import threading
    import torch
    from fadtk_amd import hip
    d, n = 512, 20000
    rng = np.random.default_rng(int(power * 10) + 3)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    lam = np.arange(1, d + 1) ** (-power / 2.0)
    a = ((rng.standard_normal((n, d)) * lam) @ q.T).astype(np.float16)
    b = ((1.05 * rng.standard_normal((n, d)) * lam) @ q.T + 0.01).astype(np.float16)
    ref = O.fad_between(a, b)
    out = {}

    def run(tag):
        with hip.Moments(d) as ma, hip.Moments(d) as mb:
            hip.Moments.update_multi([ma, mb], [torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()])
            first = hip.frechet_from_moments(ma, mb, mean_dtype=0)
            second = hip.frechet_from_moments(ma, mb, mean_dtype=0)
            multi = hip.FrechetMultiJob([(ma, mb)] * 4, mean_dtype=0).result()
            multi2 = hip.FrechetMultiJob([(ma, mb)] * 4, mean_dtype=0).result()
        out[tag] = (first, second, multi, multi2)
    t = threading.Thread(target=run, args=("wide",)); t.start(); t.join()          # (knobs and hints live per thread)
    monkeypatch.setenv("FAD_FRECHET_WIDE", "0")
    t = threading.Thread(target=run, args=("narrow",)); t.start(); t.join()
    first, second, multi, multi2 = out["wide"]
    assert first[1]["route"] == (2 if on_chain else 0), first
    assert second[0] == first[0], (first, second)                                   # the value is a function of the inputs alone
    for f, dg in (first, *multi, *multi2):
        assert abs(f - ref) <= 2e-6 * abs(ref), (power, f, ref, dg)
    #  which acceptance let the score through is in the record: decaying spectra on the chain pass on the MEASURED verification
    # products, not on the norm bound; the float64 route sets nothing
    assert first[1]["verified"] == (1 if on_chain else 0) and all(dg["verified"] == (1 if on_chain else 0) for _, dg in multi2), (first, multi2)
    for f, dg in multi2:
        assert dg["route"] == (2 if on_chain else 0), dg
        assert abs(f - multi2[0][0]) == 0.0
    narrow = out["narrow"][0]
    assert narrow[1]["route"] == (0 if power > 0.25 else 2)
    assert abs(narrow[0] - ref) <= 2e-6 * abs(ref)
    assert abs(narrow[0] - first[0]) <= 2e-6 * abs(ref)


def test_frechet_from_moments_mean_dtype_reproduces_float16_mean_term(F, golden):
    """The device route with mean_dtype = FAD_F16 gives the reference's value for float16 embeddings: fp16 means,
    fp16 difference, float32-accumulated fp16 dot -- on the pair where that matters most."""
    from fadtk_amd import _capi as K, hip
    a, b = R.shifted_pair()
    g = golden["g2"]["shifted"]
    with hip.Moments(128) as ma, hip.Moments(128) as mb:
        ma.update(a); mb.update(b)
        fad_ref_like, diag = hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16)
        fad_f64, diag64 = hip.frechet_from_moments(ma, mb)
    # float64 column means rounded the way np.mean rounds them (numpy's own float32 running sum can differ by an ulp
    # of float16 on a few entries), then numpy's float16 arithmetic: must be bit for bit what the device returns
    m1 = a.astype(np.float64).mean(0).astype(np.float32).astype(np.float16)
    m2 = b.astype(np.float64).mean(0).astype(np.float32).astype(np.float16)
    gap = m1 - m2
    assert gap.dtype == np.float16 and diag["mean_term"] == float(gap.dot(gap))
    assert abs(fad_ref_like - g["fad"]) / abs(g["fad"]) < 1e-5
    assert abs(fad_f64 - g["fad"]) / abs(g["fad"]) > 1e-5         # float64 means are a different (better) estimate


def test_frechet_jobs_in_flight_match_the_blocking_call(F):
    """fad_frechet_from_moments_begin / fad_frechet_end: several scores enqueued on different streams before any is collected
    give bit for bit what the blocking call gives -- well-conditioned pairs (float32 leg accepted), an ill-conditioned one (the
    float64 iteration runs inside end()), a dimension off the float32 route, and the errors of the blocking call."""
    import torch
    from fadtk_amd import hip, _capi as K
    rng = np.random.default_rng(3)
    cases = []
    for d, n, decay in ((128, 1500, 0.0), (256, 3000, 0.0), (512, 2048, 1.5), (96, 700, 0.0), (128, 900, 0.0)):
        lam = np.arange(1, d + 1) ** (-decay / 2.0)
        a = (rng.standard_normal((n, d)) * lam).astype(np.float16)
        b = (1.05 * rng.standard_normal((n, d)) * lam + 0.02).astype(np.float16)
        cases.append((torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), d))
    want = []
    for a, b, d in cases:
        with hip.Moments(d) as ma, hip.Moments(d) as mb:
            hip.Moments.update_multi([ma, mb], [a, b])
            want.append(hip.frechet_from_moments(ma, mb, mean_dtype=K.FAD_F16))
    streams = [torch.cuda.Stream() for _ in cases]
    handles, jobs = [], []
    for (a, b, d), st in zip(cases, streams):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            ma, mb = hip.Moments(d), hip.Moments(d)
            hip.Moments.update_multi([ma, mb], [a, b])
            jobs.append(hip.FrechetJob(ma, mb, mean_dtype=K.FAD_F16))
            handles.append((ma, mb))
    got = [j.result() for j in reversed(jobs)][::-1]                      # collected in another order than begun
    for (f0, d0), (f1, d1) in zip(want, got):
        assert f1 == f0 and d1["converged"] == d0["converged"] and d1["tr_sqrt"] == d0["tr_sqrt"]
    assert [d["converged"] for _, d in got][:2] == [3, 3] and got[2][1]["converged"] in (1, 2) and got[3][1]["converged"] in (1, 2)
    with pytest.raises(RuntimeError):
        jobs[0].result()                                                  # collected already
    ma, mb = handles[0]
    mb.reset(); mb.update(cases[0][1][:1])
    with pytest.raises(AssertionError):                                   # fewer than two frames: the error arrives at end()
        hip.FrechetJob(ma, mb).result()
    ma.reset(); ma.update(cases[0][0])
    mb.reset(); mb.update(cases[0][1])
    pending = [hip.FrechetJob(ma, mb) for _ in range(8)]                  # eight slots per thread and device
    with pytest.raises(RuntimeError):
        hip.FrechetJob(ma, mb)
    vals = [j.result()[0] for j in pending]
    assert len(set(vals)) == 1
    # dropped or cancelled jobs give their slots back; the handles may be fed again on the same stream right after begin()
    dropped = [hip.FrechetJob(ma, mb) for _ in range(8)]
    dropped[0].cancel(); dropped[0].cancel()
    del dropped
    j1 = hip.FrechetJob(ma, mb)
    ma.reset(); ma.update(cases[1][0][:, :128].contiguous()); mb.reset()          # same stream: ordered behind the job's copy
    assert j1.result()[0] == vals[0]
    for ma, mb in handles:
        ma.close(); mb.close()


def test_frechet_properties(F):
    """Size-independent properties: symmetry in its arguments and quadratic scaling."""
    a = structured_rows(21, 3000, 96, np.float32)
    b = structured_rows(22, 2500, 96, np.float32) * 1.1 + 0.05
    m1, c1, m2, c2 = _pair_stats(a, b)
    f12 = F.calc_frechet_distance(m1, c1, m2, c2)
    f21 = F.calc_frechet_distance(m2, c2, m1, c1)
    assert abs(f12 - f21) < 1e-9 * abs(f12)
    f_scaled = F.calc_frechet_distance(3 * m1, 9 * c1, 3 * m2, 9 * c2)
    assert abs(f_scaled - 9 * f12) < 1e-7 * abs(9 * f12)
    assert abs(f12 - O.frechet_distance(m1, c1, m2, c2, run_sqrtm=False)) < 1e-9 * abs(f12)


def test_frechet_errors(F):
    e8, e9 = np.eye(8), np.eye(9)
    with pytest.raises(AssertionError):
        F.calc_frechet_distance(np.zeros(8), e8, np.zeros(9), e9)
    with pytest.raises(AssertionError):
        F.calc_frechet_distance(np.zeros(8), e8, np.zeros(8), e9)
    bad = e8.copy(); bad[0, 0] = np.nan
    with pytest.raises(ValueError):
        F.calc_frechet_distance(np.zeros(8), bad, np.zeros(8), e8)
    neg = -e8                                              # product with negative eigenvalues: no real root
    with pytest.raises(ValueError):
        F.calc_frechet_distance(np.zeros(8), neg, np.zeros(8), e8)
    # zero covariance: root is zero, distance is the mean term + traces
    assert F.calc_frechet_distance(np.ones(8), np.zeros((8, 8)), np.zeros(8), e8) == pytest.approx(16.0)


def test_thread_pool_callers_are_safe(F):
    """fadtk drives these functions from thread pools (tmap at fad.py:229, 387): handle-less entry points keep
    per-thread workspaces, distinct handles are independent -- concurrent calls must give the serial answers."""
    from concurrent.futures import ThreadPoolExecutor
    rng = np.random.default_rng(5)
    jobs = []
    for k in range(12):
        d = (32, 64, 96)[k % 3]
        a = (rng.standard_normal((400 + 10 * k, d)) * (1 + 0.1 * k)).astype(np.float16)
        b = (rng.standard_normal((300 + 7 * k, d)) + 0.05 * k).astype(np.float16)
        jobs.append((a, b))

    def one(job):
        a, b = job
        m1, c1 = F.calc_embd_statistics(a)
        m2, c2 = F.calc_embd_statistics(b)
        return float(F.calc_frechet_distance(m1, c1, m2, c2))

    serial = [one(j) for j in jobs]
    with ThreadPoolExecutor(max_workers=6) as ex:
        threaded = list(ex.map(one, jobs * 3))
    np.testing.assert_allclose(threaded, serial * 3, rtol=1e-12)
    want = [O.fad_between(a, b) for a, b in jobs[:3]]
    np.testing.assert_allclose(serial[:3], want, rtol=1e-6)


def test_frechet_from_moments_matches_host_route(F):
    from fadtk_amd import hip
    a, b = R.c1_pair()
    with hip.Moments(128) as ma, hip.Moments(128) as mb:
        ma.update(a); mb.update(b)
        fad, diag = hip.frechet_from_moments(ma, mb)
        mu1, c1, _ = ma.finalize(); mu2, c2, _ = mb.finalize()
    fad_host, _ = hip.frechet(mu1, c1, mu2, c2)
    assert abs(fad - fad_host) < 1e-12 * abs(fad_host)
    assert diag["converged"] in (1, 3) and diag["iters"] < 20
    with hip.Moments(128) as ma, hip.Moments(128) as mb:
        ma.update(a); mb.update(b[:1])
        with pytest.raises(AssertionError):
            hip.frechet_from_moments(ma, mb)


def test_config3_full_size(F, golden):
    """BASELINE config 3: N=100000, D=512 fp16, whole path on the GPU vs the reference's scalar."""
    g = golden["g7"]
    a, b = R.c3_pair()
    assert R.checksum(a) == pytest.approx(g["in_checksum"][0], rel=1e-12)
    m1, c1 = F.calc_embd_statistics(a)
    m2, c2 = F.calc_embd_statistics(b)
    assert m1.dtype == np.float16
    assert np.trace(c1) == pytest.approx(g["tr1"], rel=1e-7)
    assert np.trace(c2) == pytest.approx(g["tr2"], rel=1e-7)
    d = m1 - m2
    assert float(d.dot(d)) == pytest.approx(g["mean_term"], rel=2e-3)      # fp16 scalar in the reference too
    fad = F.calc_frechet_distance(m1, c1, m2, c2)
    assert abs(fad - g["fad"]) / g["fad"] < FAD_BAR
    # device-resident route (what bench.py times)
    import torch
    from fadtk_amd import hip
    with hip.Moments(512) as ma, hip.Moments(512) as mb:
        ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
        fad_dev, diag = hip.frechet_from_moments(ma, mb)
        # this route keeps the means in float64 (no fp16 rounding of mu): compare the root, not the mean term
        assert diag["tr_sqrt"] == pytest.approx(g["tr_sqrt"], rel=2e-7)
        assert abs((fad_dev - diag["mean_term"]) - (g["fad"] - g["mean_term"])) / g["fad"] < FAD_BAR
        # ... and with the reference's float16 mean term (what bench.py's loop asks for: mean_dtype = float16 -- the means rounded to
        # float16, their difference and its dot product in float16 like numpy's, fad.py:48 / :83-84): the whole scalar, single pair and batch
        fad_h, diag_h = hip.frechet_from_moments(ma, mb, mean_dtype=0)
        assert diag_h["mean_term"] == pytest.approx(g["mean_term"], rel=2e-3)
        assert abs(fad_h - g["fad"]) / g["fad"] < FAD_BAR
        res = hip.FrechetMultiJob([(ma, mb)] * 3, mean_dtype=0).result()
        for fad_k, _ in res:
            assert abs(fad_k - g["fad"]) / g["fad"] < FAD_BAR


# --------------------------------------------------------------------------------- online statistics
def test_online_statistics_golden_g3(F, golden, golden_dir, tmp_path):
    g = golden["g3"]
    z = np.load(golden_dir / "g3_online.npz")
    blocks = R.ragged_files(g["seed"], g["n_files"], g["d"])
    files = []
    for i, blk in enumerate(blocks):
        np.save(tmp_path / f"f{i:03d}.npy", blk)
        files.append(tmp_path / f"f{i:03d}.npy")
    mu, cov = F.calculate_embd_statistics_online(files)
    assert mu.dtype == np.float64 and cov.dtype == np.float64
    np.testing.assert_allclose(mu, z["mu"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(cov, z["cov"], rtol=0, atol=2e-6 * np.abs(z["cov"]).max())
    mu_i, cov_i = F.dataset_statistics(blocks, compat=False)            # plain raw-moment estimate
    allrows = np.concatenate(blocks).astype(np.float64)
    np.testing.assert_allclose(cov_i, np.cov(allrows, rowvar=False), rtol=0, atol=2e-6 * np.abs(cov_i).max())
    # Q5: a one-row file poisons the covariance, not the mean
    np.save(tmp_path / "one.npy", blocks[0][:1])
    mu_n, cov_n = F.calculate_embd_statistics_online(files[:3] + [tmp_path / "one.npy"])
    assert np.isnan(cov_n).all() and g["cov_nan_all"]
    np.testing.assert_allclose(mu_n, z["mu_nan"], rtol=1e-12)


def test_online_statistics_at_config2_size(F, tmp_path):
    """BASELINE config 2 at its stated size: 1 000 files of ten 128-dimensional float16 frames (VGGish on 1k ten-second wavs) for the
    baseline and as many for the evaluation set, through ``calculate_embd_statistics_online`` (utils.py:19-46: per-file float16 means,
    sequential merge) and ``calc_frechet_distance``, against the oracle on the same files.  Ten frames per file: every file's scatter has
    rank 9, the float16 rounding of its mean is worth 1e-4 of the FAD -- the quirk has to be reproduced, not averaged away."""
    sets = {}
    for name, seed, gain, shift in (("base", 2000, 1.0, 0.0), ("eval", 3000, 1.05, 0.02)):
        rng = np.random.default_rng(seed)
        mix = rng.standard_normal((128, 128)) / np.sqrt(128.0)
        files, blocks = [], []
        for i in range(1000):
            blk = (gain * (rng.standard_normal((10, 128)) @ mix + 0.3 * rng.standard_normal((1, 128))) + shift + 0.25).astype(np.float16)
            np.save(tmp_path / f"{name}{i:04d}.npy", blk)
            files.append(tmp_path / f"{name}{i:04d}.npy"); blocks.append(blk)
        mu, cov = F.calculate_embd_statistics_online(files)
        mu_o, cov_o = O.statistics_online(blocks)
        np.testing.assert_allclose(mu, mu_o, rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
        sets[name] = (mu, cov, mu_o, cov_o)
    got = F.calc_frechet_distance(sets["base"][0], sets["base"][1], sets["eval"][0], sets["eval"][1])
    want = O.frechet_distance(sets["base"][2], sets["base"][3], sets["eval"][2], sets["eval"][3], run_sqrtm=False)
    assert abs(got - want) / abs(want) < 1e-6


def test_online_statistics_shifted_files_take_numpys_per_file_means(F, golden, golden_dir):
    """Round 5 fixture g3_shifted (the REFERENCE's calculate_embd_statistics_online on 12 float16 files of 9000-40000 frames with
    |mu| / sigma ~ 7): the per-file float16 means are np.mean's -- a float32 running sum per file (utils.py:16) -- which the online path
    reproduces with fad_moments_update_segmented_ref / fad_moments_update_file_means_ref, from host blocks and from a device tensor.
    With ref_means=False (rounds 1-4: rounded exact means) the dataset mean is measurably further from the reference's."""
    import torch
    from fadtk_amd.utils import OnlineStats
    g = golden["g3_shifted"]
    z = np.load(golden_dir / "g3_online.npz")
    blocks = R.shifted_files(g["seed"], g["n_files"], g["d"], min_rows=g["min_rows"], max_rows=g["max_rows"])
    mu, cov = F.dataset_statistics(blocks)
    np.testing.assert_allclose(mu, z["mu_shifted"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(cov, z["cov_shifted"], rtol=0, atol=1e-5 * np.abs(np.diag(z["cov_shifted"])).max())
    rows = torch.from_numpy(np.concatenate(blocks)).cuda()
    sizes = [b.shape[0] for b in blocks]
    out = {}
    for ref in (True, False):
        st = OnlineStats(g["d"], 0, compat=True, ref_means=ref)
        st.add_group(rows, sizes)
        out[ref] = st.finish()
        st.close()
    np.testing.assert_allclose(out[True][0], z["mu_shifted"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(out[True][1], z["cov_shifted"], rtol=0, atol=1e-5 * np.abs(np.diag(z["cov_shifted"])).max())
    assert np.abs(out[False][0] - z["mu_shifted"]).max() > 10 * np.abs(out[True][0] - z["mu_shifted"]).max() + 1e-9


def test_score_individual_shifted_songs_take_numpys_means(F, golden, tmp_path):
    """Round 5 fixture g4_shifted (the REFERENCE's score_individual on six float16 songs of 12000-30000 frames with |mu| / sigma ~ 7 against a
    baseline with the same offset): the per-song mean term uses np.mean's float16 mean (fad.py:377 -> :48), a float32 running sum --
    csrc/frechet_songs.hip: mean_like_reference walks it for 16-bit frames since round 5."""
    from pathlib import Path
    from fadtk_amd import hip
    g = golden["g4_shifted"]
    d = g["d"]
    srows = R.shifted_files(g["songs_seed"], len(g["names"]), d, min_rows=g["min_rows"], max_rows=g["max_rows"])
    rngb = np.random.default_rng(g["base_seed"])
    xb = rngb.standard_normal((g["base_n"], d)) * (1.0 + 0.3 * rngb.random(d)) + 7.0
    mu_b, cov_b = xb.mean(axis=0), np.cov(xb, rowvar=False)
    want = {ln.rsplit(",", 1)[0].rsplit("/", 1)[1]: float(ln.rsplit(",", 1)[1]) for ln in g["csv"].split("\n")}
    offs = np.concatenate([[0], np.cumsum([r.shape[0] for r in srows])])
    scores, status = hip.frechet_batched(mu_b, cov_b, np.concatenate(srows), offs, mean_mode=1)
    assert (status == 0).all()
    for nm, sc in zip(g["names"], scores):
        assert abs(sc - want[nm]) <= 1e-5 * abs(want[nm]), (nm, sc, want[nm])
    exact, _ = hip.frechet_batched(mu_b, cov_b, np.concatenate(srows), offs, mean_mode=0)      # float64 means: a different (larger) mean term
    assert max(abs(a - want[nm]) / abs(want[nm]) for nm, a in zip(g["names"], exact)) > 1e-5
    # ... and through the product's own entry point, CSV order included
    model = g["model"]
    evs = tmp_path / "evalshift"
    (evs / "embeddings" / model).mkdir(parents=True)
    for nm, rows in zip(g["names"], srows):
        (evs / nm).write_bytes(b"")
        np.save(evs / "embeddings" / model / (Path(nm).stem + ".npy"), rows)
    np.savez(tmp_path / "base_shift.npz", **{f"{model}.mu": mu_b, f"{model}.cov": cov_b})
    fad = F.FrechetAudioDistance(_Toy(model), audio_load_worker=2, load_model=False)
    csv = fad.score_individual(str(tmp_path / "base_shift.npz"), evs, tmp_path / "indiv_shift.csv")
    got = [ln.rsplit(",", 1) for ln in csv.read_text().replace(str(tmp_path), "{ROOT}").split("\n")]
    wantl = [ln.rsplit(",", 1) for ln in g["csv"].split("\n")]
    assert [a for a, _ in got] == [a for a, _ in wantl]
    np.testing.assert_allclose([float(b) for _, b in got], [float(b) for _, b in wantl], rtol=1e-5)


# --------------------------------------------------------------------------------- per-song
class _Toy:
    def __init__(self, name):
        self.name = name
        self.sr = 16000

    def load_model(self):
        raise AssertionError("not needed")


def test_score_individual_golden_g4_and_load_stats_g5(F, golden, golden_dir, tmp_path):
    g = golden["g4"]
    model, d = g["model"], g["d"]
    evald = tmp_path / "evalset"
    (evald / "embeddings" / model).mkdir(parents=True)
    from pathlib import Path
    for nm, rows in zip(g["names"], R.songs(g["songs_seed"], len(g["names"]), g["rows"], d)):
        (evald / nm).write_bytes(b"")
        np.save(evald / "embeddings" / model / (Path(nm).stem + ".npy"), rows)
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], d)
    np.savez(tmp_path / "base.npz", **{f"{model}.mu": mu_b, f"{model}.cov": cov_b})
    fad = F.FrechetAudioDistance(_Toy(model), audio_load_worker=2, load_model=False)
    out = fad.score_individual(str(tmp_path / "base.npz"), evald, tmp_path / "indiv.csv")
    got = [ln.rsplit(",", 1) for ln in out.read_text().replace(str(tmp_path), "{ROOT}").split("\n")]
    want = [ln.rsplit(",", 1) for ln in g["csv"].split("\n")]
    assert [a for a, _ in got] == [a for a, _ in want]                    # order, dropped 1-frame song, ','->'_'
    np.testing.assert_allclose([float(b) for _, b in got], [float(b) for _, b in want], rtol=1e-6)
    assert fad.score_individual(str(tmp_path / "base.npz"), evald, tmp_path / "indiv.csv") == out   # resume-by-cache

    g5 = golden["g5"]
    clean = tmp_path / "clean"
    (clean / "embeddings" / model).mkdir(parents=True)
    for i, blk in enumerate(R.ragged_files(g5["seed"], g5["n_files"], d)):
        np.save(clean / "embeddings" / model / f"c{i}.npy", blk)
    mu_c, cov_c = fad.load_stats(clean)
    z = np.load(golden_dir / "g5_stats.npz")
    np.testing.assert_allclose(mu_c, z["mu"], rtol=1e-10, atol=1e-12)    # merge order differs: glob order is fs-dependent
    np.testing.assert_allclose(cov_c, z["cov"], rtol=0, atol=2e-6 * np.abs(z["cov"]).max())
    assert sorted(p.name for p in (clean / "stats" / model).glob("*")) == g5["cache_files"]
    assert str(np.load(clean / "stats" / model / "mu.npy").dtype) == g5["mu_dtype"]
    assert str(np.load(clean / "stats" / model / "cov.npy").dtype) == g5["cov_dtype"]
    mu_again, cov_again = fad.load_stats(clean)
    assert np.array_equal(mu_c, mu_again) and np.array_equal(cov_c, cov_again)
    mu_n, cov_n = fad.load_stats(str(tmp_path / "base.npz"))
    assert np.array_equal(mu_n, mu_b) and np.array_equal(cov_n, cov_b)
    with pytest.raises(ValueError):
        F.FrechetAudioDistance(_Toy("other-model"), load_model=False).load_stats(str(tmp_path / "base.npz"))
    score = fad.score(str(tmp_path / "base.npz"), clean)
    assert abs(score - g5["fad_clean_vs_npz"]) / g5["fad_clean_vs_npz"] < 1e-6


def test_two_frame_songs_config5_shape_g8(F, golden):
    from fadtk_amd import hip
    g = golden["g8"]
    d = g["d"]
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], d)
    sg = R.songs(g["songs_seed"], g["n_songs"], g["rows"], d)
    rows = np.concatenate(sg)
    offs = np.arange(0, 2 * g["n_songs"] + 1, 2)
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert (status == 0).all()
    np.testing.assert_allclose(scores, g["scores"], rtol=1e-6)
    import torch
    scores_dev, _ = hip.frechet_batched(mu_b, cov_b, torch.from_numpy(rows).cuda(), offs, mean_mode=1)
    np.testing.assert_allclose(scores_dev, scores, rtol=1e-12)
    # a baseline that already lives in HBM is used in place (no upload per call); with host rows it is brought back
    mu_d, cov_d = torch.from_numpy(mu_b).cuda(), torch.from_numpy(cov_b).cuda()
    scores_res, st = hip.frechet_batched(mu_d, cov_d, torch.from_numpy(rows).cuda(), offs, mean_mode=1)
    assert (st == 0).all() and np.array_equal(scores_res, scores_dev)
    scores_mix, _ = hip.frechet_batched(mu_d, cov_d, rows, offs, mean_mode=1)
    assert np.array_equal(scores_mix, scores)


def test_multi_frame_songs_g8(F, golden):
    from fadtk_amd import hip
    m = golden["g8"]["multi"]
    d = m["d"]
    mu_b, cov_b = R.baseline_stats(m["base_seed"], m["base_n"], d)
    sg = R.songs(m["songs_seed"], m["n_songs"], m["rows"], d)
    sg_with_bad = sg[:3] + [sg[0][:1], sg[0][:0]] + sg[3:]                  # 1-frame and empty songs in the middle
    rows = np.concatenate(sg_with_bad)
    offs = np.concatenate([[0], np.cumsum([s.shape[0] for s in sg_with_bad])])
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert status[3] == -6 and status[4] == -6 and np.isnan(scores[3]) and np.isnan(scores[4])
    keep = [i for i in range(len(sg_with_bad)) if i not in (3, 4)]
    assert (status[keep] == 0).all()
    np.testing.assert_allclose(scores[keep], m["scores"], rtol=1e-6)


def test_gram_path_matches_oracle_for_3_to_64_frames(F):
    """3 <= n <= 64 frames: n x n Gram matrix + Jacobi eigenvalues instead of a D x D root per song."""
    from fadtk_amd import hip
    d = 96
    mu_b, cov_b = R.baseline_stats(41, 5 * d, d)
    rows_per_song = [3, 4, 7, 16, 33, 63, 64, 65, 10, 5]              # 65 frames -> the D x D iteration
    sg = R.songs(42, len(rows_per_song), rows_per_song, d)
    sg[3][5:] = sg[3][4]                                              # repeated frames: extra rank deficiency
    rows = np.concatenate(sg)
    offs = np.concatenate([[0], np.cumsum(rows_per_song)])
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert (status == 0).all()
    want = O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False)
    np.testing.assert_allclose(scores, want, rtol=1e-6)


def test_per_song_routes_mixed_in_one_call(F):
    """Two-frame songs (closed form, scored on the device), songs of both Gram routes, a D x D song and songs that cannot be scored, interleaved
    in one call: every route writes only its own songs, with host rows, device rows and a device-resident baseline alike."""
    import torch
    from fadtk_amd import hip
    d = 96
    mu_b, cov_b = R.baseline_stats(51, 6 * d, d)
    rows_per_song = [2, 5, 2, 1, 130, 2, 0, 80, 2, 33]
    sg = R.songs(52, len(rows_per_song), rows_per_song, d)
    rows = np.concatenate([s for s in sg if s.shape[0]])
    offs = np.concatenate([[0], np.cumsum(rows_per_song)])
    ok = [i for i, n in enumerate(rows_per_song) if n >= 2]
    want = O.individual_scores(mu_b, cov_b, [sg[i] for i in ok], run_sqrtm=False)
    dev_rows = torch.from_numpy(rows).cuda()
    for base, rr in ((( mu_b, cov_b), rows), ((mu_b, cov_b), dev_rows),
                     ((torch.from_numpy(mu_b).cuda(), torch.from_numpy(cov_b).cuda()), dev_rows)):
        scores, status = hip.frechet_batched(base[0], base[1], rr, offs, mean_mode=1)
        assert status[3] == -6 and status[6] == -6 and np.isnan(scores[3]) and np.isnan(scores[6])
        assert (status[ok] == 0).all()
        np.testing.assert_allclose(scores[ok], want, rtol=1e-6)


def test_gram_iteration_path_matches_oracle_for_65_to_d_frames(F):
    """64 < n <= D frames: the n x n Gram matrix goes through the batched Newton-Schulz iteration (padded to the longest song of
    the call, deflated by the null vector of the centring) instead of the rank-deficient D x D product: songs of different
    lengths in one call, one with repeated frames (extra rank deficiency), one whose frames are all equal (no spread at all)."""
    from fadtk_amd import hip
    d = 256
    mu_b, cov_b = R.baseline_stats(61, 5 * d, d)
    rows_per_song = [65, 100, 128, 129, 200, 256, 90, 70]
    sg = R.songs(62, len(rows_per_song), rows_per_song, d)
    sg[6][40:] = sg[6][39]                                            # 51 copies of one frame
    sg[7][:] = sg[7][0]                                               # a constant song: Sigma_s = 0
    rows = np.concatenate(sg)
    offs = np.concatenate([[0], np.cumsum(rows_per_song)])
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert (status == 0).all(), status
    want = O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False)
    # the reference's eig on a rank-deficient product returns the D - n + 1 zero eigenvalues as +-1e-16 and takes their roots
    # (fad.py:91-92): it is itself only good to a few 1e-8 here
    np.testing.assert_allclose(scores[:6], want[:6], rtol=2e-7)
    np.testing.assert_allclose(scores[6:], want[6:], rtol=1e-6)


def test_symmetric_dxd_route_and_its_fallbacks(F):
    """Songs of D+1 .. 8D frames take the symmetric form (B = sqrt(Sigma_b) once, cov(Xc B) per song); longer songs in the same
    call, and every song when Sigma_b is singular (its root does not converge), keep the product Sigma_b Sigma_s: all of them must
    agree with the oracle."""
    from fadtk_amd import hip
    d = 64
    rows_per_song = [100, 600, 65, 513, 512, 300]                    # 600 and 513 frames: more than 8 D
    sg = R.songs(72, len(rows_per_song), rows_per_song, d)
    rows = np.concatenate(sg)
    offs = np.concatenate([[0], np.cumsum(rows_per_song)])
    mu_b, cov_b = R.baseline_stats(71, 5 * d, d)
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert (status == 0).all(), status
    np.testing.assert_allclose(scores, O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False), rtol=1e-9)
    mu_s, cov_s = R.baseline_stats(73, d // 2, d)                     # 32 rows at D = 64: a singular baseline
    scores, status = hip.frechet_batched(mu_s, cov_s, rows, offs, mean_mode=1)
    assert (status == 0).all(), status
    np.testing.assert_allclose(scores, O.individual_scores(mu_s, cov_s, sg, run_sqrtm=False), rtol=1e-6)


def test_score_inf_golden_g6(F, golden, tmp_path):
    g = golden["g6"]
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], g["d"])
    rows = R.normal_rows(g["rows_seed"], g["n"], g["d"], 1.1, 0.05)
    np.savez(tmp_path / "b.npz", **{"toy.mu": mu_b, "toy.cov": cov_b})
    files = []
    for i in range(4):
        np.save(tmp_path / f"e{i}.npy", rows[i * 500:(i + 1) * 500])
        files.append(tmp_path / f"e{i}.npy")
    fad = F.FrechetAudioDistance(_Toy("toy"), load_model=False)
    np.random.seed(0)
    res = fad.score_inf(str(tmp_path / "b.npz"), files)
    assert [p[0] for p in res.points] == [p[0] for p in g["points"]]
    np.testing.assert_allclose([p[1] for p in res.points], [p[1] for p in g["points"]], rtol=FAD_BAR / 10)
    assert res.score == pytest.approx(g["score"], rel=FAD_BAR)
    assert res.r2 == pytest.approx(g["r2"], rel=1e-3)


def test_moments_allreduce_through_rccl_single_rank(F):
    """fad_moments_allreduce runs ncclAllReduce over the packed statistics with the caller's communicator.  One GPU
    here, so the communicator has one rank and the sum must leave the statistics unchanged -- this checks the
    symbol lookup, the datatype/op constants and the in-place call, not the scaling."""
    import ctypes as C
    import torch
    from pathlib import Path
    from fadtk_amd.hip import Moments
    rccl = C.CDLL(str(Path(torch.__file__).parent / "lib" / "librccl.so"))      # the RCCL torch itself uses
    comm = C.c_void_p()
    devs = (C.c_int * 1)(0)
    rccl.ncclCommInitAll.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_int)]
    assert rccl.ncclCommInitAll(C.byref(comm), 1, devs) == 0
    try:
        x = structured_rows(5, 3000, 256, np.float16)
        with Moments(256) as m:
            m.update(x)
            before = m.export()
            m.allreduce_rccl(comm.value)
            torch.cuda.synchronize()
            after = m.export()
        np.testing.assert_array_equal(before, after)
        assert before[0] == 3000
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_moments_bind_keeps_statistics_in_callers_buffer(F):
    """fad_moments_bind: the packed statistics live in the caller's tensor (what a collective then reduces in place);
    binding resets, updates land in the tensor, reset + update overwrites, and unrelated memory stays untouched."""
    import torch
    from fadtk_amd.hip import Moments
    d = 128
    x = structured_rows(3, 2500, d, np.float16)
    with Moments(d) as ref, Moments(d) as m:
        ref.update(x)
        want = ref.export()
        plen = m.packed_len
        buf = torch.full((plen + 8,), 7.0, dtype=torch.float64, device="cuda")
        m.update(x[:100])                                   # statistics from before the bind are dropped
        m.bind(buf[:plen])
        assert m.count == 0 and float(buf[0]) == 0.0        # (reading settles the pending reset into the buffer)
        m.update(x)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(buf[:plen].cpu().numpy(), want)
        np.testing.assert_array_equal(m.export(), want)
        assert bool((buf[plen:] == 7.0).all())
        buf[:plen] *= 2.0                                    # e.g. an all-reduce over two identical ranks
        mu, cov, n = m.finalize()
        mu_r, cov_r, n_r = ref.finalize()
        assert n == 2 * n_r
        np.testing.assert_allclose(mu, mu_r, rtol=1e-14)
        m.reset(); m.update(x[:1000]); m.update(x[1000:])
        torch.cuda.synchronize()
        # (two blocks group the fp32 partial sums differently: fp32-level agreement, as in the streaming test)
        np.testing.assert_allclose(buf[:plen].cpu().numpy(), want, rtol=0, atol=1e-6 * np.abs(want).max())


@pytest.mark.parametrize("d,n", [(256, 3000), (384, 4500), (512, 6000), (768, 9000), (1024, 12000)])
def test_frechet_nine_launch_chain_matches_float32_chain_and_oracle(F, monkeypatch, d, n):
    """Round 3: for D in {256, 512, 768, 1024} the square root runs as nine launches -- exact products on the int8 MFMA, iteration
    on split-float16 operands (csrc/ns_fast.h).  From packed moments (the bench's route, float16 frames, the reference's float16
    mean term) and from host matrices it must give what round 2's float32 chain gives (FAD_FRECHET_FAST=0, new thread = new
    workspace), what the all-float64 iteration gives, and the oracle's value far below the 1e-4 bar."""
    import threading
    import torch
    from fadtk_amd import hip
    rng = np.random.default_rng(3 * d + n)
    a = (rng.standard_normal((n, d)) * (0.6 + 0.8 * rng.random(d))).astype(np.float16)
    b = (1.04 * rng.standard_normal((n, d)) * (0.6 + 0.8 * rng.random(d)) + 0.015).astype(np.float16)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()

    def from_moments():
        with hip.Moments(d) as ma, hip.Moments(d) as mb:
            hip.Moments.update_multi([ma, mb], [ta, tb])
            return hip.frechet_from_moments(ma, mb, mean_dtype=0)

    fad, diag = from_moments()
    assert diag["converged"] == 3, diag
    fad2, diag2 = from_moments()                       # batch sized by the first call: same decisions, same bits
    assert fad2 == fad and diag2["iters"] == diag["iters"]
    out = {}

    def other(key):
        out[key] = from_moments()
    monkeypatch.setenv("FAD_FRECHET_FAST", "0")
    t = threading.Thread(target=other, args=("f32",)); t.start(); t.join()
    monkeypatch.setenv("FAD_FRECHET_MIXED", "0")
    t = threading.Thread(target=other, args=("f64",)); t.start(); t.join()
    assert out["f32"][1]["converged"] == 3 and out["f64"][1]["converged"] in (1, 2)
    for key in ("f32", "f64"):
        assert abs(diag["tr_sqrt"] - out[key][1]["tr_sqrt"]) <= 2e-10 * abs(diag["tr_sqrt"]), (key, diag, out[key][1])
        assert abs(diag["tr1"] - out[key][1]["tr1"]) <= 1e-12 * abs(diag["tr1"]) and diag["mean_term"] == out[key][1]["mean_term"]
    ref = O.fad_between(a, b)                          # the reference's path on the same float16 frames (float16 mean term)
    assert abs(fad - ref) <= 2e-6 * abs(ref), (fad, ref)
    # host matrices through fad_frechet (caller-given Sigma: no moments, the kernels digitise the matrices as they come)
    monkeypatch.delenv("FAD_FRECHET_FAST"); monkeypatch.delenv("FAD_FRECHET_MIXED")
    m1, c1, m2, c2 = _pair_stats(a.astype(np.float64), b.astype(np.float64))
    fad_h, diag_h = hip.frechet(m1, c1, m2, c2)
    assert diag_h["converged"] == 3
    ref_h = O.frechet_distance(m1, c1, m2, c2, run_sqrtm=False)
    assert abs(fad_h - ref_h) <= 1e-7 * abs(ref_h)


def test_frechet_nine_launch_chain_degenerate_inputs(F):
    """What must NOT stay on the fast chain: a zero covariance, NaNs, a hopeless spectrum, fewer than two rows -- the same
    answers / errors as before, through the float64 route."""
    import torch
    from fadtk_amd import hip
    d = 256
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2000, d))
    m, c = x.mean(0), np.cov(x, rowvar=False)
    fad, diag = hip.frechet(m, c, m, np.zeros((d, d)))                       # zero product: root 0
    assert abs(fad - np.trace(c)) <= 1e-9 * np.trace(c)
    cn = c.copy(); cn[3, 4] = np.nan
    with pytest.raises(ValueError):
        hip.frechet(m, cn, m, c)
    y = (x * (np.arange(1, d + 1) ** -1.5)).astype(np.float64)             # decaying spectrum: participation ratio far below d/4
    my, cy = y.mean(0), np.cov(y, rowvar=False)
    fad_y, diag_y = hip.frechet(my, cy, 1.01 * my, 1.1 * cy)
    assert diag_y["converged"] in (1, 2)
    ref = O.frechet_distance(my, cy, 1.01 * my, 1.1 * cy, run_sqrtm=False)
    assert abs(fad_y - ref) <= 1e-6 * abs(ref)
    with hip.Moments(d) as ma, hip.Moments(d) as mb:
        ma.update(torch.from_numpy(x.astype(np.float16)).cuda())
        mb.update(torch.from_numpy(x[:1].astype(np.float16)).cuda())
        with pytest.raises(AssertionError):
            hip.frechet_from_moments(ma, mb)


@pytest.mark.parametrize("d,dtype", [(512, np.float16), (128, np.float16), (64, np.float32)])
def test_score_inf_points_on_device_match_the_sequential_route(F, d, dtype):
    """score_inf's batched route (frames in HBM, gather on the device, eight resamples per moments launch, distances straight from
    the accumulators with only the resample's mean rounded to the frames' dtype) against the point-by-point route through
    calc_embd_statistics / calc_frechet_distance and against the oracle's arithmetic (fad.py:333-341)."""
    rng = np.random.default_rng(d)
    rows = (1.1 * rng.standard_normal((6000, d)) * (0.5 + rng.random(d)) + 0.3).astype(dtype)
    base = rng.standard_normal((8000, d)) * (0.5 + rng.random(d)) + 0.25
    mu_b, cov_b = base.mean(0), np.cov(base, rowvar=False)
    fad = F.FrechetAudioDistance(_Toy("toy"), load_model=False)
    ns = [int(n) for n in np.linspace(500, rows.shape[0], 11)]
    picks = [rng.integers(0, rows.shape[0], size=n) for n in ns]
    got = fad._score_inf_points_on_device(mu_b, cov_b, rows, picks)
    assert got is not None and len(got) == len(picks)
    seq = fad._score_inf_points_sequential(mu_b, cov_b, rows, picks)
    np.testing.assert_allclose(got, seq, rtol=2e-6)
    for idx, v in list(zip(picks, got))[::5]:
        mu_e, cov_e = O.embd_statistics(rows[idx])
        ref = O.frechet_distance(mu_b, cov_b, mu_e, cov_e, run_sqrtm=False)
        assert abs(v - ref) <= FAD_BAR / 10 * abs(ref)


@pytest.mark.parametrize("d,frames", [(128, [129, 300, 2250, 140, 777, 500, 200, 350, 9000]), (256, [257, 600, 300, 1500]), (384, [900, 500]), (512, [1100, 513]), (768, [1500, 900]),
                                      (1024, [2100])])
def test_songs_full_rank_route_on_the_matrix_pipes(F, monkeypatch, d, frames):
    """Songs with at least D + 1 frames, D in {128, 256, 512, 768, 1024}: the eight-launch chain of the single pair, batched over
    the songs (frechet.hip: fast_songs) -- exact int8-MFMA products, split-float16 Newton-Schulz, one correction, decided per song
    on the host.  Against the oracle (fad.py:373-378) and against the float64 routes (FAD_SONG_FAST=0); a song without any spread
    and a song whose spectrum the chain does not accept ride along and must come back through the float64 routes."""
    from fadtk_amd import hip
    rng = np.random.default_rng(d)
    mu_b, cov_b = R.baseline_stats(700 + d, 6 * d, d)
    sg = [(rng.standard_normal((n, d)) * (0.7 + 0.6 * rng.random(d)) + 0.1 * rng.standard_normal(d)).astype(np.float16) for n in frames]
    flat = np.tile(sg[0][:1], (d + 5, 1))                                       # all frames equal: Sigma_s = 0
    steep = (rng.standard_normal((2 * d, d)) * np.arange(1, d + 1) ** -1.5).astype(np.float16)      # decaying spectrum: not for this chain
    sg = sg + [flat, steep]
    rows = np.concatenate(sg)
    offs = np.concatenate([[0], np.cumsum([s.shape[0] for s in sg])])
    monkeypatch.setenv("FAD_SONG_FAST", "2")                                  # strict: an error if the chain accepts no song at all
    want = O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False)
    # iteration on 128 x 128 tiles (ns_fast_big.h) / on 32 x 32 tiles; D = 128 first through its resident kernel (ns_fast_res.h)
    # (FAD_SONG_RES: 2 / unset = products and iteration in one workgroup per song, 1 = the iteration only, 0 = the batched kernels)
    variants = ([("8", "2"), ("8", "1")] if d == 128 else []) + [("1", "0"), ("0", "0")]
    for big_min, res in variants:
        monkeypatch.setenv("FAD_SONG_BIG", big_min)
        monkeypatch.setenv("FAD_SONG_RES", res)
        scores_v, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
        assert (status == 0).all(), (big_min, res, status)
        np.testing.assert_allclose(scores_v, want, rtol=2e-6, err_msg=f"FAD_SONG_BIG={big_min} FAD_SONG_RES={res}")
    monkeypatch.delenv("FAD_SONG_BIG"); monkeypatch.delenv("FAD_SONG_RES")
    scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)          # the defaults
    assert (status == 0).all(), status
    np.testing.assert_allclose(scores, want, rtol=2e-6)
    monkeypatch.setenv("FAD_SONG_FAST", "0")                                  # the float64 routes on the same call
    scores64, status64 = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
    assert (status64 == 0).all()
    np.testing.assert_allclose(scores, scores64, rtol=2e-6)
    with pytest.raises(RuntimeError):                                         # only songs the chain cannot take: strict mode must say so
        monkeypatch.setenv("FAD_SONG_FAST", "2")
        hip.frechet_batched(mu_b, cov_b, np.concatenate([flat, steep]), [0, flat.shape[0], flat.shape[0] + steep.shape[0]], mean_mode=1)


@pytest.mark.parametrize("dtype,n,d,offset", [(np.float16, 60000, 256, 3.0), (np.float16, 20001, 128, 0.5), (np.float32, 30000, 96, 3.0), (np.float16, 513, 512, 10.0),
                                              (np.float16, 70001, 512, 3.0)])      # (72 MB of host rows: copied and accumulated in ~24 MB pieces)
def test_calc_embd_statistics_returns_numpys_own_mean_for_frames_with_an_offset(F, dtype, n, d, offset):
    """np.mean(embd_lst, axis=0) (fadtk/fad.py:48) adds the rows one after the other in float32: for frames with an offset the float16
    result is NOT the rounded exact mean in a few dimensions (worth 2e-5 .. 5e-4 of a small FAD at config-3 size).  calc_embd_statistics
    carries numpy's running sums on the GPU (fad_moments_set_reference_mean) and returns numpy's mean bit for bit -- from host rows (staged
    in blocks), from a device tensor, and across two updates of one handle; the covariance stays the exact one."""
    import torch
    from fadtk_amd import hip
    rng = np.random.default_rng(n + d)
    x = (np.maximum(rng.standard_normal((n, d)) * 0.3 + offset, 0)).astype(dtype)
    want_mu = np.mean(x, axis=0)
    mu, cov = F.calc_embd_statistics(x)
    assert mu.dtype == want_mu.dtype
    np.testing.assert_array_equal(mu, want_mu)
    want_cov = np.cov(x, rowvar=False)
    np.testing.assert_allclose(cov, want_cov, rtol=0, atol=2e-6 * np.abs(want_cov).max())
    mu_t, _ = F.calc_embd_statistics(torch.from_numpy(x).cuda())
    np.testing.assert_array_equal(mu_t, want_mu)
    if dtype == np.float16 and offset >= 3.0 and n >= 60000:
        exact = x.astype(np.float64).mean(0).astype(np.float16)
        assert (exact != want_mu).any(), "the case should show the difference this test is about"
    with hip.Moments(d) as acc:                       # two updates: the running sums are carried like numpy carries them over the rows
        acc.set_reference_mean(True)
        acc.update(x[: n // 3]); acc.update(x[n // 3:])
        mu2, _, _ = acc.finalize()
        np.testing.assert_array_equal(mu2.astype(np.float32).astype(dtype), want_mu)
        acc.reset()                                   # ... and a reset starts them again
        acc.update(x[: n // 2])
        mu3, _, _ = acc.finalize()
        np.testing.assert_array_equal(mu3.astype(np.float32).astype(dtype), np.mean(x[: n // 2], axis=0))
    with hip.Moments(d) as plain:                     # the switch off: the exact mean, as before
        plain.update(x)
        mu4, _, _ = plain.finalize()
        np.testing.assert_allclose(mu4, x.astype(np.float64).mean(0), rtol=2e-6)      # (float16 frames: column sums of bounded float32 runs)


@pytest.mark.parametrize("d,frames", [(128, 2250), (512, 1200)])
def test_songs_scaled_steps_agree_with_plain_steps_and_with_the_oracle(F, monkeypatch, d, frames):
    """Round 4: scaled Newton-Schulz steps per song (ns_check.h: step scale from a lower bound of sqrt(lambda / c) that the check of
    every iteration refines from the residual and caps by it).  Songs whose variances differ by dimension (products with condition
    numbers of a few thousand, bench.py's per-song extras): scaled (default), plain (FAD_SONG_SCALED=0) and scaled from a start thirty
    times too small / three times too large (FAD_SONG_L0_SCALE) all give the oracle's scores (fad.py:373-378) -- and the chain itself
    accepts every song (FAD_SONG_FAST=2 raises if it accepts none; status 0 for all)."""
    from fadtk_amd import hip
    rng = np.random.default_rng(40 + d)
    n_songs = 12
    scale = 0.5 + rng.random(d)
    base = rng.standard_normal((20 * d, d)) * scale * 1.05 + 0.01
    mu_b, cov_b = base.mean(0), np.cov(base, rowvar=False)
    sg = [(rng.standard_normal((frames, d)) * scale).astype(np.float16) for _ in range(n_songs)]
    rows = np.concatenate(sg)
    offs = np.arange(0, n_songs * frames + 1, frames)
    want = O.individual_scores(mu_b, cov_b, sg[:4], run_sqrtm=False)
    monkeypatch.setenv("FAD_SONG_FAST", "2")
    got = {}
    for name, env in (("scaled", {}), ("plain", {"FAD_SONG_SCALED": "0"}), ("start / 30", {"FAD_SONG_L0_SCALE": "0.0167"}), ("start x 3", {"FAD_SONG_L0_SCALE": "1.5"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        scores, status = hip.frechet_batched(mu_b, cov_b, rows, offs, mean_mode=1)
        for k in env:
            monkeypatch.delenv(k)
        assert (status == 0).all(), (name, status)
        np.testing.assert_allclose(scores[:4], want, rtol=2e-6, err_msg=name)
        got[name] = scores
    for name in ("plain", "start / 30", "start x 3"):
        np.testing.assert_allclose(got[name], got["scaled"], rtol=1e-7, err_msg=name)


def test_frechet_multi_job_matches_single_scores(F):
    """fad_frechet_from_moments_multi_begin / _multi_end: B pairs through ONE batched chain give what fad_frechet_from_moments gives
    pair by pair -- flat pairs (accepted by the batch), a decaying-spectrum pair and a low-rank pair (handed to the single route),
    and a second batch on the same thread (the learnt iteration count is in use then)."""
    import torch
    from fadtk_amd import hip
    d = 512
    rng = np.random.default_rng(2024)
    decay = (np.arange(1, d + 1) ** -0.75)
    sets = []
    for k in range(4):
        a = rng.standard_normal((9000, d)).astype(np.float16)
        b = ((1.0 + 0.03 * k) * rng.standard_normal((9500, d)) + 0.01 * k).astype(np.float16)
        sets.append((a, b))
    sets.append(((rng.standard_normal((9000, d)) * decay).astype(np.float16), (1.1 * rng.standard_normal((9000, d)) * decay).astype(np.float16)))
    sets.append((rng.standard_normal((9000, d)).astype(np.float16), rng.standard_normal((300, d)).astype(np.float16)))      # fewer rows than dimensions
    handles = []
    for a, b in sets:
        ma, mb = hip.Moments(d), hip.Moments(d)
        ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
        handles.append((ma, mb))
    want = [hip.frechet_from_moments(ma, mb, mean_dtype=0) for ma, mb in handles]
    for rep in range(2):
        got = hip.FrechetMultiJob(handles, mean_dtype=0).result()
        assert len(got) == len(handles)
        for k, ((f, dg), (fw, dw)) in enumerate(zip(got, want)):
            # (the decaying pair stays on the chain since round 5 -- scaled steps -- on 128 x 128 tiles in the batch and on 32 x 32 tiles alone:
            #  two float32-class iterations whose corrected traces agree to the second-order term the verification estimates)
            tol = 2e-9 if k != 4 else 2e-6
            assert abs(f - fw) <= tol * abs(fw), (k, f, fw, dg, dw)
            assert abs(dg["tr_sqrt"] - dw["tr_sqrt"]) <= (1e-9 if k != 4 else 1e-8) * abs(dw["tr_sqrt"])
        assert [dg["route"] for dg, _ in [(g[1], None) for g in got[:4]]] == [2, 2, 2, 2]           # the flat pairs stayed on the batched chain
    for (a, b), (f, _) in zip(sets[:2] + sets[4:], got[:2] + got[4:]):                              # ... and against the oracle
        ref = O.fad_between(a, b)
        assert abs(f - ref) <= 2e-6 * abs(ref)
    # the largest batch a job takes (FAD_MULTI_MAX_PAIRS = 32; 8 until round 5): the six pairs over and over -- flat, decaying and low-rank ones
    # side by side in one chain -- and one pair more than that is refused
    many = (handles * 6)[:hip.FrechetMultiJob.MAX_PAIRS]
    got32 = hip.FrechetMultiJob(many, mean_dtype=0).result()
    assert len(got32) == hip.FrechetMultiJob.MAX_PAIRS == 32
    for k, (f, dg) in enumerate(got32):
        fw = want[k % len(handles)][0]
        assert abs(f - fw) <= (2e-9 if k % len(handles) != 4 else 2e-6) * abs(fw), (k, f, fw, dg)
    with pytest.raises(ValueError):
        hip.FrechetMultiJob((handles * 6)[:33], mean_dtype=0)
    two = hip.FrechetMultiJob(handles[1:3], mean_dtype=0).result()                                 # a smaller batch out of the same slot
    assert abs(two[0][0] - want[1][0]) <= 2e-9 * abs(want[1][0]) and abs(two[1][0] - want[2][0]) <= 2e-9 * abs(want[2][0])
    for ma, mb in handles:
        ma.close(); mb.close()


@pytest.mark.parametrize("d, counts", [(256, (5, 11)), (768, (3, 9, 15)), (1024, (3, 5, 9))])
def test_frechet_multi_job_tile_shapes_and_leftover_pairs(F, d, counts):
    """The batched chain picks its workgroup tile by the launch's size (csrc/big_slots.h: 128 x 64 tiles while the wide ones would leave CUs
    with a lone workgroup -- T / FIRST and U cross that line at different batch sizes) and cuts the pairs left over after the groups of eight
    over all XCDs: batches on either side of both lines, with and without leftovers, give what the single route gives pair by pair, all of
    them on the chain."""
    import torch
    from fadtk_amd import hip
    rng = np.random.default_rng(31 + d)
    n = 4 * d
    handles = []
    for k in range(max(counts)):
        a = rng.standard_normal((n, d)).astype(np.float16)
        b = ((1.0 + 0.02 * (k % 7)) * rng.standard_normal((n + 64 * (k % 3), d)) + 0.01 * (k % 5)).astype(np.float16)
        ma, mb = hip.Moments(d), hip.Moments(d)
        ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
        handles.append((ma, mb))
    want = [hip.frechet_from_moments(ma, mb, mean_dtype=0) for ma, mb in handles]
    assert all(dw["route"] == 2 for _, dw in want)
    for count in counts:
        got = hip.FrechetMultiJob(handles[:count], mean_dtype=0).result()
        assert len(got) == count
        for k, ((f, dg), (fw, dw)) in enumerate(zip(got, want)):
            assert dg["route"] == 2, (d, count, k, dg)
            assert abs(f - fw) <= 2e-9 * abs(fw), (d, count, k, f, fw)
            assert abs(dg["tr_sqrt"] - dw["tr_sqrt"]) <= 1e-9 * abs(dw["tr_sqrt"])
    for ma, mb in handles:
        ma.close(); mb.close()


def test_frechet_multi_job_hands_declined_pairs_to_one_batched_float64_iteration(F):
    """A batch whose pairs the low-precision chain declines (spectra k^-2 and steeper: condition 1e9 .. 1e13 of Sigma_1 Sigma_2) is closed by
    ONE float64 Newton-Schulz iteration over all of them (round 5; fad_frechet_multi_end) instead of pair by pair: same distances as the
    blocking call gives each pair (both are float64 iterations on the same (mu, Sigma): 1e-9), the reference's eig value (fad.py:91-92) to
    1e-6, `route` 0 in the diagnostics -- for a batch of declined pairs only, and for one that mixes them with flat pairs the chain keeps."""
    import torch
    from fadtk_amd import hip
    d = 256
    rng = np.random.default_rng(909)
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    def pair(power, n1, n2, scale=1.0):
        lam = np.arange(1, d + 1) ** (-power / 2.0)
        a = (((rng.standard_normal((n1, d)) * lam) @ q.T) * scale).astype(np.float16)
        b = (((1.07 * rng.standard_normal((n2, d)) * lam) @ q.T + 0.01) * scale).astype(np.float16)
        return a, b
    steep = [pair(2.0, 9000, 8000), pair(2.5, 12000, 9000), pair(2.0, 7000, 7000, 30.0), pair(3.0, 20000, 15000), pair(2.2, 9000, 9000, 1e-2)]
    flat = [pair(0.0, 6000, 6500), pair(0.0, 5000, 5000)]
    def handles_of(sets):
        out = []
        for a, b in sets:
            ma, mb = hip.Moments(d), hip.Moments(d)
            ma.update(torch.from_numpy(a).cuda()); mb.update(torch.from_numpy(b).cuda())
            out.append((ma, mb))
        return out
    hs, hf = handles_of(steep), handles_of(flat)
    single = [hip.frechet_from_moments(ma, mb, mean_dtype=-1) for ma, mb in hs + hf]
    assert all(dg["route"] == 0 for _, dg in single[:len(hs)]), [dg["route"] for _, dg in single]      # (the case list must stay declined)
    for rep in range(2):                                                        # (the second batch runs on the learnt launch count)
        got = hip.FrechetMultiJob(hs, mean_dtype=-1).result()
        for k, ((f, dg), (fw, dw)) in enumerate(zip(got, single[:len(hs)])):
            assert dg["route"] == 0 and dg["converged"] in (1, 2), (k, dg)
            # (two float64 iterations on (mu, Sigma) that two kernels formed from the same sums: an ulp apart, and a k^-3 product -- condition
            #  1e13 -- turns that into 2e-10 of the traces; seen 5.4e-10 absolute at tr Sigma_1 + tr Sigma_2 = 2.6)
            assert abs(f - fw) <= 1e-9 * abs(fw) + 5e-10 * (dw["tr1"] + dw["tr2"]), (k, f, fw, dg, dw)
    mixed = hip.FrechetMultiJob([hf[0], hs[0], hs[3], hf[1], hs[1]], mean_dtype=-1).result()
    for (f, dg), (fw, dw), route in zip(mixed, [single[5], single[0], single[3], single[6], single[1]], [2, 0, 0, 2, 0]):
        assert dg["route"] == route, (dg, route)
        assert abs(f - fw) <= (1e-9 if route == 0 else 2e-9) * abs(fw) + 5e-10 * (dw["tr1"] + dw["tr2"]), (f, fw, dg)
    for (a, b), (f, _) in zip(steep[:3], got[:3]):                              # ... and against the reference's formula
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        ref = O.frechet_distance(a64.mean(0), np.cov(a64, rowvar=False), b64.mean(0), np.cov(b64, rowvar=False), run_sqrtm=False)
        assert abs(f - ref) <= 1e-6 * abs(ref), (f, ref)
    for ma, mb in hs + hf:
        ma.close(); mb.close()


def test_frechet_fast_chain_forms_sigma1_sigma2_for_an_asymmetric_caller_matrix(F, monkeypatch):
    """fad_frechet on host matrices -- a slightly ASYMMETRIC cov2 must give the same value on the eight-launch chain as
    on the float64 route (both form Sigma_1 Sigma_2, fad.py:88); before, the chain digitised cov2 as its own transpose."""
    from fadtk_amd import hip
    d = 512
    rng = np.random.default_rng(31)
    a = rng.standard_normal((4 * d, d)); b = 1.05 * rng.standard_normal((4 * d, d))
    c1 = np.cov(a, rowvar=False); c2 = np.cov(b, rowvar=False)
    c2 = c2 + 2e-3 * np.triu(rng.standard_normal((d, d)), 1)          # asymmetric perturbation
    mu = np.zeros(d)
    f_fast, dg = hip.frechet(mu, c1, mu, c2)
    monkeypatch.setenv("FAD_FRECHET_MIXED", "0")
    import threading
    out = {}
    t = threading.Thread(target=lambda: out.update(r=hip.frechet(mu, c1, mu, c2)))      # (the knob is read once per thread)
    t.start(); t.join()
    f64, dg64 = out["r"]
    assert dg64["route"] == 0
    assert abs(f_fast - f64) <= 1e-7 * abs(f64), (f_fast, f64, dg, dg64)
    f_t, _ = hip.frechet(mu, c1, mu, c2.T.copy())
    assert abs(f_t - f64) <= 1e-2 * abs(f64)                          # (sanity: the transposed matrix is a different, nearby problem)


def test_prepared_multi_update_and_result_arrays_are_the_plain_calls(F):
    """Round 6 host-side short cuts for loops over resident sets: ``hip.PreparedMultiUpdate`` (tables built once, ``fad_moments_reset_multi``)
    feeds what ``Moments.update_multi`` feeds -- 20 handles: more than sixteen share ONE launch on the 256-column-slab route -- and
    ``FrechetMultiJob.result_arrays`` returns what ``result`` returns."""
    import torch
    from fadtk_amd import hip
    d, n = 512, 9000
    xs = [torch.from_numpy(structured_rows(300 + i, n + 17 * i, d, np.float16)).cuda() for i in range(20)]
    with contextlib.ExitStack() as es:
        plain = [es.enter_context(hip.Moments(d)) for _ in xs]
        prep = [es.enter_context(hip.Moments(d)) for _ in xs]
        plain[0].set_timing(True)
        hip.Moments.update_multi(plain, xs)
        assert plain[0].last_timing()[2] == 2                         # one launch of the slab kernel for all twenty
        pu = hip.Moments.prepared_update_multi(prep, xs)
        pu.run()
        pu.run(reset=True)                                            # (fed twice, reset in between: the second feed alone counts)
        for a, b, x in zip(plain, prep, xs):
            pa, pb = a.export(), b.export()
            assert pa[0] == pb[0] == x.shape[0]
            np.testing.assert_array_equal(pa, pb)
        x64 = xs[3].cpu().numpy().astype(np.float64)
        np.testing.assert_allclose(plain[3].export()[1 + d:].reshape(d, d), x64.T @ x64, rtol=0, atol=1e-6 * np.abs(x64.T @ x64).max())
        pairs = [(plain[2 * k], plain[2 * k + 1]) for k in range(10)]
        res = hip.FrechetMultiJob(pairs, mean_dtype=0).result()
        vals, diags = hip.FrechetMultiJob(pairs, mean_dtype=0).result_arrays()
        # (two runs of the chain: the thread's launch-count hints may give the second one an iteration more -- the values agree to 1e-9,
        #  not to the bit)
        np.testing.assert_allclose([r[0] for r in res], vals, rtol=1e-9)
        assert set(res[4][1]) == set(diags[4].as_dict()) and res[4][1]["route"] == diags[4].as_dict()["route"]


@pytest.mark.parametrize("d,n_src", [(512, 30000), (768, 20000), (640, 21000), (128, 9000), (256, 6000)])
def test_update_multi_indexed_is_the_update_of_the_gathered_rows(F, d, n_src):
    """``fad_moments_update_multi_indexed`` (round 6: FAD-inf's resamples, fad.py:333-337, without materialising them): the moments of
    rows[idx] -- with replacement, unsorted, lengths that are no multiple of anything -- equal those of the gathered matrix: on the slab
    kernel's indexed loads (D = 512; 768: a Z item, two rows per piece; 640: a ragged last superblock), on the fallback that gathers
    first (D = 128 / 256), with numpy's running-sum means (the walk reads rows[idx] in the resample's order) and without."""
    import torch
    from fadtk_amd import hip
    rng = np.random.default_rng(d)
    x = structured_rows(77, n_src, d, np.float16)
    x[:, 3:9] += np.float16(2.0)                                        # (an offset: the running-sum mean differs from the rounded exact one)
    if d == 768:                                                        # outlier columns: the shift guard's second pass on the indexed route
        x[:, 600:640] = (30.0 + 0.04 * rng.standard_normal((n_src, 40))).astype(np.float16)
    rows = torch.from_numpy(x).cuda()
    sizes = [16 * d + 37, 16 * d + 1, 2 * n_src + 5, 16 * d + 4099]
    idx = [rng.integers(0, n_src, size=s_) for s_ in sizes]
    with contextlib.ExitStack() as es:
        for ref in (False, True):
            got = [es.enter_context(hip.Moments(d)) for _ in sizes]
            want = [es.enter_context(hip.Moments(d)) for _ in sizes]
            for h in got + want:
                h.set_reference_mean(ref)
            got[0].set_timing(True)
            hip.Moments.update_multi_indexed(got, rows, [torch.from_numpy(i.astype(np.int32)).cuda() for i in idx])
            if d >= 512:
                assert got[0].last_timing()[2] == 2
            hip.Moments.update_multi(want, [rows[torch.from_numpy(i).cuda()] for i in idx])
            for g, w, i in zip(got, want, idx):
                pg, pw = g.export(), w.export()
                assert pg[0] == pw[0] == i.size
                np.testing.assert_allclose(pg[1:1 + d], pw[1:1 + d], rtol=1e-12, atol=1e-9)
                M = pw[1 + d:]
                np.testing.assert_allclose(pg[1 + d:], M, rtol=0, atol=1e-9 * np.abs(M).max())
                mg, mw = g.finalize()[0], w.finalize()[0]
                assert np.array_equal(mg, mw)                           # the mean: bit for bit (the walk adds the same rows in the same order)
                if ref:                                                 # ... and numpy's own float16 mean of the gathered matrix
                    assert np.array_equal(mg.astype(np.float32).astype(np.float16), x[i].mean(axis=0))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_moments_tile256_fuzz_pitch_and_edges(F, seed):
    """The slab kernel's loads are range-checked buffer loads (round 6): rows of a last, partial stage and columns of a ragged last superblock
    come back as zeros by the hardware's bounds check, the row pitch may exceed D.  Random D (multiples of 8 in [512, 2048]), row counts that are
    no multiple of anything, a pitch with NaN padding behind every row (never to be read), several matrices of different lengths per launch."""
    import torch
    from fadtk_amd.hip import Moments
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([512, 520, 600, 768, 776, 1024, 1536, 2040, 2048]))
    pad = int(rng.choice([0, 8, 24, 264]))
    ns = [16 * d + int(rng.integers(0, 3000)) for _ in range(int(rng.integers(1, 4)))]
    mats, views = [], []
    for k, n in enumerate(ns):
        full = torch.full((n, d + pad), float("nan"), dtype=torch.float16, device="cuda")
        x = (rng.standard_normal((n, d)) * (1.0 + 0.5 * k) + 0.1 * k).astype(np.float16)
        full[:, :d] = torch.from_numpy(x).cuda()
        mats.append(x); views.append(full[:, :d])                      # a strided view: pitch d + pad
    with contextlib.ExitStack() as es:
        hs = [es.enter_context(Moments(d)) for _ in ns]
        hs[0].set_timing(True)
        Moments.update_multi(hs, views)
        assert hs[0].last_timing()[2] == 2
        for h, x in zip(hs, mats):
            p = h.export()
            x64 = x.astype(np.float64)
            M = x64.T @ x64
            assert p[0] == x.shape[0] and np.isfinite(p).all()
            np.testing.assert_allclose(p[1:1 + d], x64.sum(0), rtol=1e-7, atol=1e-5)
            np.testing.assert_allclose(p[1 + d:].reshape(d, d), M, rtol=0, atol=1e-6 * np.abs(M).max())


@pytest.mark.parametrize("kind", ["two_cluster", "gapped", "low_rank_plus_floor", "rotated_bases", "one_dominant"])
def test_frechet_chain_acceptance_on_spectra_outside_the_power_law_family(F, kind):
    """the acceptance of a score on the chain's verification record is an ESTIMATE (4 x 1/8 |tr(Z P P)| + ||E||^2 ||P||,
    csrc/frechet.hip), emulated and fuzzed on power-law spectra only.  Spectra outside that family -- two clusters three decades apart, a
    gap in a k^-1 decay, rank 100 over a noise floor, the two sets in DIFFERENT bases (a strongly non-normal product), one dominant
    direction -- through the moments handles (single pair and batch), whatever route the library picks: within 1e-5 of the oracle."""
    import torch
    from fadtk_amd import hip
    d, n = 512, 24000
    rng = np.random.default_rng(sum(map(ord, kind)))
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    q2 = q
    k = np.arange(1, d + 1, dtype=np.float64)
    if kind == "two_cluster":
        lam = np.where(k <= d // 2, 1.0, 1e-3)
    elif kind == "gapped":
        lam = k ** -1.0; lam[50:] *= 1e-2
    elif kind == "low_rank_plus_floor":
        lam = np.where(k <= 100, k ** -0.5, 1e-6)
    elif kind == "one_dominant":
        lam = np.where(k == 1, 100.0, k ** -0.5)
    else:                                                           # rotated_bases: k^-0.75 in two bases 0.3 rad apart in random planes
        lam = k ** -0.75
        g, _ = np.linalg.qr(rng.standard_normal((d, d)))
        th = 0.3
        rot = np.eye(d)
        for i in range(0, d - 1, 2):
            rot[i, i] = rot[i + 1, i + 1] = np.cos(th); rot[i, i + 1] = -np.sin(th); rot[i + 1, i] = np.sin(th)
        q2 = q @ (g @ rot @ g.T)
    a = ((rng.standard_normal((n, d)) * np.sqrt(lam)) @ q.T).astype(np.float16)
    b = ((1.04 * rng.standard_normal((n, d)) * np.sqrt(lam)) @ q2.T + 0.003).astype(np.float16)
    ref = O.fad_between(a, b)
    with hip.Moments(d) as ma, hip.Moments(d) as mb:
        hip.Moments.update_multi([ma, mb], [torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()])
        one = hip.frechet_from_moments(ma, mb, mean_dtype=0)
        again = hip.frechet_from_moments(ma, mb, mean_dtype=0)
        batch = hip.FrechetMultiJob([(ma, mb)] * 3, mean_dtype=0).result()
    for f, dg in (one, again, *batch):
        assert abs(f - ref) <= 1e-5 * abs(ref), (kind, f, ref, dg)
