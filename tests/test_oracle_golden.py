"""Pin the CPU oracle (oracle/fad_oracle.py) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by running microsoft/fadtk's own functions
(tests/golden/make_golden.py).  These tests run on CPU (-m "not gpu").
"""
import json
import logging
from pathlib import Path

import numpy as np
import pytest

import recipes as R
from oracle import fad_oracle as O

logging.getLogger("fad_oracle").setLevel(logging.CRITICAL)

RTOL = 1e-9      # oracle vs reference scalars (same LAPACK here; differs only by BLAS build)


def test_g1_embd_statistics(golden, golden_dir):
    z = np.load(golden_dir / "g1_stats.npz")
    for c in golden["g1"]:
        x = R.normal_rows(c["seed"], c["n"], c["d"], c["scale"], c["shift"], dtype=np.dtype(c["dtype"]))
        assert R.checksum(x) == pytest.approx(c["in_checksum"], rel=1e-12)
        mu, cov = O.embd_statistics(x)
        assert str(mu.dtype) == c["mu_dtype"] and str(cov.dtype) == c["cov_dtype"]
        np.testing.assert_array_equal(mu, z[f"mu{c['id']}"])          # bit-exact incl. fp16 rounding
        np.testing.assert_allclose(cov, z[f"cov{c['id']}"], rtol=1e-12, atol=1e-14)


def test_g1_requires_two_rows():
    with pytest.raises(AssertionError):
        O.embd_statistics(np.zeros((1, 4), np.float16))


def _fd(a, b):
    m1, c1 = O.embd_statistics(a)
    m2, c2 = O.embd_statistics(b)
    return O.frechet_distance(m1, c1, m2, c2, run_sqrtm=False), np.trace(c1), np.trace(c2)


@pytest.mark.parametrize("case", ["c1_iid", "c1_iid_f32", "c1_iid_f64", "shifted", "shifted_long", "identical"])
def test_g2_frechet_pairs(golden, case):
    g = golden["g2"][case]
    if case == "c1_iid" or case == "identical":
        a, b = R.c1_pair()
        if case == "identical":
            b = a
    elif case == "c1_iid_f32":
        a, b = R.c1_pair(np.float32)
    elif case == "c1_iid_f64":
        a, b = R.c1_pair(np.float64)
    elif case == "shifted_long":             # 60000 rows: the reference's float16 means come out of a float32 running sum (fad.py:48)
        a, b = R.shifted_pair(n=60000)
        assert R.checksum(a) == pytest.approx(g["in_checksum"][0], rel=1e-12)
    else:
        a, b = R.shifted_pair()
    fad, t1, t2 = _fd(a, b)
    assert t1 == pytest.approx(g["tr1"], rel=1e-12)
    assert t2 == pytest.approx(g["tr2"], rel=1e-12)
    if case == "identical":
        assert abs(fad) < 1e-9 and abs(g["fad"]) < 1e-9
    else:
        assert fad == pytest.approx(g["fad"], rel=RTOL)


@pytest.mark.parametrize("d", [64, 512])
def test_g2_decaying_spectrum(golden, d):
    x1 = R.decaying_rows(30, 4 * d, d, basis_seed=40)
    x2 = R.decaying_rows(31, 4 * d, d, basis_seed=40, gain=1.1)
    x3 = R.decaying_rows(32, 4 * d, d, basis_seed=41)
    # cond ~ 1e6..1e8: LAPACK eig noise itself is ~1e-9 relative here
    assert _fd(x1, x2)[0] == pytest.approx(golden["g2"][f"decay_same_basis_d{d}"]["fad"], rel=1e-6)
    assert _fd(x1, x3)[0] == pytest.approx(golden["g2"][f"decay_diff_basis_d{d}"]["fad"], rel=1e-6)


@pytest.mark.parametrize("d,rows", [(128, 2), (128, 10), (128, 50), (768, 2)])
def test_g2_rank_deficient(golden, d, rows):
    g = golden["g2"][f"rankdef_d{d}_n{rows}"]
    mu_b, cov_b = R.baseline_stats(50 + d, 4 * d, d)
    s = R.songs(60 + rows, 1, rows, d)[0]
    assert R.checksum(s) == pytest.approx(g["in_checksum"][1], rel=1e-12)
    mu_s, cov_s = O.embd_statistics(s)
    fad = O.frechet_distance(mu_b, cov_b, mu_s, cov_s, run_sqrtm=False)
    assert fad == pytest.approx(g["fad"], rel=1e-7)     # eig noise of a singular product


def _g10_case(key):
    """(mu1, cov1, mu2, cov2) of a G10 fixture, rebuilt from the seeded recipe."""
    if key.startswith("short_eval"):
        p = float(key.rsplit("_p", 1)[1])
        base = R.decaying_rows(50, 4096, 256, 52, power=p)
        song = R.decaying_rows(51, 150, 256, 53, power=p)
        return (*O.embd_statistics(base), *O.embd_statistics(song))
    p = float(key.rsplit("_p", 1)[1])
    a = R.decaying_rows(30, 4096, 128, 32, power=p)
    b = R.decaying_rows(31, 4096, 128, 33, power=p, gain=1.05)
    m1, c1 = O.embd_statistics(a)
    m2, c2 = O.embd_statistics(b)
    if key.startswith("f32cov"):
        c1, c2 = c1.astype(np.float32).astype(np.float64), c2.astype(np.float32).astype(np.float64)
    return m1, c1, m2, c2


@pytest.mark.parametrize("key", ["short_eval_d256_n150_p2", "short_eval_d256_n150_p3", "short_eval_d256_n150_p4",
                                 "fullrank_d128_p3", "fullrank_d128_p4", "f32cov_d128_p3", "f32cov_d128_p4"])
def test_g10_hard_spectra(golden, key):
    """Near-singular products (eval set of N < D frames, power-law spectra k^-2..k^-4, covariances that went through
    float32): the oracle must reproduce what the reference's eig route returns for them."""
    fad = O.frechet_distance(*_g10_case(key), run_sqrtm=False)
    assert fad == pytest.approx(golden["g10"][key]["fad"], rel=1e-7)


def test_g2_shape_asserts_and_both_roots():
    a, b = R.c1_pair()
    m1, c1 = O.embd_statistics(a[:, :16])
    m2, c2 = O.embd_statistics(b[:, :8])
    with pytest.raises(AssertionError):
        O.frechet_distance(m1, c1, m2, c2)
    m2, c2 = O.embd_statistics(b[:, :16])
    p = O.frechet_parts(m1, c1, m2, c2, run_sqrtm=True)
    assert abs(p.tr_sqrt_eig - p.tr_sqrt_schur) < 1e-8 and not p.used_eps
    assert O.frechet_distance(m1, c1, m2, c2).dtype == np.float64
    assert p.mean_term.dtype == np.float16          # Q1: fp16 means -> fp16 dot


def test_g3_online_statistics(golden, golden_dir):
    g = golden["g3"]
    z = np.load(golden_dir / "g3_online.npz")
    blocks = R.ragged_files(g["seed"], g["n_files"], g["d"])
    assert [b.shape[0] for b in blocks] == g["sizes"]
    mu, cov = O.statistics_online(blocks)
    assert str(mu.dtype) == g["mu_dtype"] and str(cov.dtype) == g["cov_dtype"]
    np.testing.assert_allclose(mu, z["mu"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(cov, z["cov"], rtol=1e-11, atol=1e-13)
    # Q5: a one-row file poisons the covariance
    mu_n, cov_n = O.statistics_online(blocks[:3] + [blocks[0][:1]])
    assert np.isnan(cov_n).all() == g["cov_nan_all"]
    np.testing.assert_allclose(mu_n, z["mu_nan"], rtol=1e-13)


def test_g3_g4_shifted_fixtures_pin_numpys_running_sum_means(golden, golden_dir):
    """Round 5 fixtures (reference-generated): long float16 files / songs with |mu| / sigma ~ 7, where np.mean's float32 running sum
    (utils.py:16, fad.py:377 -> :48) gives float16 means that differ from the rounded exact means.  The oracle uses np.mean itself and
    must reproduce the reference; the rounded-exact-mean variant must NOT (else the fixture pins nothing)."""
    g = golden["g3_shifted"]
    z = np.load(golden_dir / "g3_online.npz")
    blocks = R.shifted_files(g["seed"], g["n_files"], g["d"], min_rows=g["min_rows"], max_rows=g["max_rows"])
    assert [b.shape[0] for b in blocks] == g["sizes"]
    mu, cov = O.statistics_online(blocks)
    np.testing.assert_allclose(mu, z["mu_shifted"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(cov, z["cov_shifted"], rtol=1e-10, atol=1e-12)
    exact16 = [b.astype(np.float64).mean(axis=0).astype(np.float32).astype(np.float16) for b in blocks]
    numpy16 = [b.mean(axis=0) for b in blocks]
    assert sum(int((a != b).sum()) for a, b in zip(exact16, numpy16)) > 0
    g4 = golden["g4_shifted"]
    srows = R.shifted_files(g4["songs_seed"], len(g4["names"]), g4["d"], min_rows=g4["min_rows"], max_rows=g4["max_rows"])
    assert [r.shape[0] for r in srows] == g4["rows"]
    rngb = np.random.default_rng(g4["base_seed"])
    xb = rngb.standard_normal((g4["base_n"], g4["d"])) * (1.0 + 0.3 * rngb.random(g4["d"])) + 7.0
    mu_b, cov_b = xb.mean(axis=0), np.cov(xb, rowvar=False)
    scores = O.individual_scores(mu_b, cov_b, srows, run_sqrtm=False)
    want = {ln.rsplit(",", 1)[0].rsplit("/", 1)[1]: float(ln.rsplit(",", 1)[1]) for ln in g4["csv"].split("\n")}
    for nm, sc in zip(g4["names"], scores):
        assert abs(sc - want[nm]) <= 1e-9 * abs(want[nm]), (nm, sc, want[nm])


def test_g4_individual_csv(golden):
    g = golden["g4"]
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], g["d"])
    song_rows = R.songs(g["songs_seed"], len(g["names"]), g["rows"], g["d"])
    by_name = dict(zip(g["names"], song_rows))
    # the reference iterates Path.glob order; scores are sorted afterwards so any order works
    paths = ["{ROOT}/evalset/" + n for n in g["names"]]
    scores = O.individual_scores(mu_b, cov_b, [by_name[n] for n in g["names"]], run_sqrtm=False)
    assert scores[g["names"].index("short.wav")] is None            # 1 frame -> dropped
    text = O.individual_csv_text(paths, scores)
    got = [ln.rsplit(",", 1) for ln in text.split("\n")]
    want = [ln.rsplit(",", 1) for ln in g["csv"].split("\n")]
    assert [a for a, _ in got] == [a for a, _ in want]
    np.testing.assert_allclose([float(b) for _, b in got], [float(b) for _, b in want], rtol=1e-8)
    assert "be_ta.wav" in text and "be,ta" not in text


def test_g6_score_inf(golden):
    g = golden["g6"]
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], g["d"])
    rows = R.normal_rows(g["rows_seed"], g["n"], g["d"], 1.1, 0.05)
    np.random.seed(0)
    res = O.score_inf(mu_b, cov_b, rows, run_sqrtm=False)
    assert [p[0] for p in res.points] == [p[0] for p in g["points"]]
    np.testing.assert_allclose([p[1] for p in res.points], [p[1] for p in g["points"]], rtol=1e-8)
    assert res.score == pytest.approx(g["score"], rel=1e-7)
    assert res.slope == pytest.approx(g["slope"], rel=1e-6)
    assert res.r2 == pytest.approx(g["r2"], rel=1e-6)


def test_g7_config3_scalar(golden):
    """BASELINE config 3 at full size (N = 100 000, D = 512, float16; seeds 10 / 11): the ORACLE against the scalar the reference itself
    returned for this recipe (round 6: rounds 1-5 held only the HIP path against it).  ~4 s of BLAS."""
    g = golden["g7"]
    a, b = R.c3_pair()
    assert R.checksum(a) == pytest.approx(g["in_checksum"][0], rel=1e-12) and R.checksum(b) == pytest.approx(g["in_checksum"][1], rel=1e-12)
    m1, c1 = O.embd_statistics(a)
    m2, c2 = O.embd_statistics(b)
    assert m1.dtype == np.float16 and str(m1.dtype) == g["mean_term_dtype"]
    assert np.trace(c1) == pytest.approx(g["tr1"], rel=1e-12) and np.trace(c2) == pytest.approx(g["tr2"], rel=1e-12)
    d = m1 - m2
    assert float(d.dot(d)) == g["mean_term"]                    # a float16 scalar in the reference: bit for bit
    fad = O.frechet_distance(m1, c1, m2, c2, run_sqrtm=False)
    assert abs(fad - g["fad"]) / g["fad"] < 1e-10


def test_g8_two_row_songs_subset(golden):
    g = golden["g8"]
    d = g["d"]
    mu_b, cov_b = R.baseline_stats(g["base_seed"], g["base_n"], d)
    sg = R.songs(g["songs_seed"], g["n_songs"], g["rows"], d)
    for k in (0, 17):                                   # ~1 s each at D=768 on CPU
        mu_s, cov_s = O.embd_statistics(sg[k])
        assert O.frechet_distance(mu_b, cov_b, mu_s, cov_s, run_sqrtm=False) == \
            pytest.approx(g["scores"][k], rel=1e-7)
    m = g["multi"]
    mu_b, cov_b = R.baseline_stats(m["base_seed"], m["base_n"], m["d"])
    sg = R.songs(m["songs_seed"], m["n_songs"], m["rows"], m["d"])
    got = O.individual_scores(mu_b, cov_b, sg, run_sqrtm=False)
    np.testing.assert_allclose(got, m["scores"], rtol=1e-7)
