"""Log-mel front ends: oracle vs third-party golden vectors (CPU) and HIP kernels vs oracle/golden (GPU)."""
import json

import numpy as np
import pytest

import recipes as R
from oracle import logmel_oracle as L


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(golden_dir / "g9_frontend.npz"), json.loads((golden_dir / "g9_frontend.json").read_text())


def test_oracle_matches_transformers_extractors(g9):
    z, meta = g9
    for c in meta["cases"]:
        x = R.audio_clip(c["seed"], c["n"], c["sr"])
        full = L.whisper_features(x) if c["name"].startswith("whisper") else L.htsat_logmel(x)
        assert list(full.shape) == c["shape"]
        dec = full[:, ::c["stride"]] if c["name"].startswith("whisper") else full[::c["stride"]]
        np.testing.assert_allclose(dec, z[c["name"]], rtol=0, atol=2e-5)
        assert full.sum() == pytest.approx(c["sum"], rel=1e-6)


def test_oracle_vggish_shapes_and_filters():
    assert [L.vggish_num_examples(n) for n in (0, 399, 15599, 15600, 15760, 160000)] == [0, 0, 0, 1, 1, 10]
    w = L.mel_htk_vggish()
    assert w.shape == (257, 64) and (w[0] == 0).all() and w.min() >= 0 and w.max() <= 1.0
    ex = L.vggish_examples(R.audio_clip(320, 16000 * 3, 16000))
    assert ex.shape == (3, 96, 64) and np.isfinite(ex).all() and ex.min() >= np.log(0.01) - 1e-12


@pytest.mark.gpu
def test_hip_whisper_and_htsat_match_golden_and_oracle(g9):
    from fadtk_amd import hip
    z, meta = g9
    wh = [c for c in meta["cases"] if c["name"].startswith("whisper")]
    clips = [R.audio_clip(c["seed"], c["n"], c["sr"]) for c in wh]
    out = hip.logmel_whisper(clips)                                  # one batched call, ragged clips
    assert out.shape == (len(wh), 80, 3000) and out.dtype == np.float32
    for k, c in enumerate(wh):
        np.testing.assert_allclose(out[k][:, ::c["stride"]], z[c["name"]], rtol=0, atol=2e-4)
        np.testing.assert_allclose(out[k], L.whisper_features(clips[k]), rtol=0, atol=2e-4)
    out128 = hip.logmel_whisper(clips[:1], n_mels=128)
    np.testing.assert_allclose(out128[0], L.whisper_features(clips[0], 128), rtol=0, atol=2e-4)

    ht = [c for c in meta["cases"] if c["name"].startswith("htsat")]
    clips = [R.audio_clip(c["seed"], c["n"], c["sr"]) for c in ht]
    out = hip.logmel_htsat(clips)
    assert out.shape == (2, 1001, 64)
    for k, c in enumerate(ht):
        np.testing.assert_allclose(out[k][::c["stride"]], z[c["name"]], rtol=0, atol=2e-3)     # dB scale
        np.testing.assert_allclose(out[k], L.htsat_logmel(clips[k]), rtol=0, atol=2e-3)
    with pytest.raises(AssertionError):
        hip.logmel_htsat([clips[0], clips[1][:1000]])               # clips must share one length


@pytest.mark.gpu
def test_hip_vggish_matches_oracle():
    import torch
    from fadtk_amd import hip
    lens = [16000 * 10, 15599, 15600, 16000 * 3 + 77, 0, 400]
    clips = [R.audio_clip(330 + i, n, 16000) for i, n in enumerate(lens)]
    ex, off = hip.logmel_vggish(clips)
    want = [L.vggish_examples(c) for c in clips]
    assert list(np.diff(off)) == [len(w) for w in want] == [10, 0, 1, 3, 0, 0]
    np.testing.assert_allclose(ex, np.concatenate(want), rtol=0, atol=3e-4)
    ex_dev, off_dev = hip.logmel_vggish([torch.from_numpy(c).cuda() for c in clips])       # device-resident route
    assert ex_dev.is_cuda and np.array_equal(off_dev, off)
    np.testing.assert_allclose(ex_dev.cpu().numpy(), ex, rtol=0, atol=1e-6)
