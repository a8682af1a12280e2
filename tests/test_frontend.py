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


def test_oracle_building_blocks_against_independent_implementations():
    """The HTSAT pin goes through a private method of transformers' ClapFeatureExtractor and VGGish has no third-party
    implementation to run here, so the two building blocks every front end is made of are ALSO checked against
    independent public code: the STFT against ``torch.stft`` and the mel banks against
    ``transformers.audio_utils.mel_filter_bank`` (Slaney scale + norm for HTSAT / Whisper, HTK scale for VGGish)."""
    import torch
    from transformers.audio_utils import mel_filter_bank
    # --- HTSAT: torchlibrosa Spectrogram(n_fft=1024, hop=480, hann, center, reflect, power=2) == torch.stft with those arguments
    x = R.audio_clip(330, 48000, 48000)
    st = torch.stft(torch.from_numpy(x.astype(np.float64)), n_fft=1024, hop_length=480, win_length=1024,
                    window=torch.hann_window(1024, periodic=True, dtype=torch.float64), center=True, pad_mode="reflect",
                    return_complex=True)
    power_t = (st.abs() ** 2).numpy().T                                   # [frames, 513]
    power_o = L._centered_power(x.astype(np.float64), 1024, 480)
    assert power_t.shape == power_o.shape
    np.testing.assert_allclose(power_o, power_t, rtol=1e-9, atol=1e-12 * power_t.max())
    fb_t = mel_filter_bank(num_frequency_bins=513, num_mel_filters=64, min_frequency=50.0, max_frequency=14000.0,
                           sampling_rate=48000, norm="slaney", mel_scale="slaney")
    fb_o = L.mel_slaney(513, 64, 48000.0, 50.0, 14000.0)
    np.testing.assert_allclose(fb_o, fb_t, rtol=1e-9, atol=1e-15)
    logmel_t = 10.0 * np.log10(np.maximum(power_t @ fb_t, 1e-10))
    np.testing.assert_allclose(L.htsat_logmel(x), logmel_t, rtol=0, atol=1e-8)
    # --- Whisper: same bank family (201 bins, 80 mels, 0-8 kHz), same centred STFT at 400/160
    np.testing.assert_allclose(L.mel_slaney(201, 80, 16000.0, 0.0, 8000.0),
                               mel_filter_bank(num_frequency_bins=201, num_mel_filters=80, min_frequency=0.0, max_frequency=8000.0,
                                               sampling_rate=16000, norm="slaney", mel_scale="slaney"), rtol=1e-9, atol=1e-15)
    # --- VGGish: un-centred magnitude STFT (periodic Hann 400, hop 160, FFT 512) and HTK triangles 125-7500 Hz
    y = R.audio_clip(331, 16000, 16000).astype(np.float64)
    win = torch.zeros(512, dtype=torch.float64)
    win[:400] = torch.hann_window(400, periodic=True, dtype=torch.float64)          # the 400-sample window, zero-padded to the FFT
    st = torch.stft(torch.from_numpy(y), n_fft=512, hop_length=160, win_length=512, window=win, center=False, return_complex=True)
    mag_t = st.abs().numpy().T
    fr = L._frames(y, 400, 160) * L._periodic_hann(400)
    mag_o = np.abs(np.fft.rfft(fr, 512, axis=1))
    n = min(len(mag_t), len(mag_o))                                       # torch needs 512 samples for its last frame, the oracle 400
    assert n >= len(mag_o) - 1
    np.testing.assert_allclose(mag_o[:n], mag_t[:n], rtol=1e-9, atol=1e-12 * mag_t.max())
    htk_t = mel_filter_bank(num_frequency_bins=257, num_mel_filters=64, min_frequency=125.0, max_frequency=7500.0,
                            sampling_rate=16000, norm=None, mel_scale="htk", triangularize_in_mel_space=True)   # AudioSet's triangles are linear in mel
    htk_o = L.mel_htk_vggish()
    np.testing.assert_allclose(htk_o[1:], htk_t[1:], rtol=1e-9, atol=1e-12)           # AudioSet additionally zeroes the DC bin
    assert (htk_o[0] == 0).all()


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
