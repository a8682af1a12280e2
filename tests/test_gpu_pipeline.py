"""End to end on the GPU: synthetic audio -> HIP front end + PyTorch-ROCm forward -> fp16 embedding cache
-> HIP moments / Frechet -> score, checked against the oracle run on the very same cached embeddings
(BASELINE config 2 shape, scaled down; weights are seeded random: no checkpoints exist offline)."""
import logging

import numpy as np
import pytest

import recipes as R
from oracle import fad_oracle as O

pytestmark = pytest.mark.gpu
logging.getLogger("fad_oracle").setLevel(logging.CRITICAL)


def _make_set(root, n_files, seconds, sr, seed0, gain=1.0):
    from fadtk_amd import audio
    root.mkdir(parents=True)
    for i in range(n_files):
        audio.write_pcm16(root / f"clip{i:03d}.wav", gain * R.audio_clip(seed0 + i, int(seconds * sr), sr), sr)
    return root


def test_vggish_directory_score_matches_oracle(tmp_path, monkeypatch):
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    import fadtk_amd
    from fadtk_amd.fad_batch import cache_embedding_files
    from fadtk_amd.model_loader import VGGishModel
    base = _make_set(tmp_path / "base", 12, 10.0, 16000, 500)
    evl = _make_set(tmp_path / "eval", 9, 10.0, 16000, 600, gain=0.7)
    ml = VGGishModel()
    for d in (base, evl):
        cache_embedding_files(d, ml, workers=4)
    embs = sorted((base / "embeddings" / "vggish").glob("*.npy"))
    assert len(embs) == 12
    e0 = np.load(embs[0])
    assert e0.shape == (10, 128) and e0.dtype == np.float16          # 10 s -> 10 examples of 0.96 s
    assert (base / "convert" / "16000" / "clip000.wav").exists()     # normalised-audio cache
    cache_embedding_files(base, ml, workers=4)                        # second call: everything cached

    fad = fadtk_amd.FrechetAudioDistance(ml, load_model=False)
    score = fad.score(base, evl)
    blocks_b = [np.load(p) for p in (base / "embeddings" / "vggish").glob("*.npy")]
    blocks_e = [np.load(p) for p in (evl / "embeddings" / "vggish").glob("*.npy")]
    want = O.frechet_distance(*O.statistics_online(blocks_b), *O.statistics_online(blocks_e), run_sqrtm=False)
    assert abs(score - want) / abs(want) < 1e-4
    assert (base / "stats" / "vggish" / "mu.npy").exists()

    out = fad.score_individual(base, evl, tmp_path / "indiv.csv")
    lines = out.read_text().split("\n")
    assert len(lines) == 9
    mu_b, cov_b = fad.load_stats(base)
    by_name = {ln.rsplit(",", 1)[0]: float(ln.rsplit(",", 1)[1]) for ln in lines}
    for p in sorted(evl.glob("*.wav"))[:3]:
        e = np.load(evl / "embeddings" / "vggish" / (p.stem + ".npy"))
        ref = O.frechet_distance(mu_b, cov_b, *O.embd_statistics(e), run_sqrtm=False)
        assert abs(by_name[str(p)] - ref) / abs(ref) < 1e-4


def test_fused_embed_and_accumulate_matches_cache_route(tmp_path, monkeypatch):
    """Encodec-shaped flow (config 4, scaled down): frames go straight from the model output into the HBM moments."""
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    import fadtk_amd
    from fadtk_amd.fad_batch import embed_and_accumulate
    from fadtk_amd.model_loader import EncodecEmbModel
    root = _make_set(tmp_path / "set", 6, 2.0, 24000, 800)
    ml = EncodecEmbModel("24k")
    mu, cov = embed_and_accumulate(root, ml, workers=2)
    blocks = [np.load(p) for p in sorted((root / "embeddings" / ml.name).glob("*.npy"))]
    assert len(blocks) == 6 and blocks[0].shape == (150, 128) and blocks[0].dtype == np.float16
    mu_o, cov_o = O.statistics_online(blocks)                 # the reference's online path, float16 file means included
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=1e-9 * np.abs(mu_o).max() + 1e-12)
    np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
    import shutil
    shutil.rmtree(root / "stats")
    mu_p, cov_p = embed_and_accumulate(root, ml, workers=2, compat=False)      # plain raw moments, from the cache now
    mu_i, cov_i = O.embd_statistics(np.concatenate(blocks).astype(np.float64))
    np.testing.assert_allclose(mu_p, mu_i, rtol=0, atol=1e-6 * np.abs(mu_i).max() + 1e-9)
    np.testing.assert_allclose(cov_p, cov_i, rtol=0, atol=2e-6 * np.abs(cov_i).max())
    mu, cov = embed_and_accumulate(root, ml, workers=2)
    fad = fadtk_amd.FrechetAudioDistance(ml, load_model=False)
    mu_c, cov_c = fad.load_stats(root)                       # served from the stats cache the fused pass wrote
    assert np.array_equal(mu_c, mu) and np.array_equal(cov_c, cov)


@pytest.mark.parametrize("which", ["whisper-tiny", "encodec-emb", "clap-laion-audio", "hubert-base", "MERT-v1-95M-6"])
def test_loader_shapes_on_gpu(which, monkeypatch, tmp_path):
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    from fadtk_amd import audio
    from fadtk_amd.model_loader import get_all_models
    ml = {m.name: m for m in get_all_models()}[which]
    ml.load_model()
    secs = 3
    audio.write_pcm16(tmp_path / "x.wav", R.audio_clip(700, secs * ml.sr, ml.sr), ml.sr)
    emb = ml.get_embedding(ml.load_wav(tmp_path / "x.wav"))
    assert emb.dtype == np.float16 and emb.ndim == 2 and emb.shape[1] == ml.num_features and np.isfinite(emb).all()
    expect = {"whisper-tiny": 2, "encodec-emb": 75 * secs, "clap-laion-audio": secs, "hubert-base": 50 * secs - 1,
              "MERT-v1-95M-6": 75 * secs - 1}[which]
    assert emb.shape[0] == expect


def test_cli_end_to_end(tmp_path, monkeypatch):
    """`python -m fadtk <model> <baseline> <eval> <csv>` and `--indiv`, as a user would run them."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    base = _make_set(tmp_path / "base", 5, 3.0, 16000, 900)
    evl = _make_set(tmp_path / "eval", 4, 3.0, 16000, 950, gain=0.8)
    env = dict(os.environ, FADTK_AMD_RANDOM_WEIGHTS="1", PYTHONPATH=str(root))
    csv = tmp_path / "out" / "scores.csv"
    r = subprocess.run([sys.executable, "-m", "fadtk", "vggish", str(base), str(evl), str(csv), "-w", "2", "--fused-stats"],
                       capture_output=True, text=True, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = csv.read_text().strip().split("\n")
    assert lines[0] == "model,baseline,eval,score,inf_r2,time" and len(lines) == 2
    cols = lines[1].split(",")
    assert cols[0] == "vggish" and cols[1] == str(base) and cols[4] == "None" and np.isfinite(float(cols[3]))
    assert "The FAD vggish score between" in r.stderr
    blocks_b = [np.load(p) for p in (base / "embeddings" / "vggish").glob("*.npy")]
    blocks_e = [np.load(p) for p in (evl / "embeddings" / "vggish").glob("*.npy")]
    # the reference CLI takes dataset statistics from its online path (float64 mu); --fused-stats skips only that
    # path's per-file float16 rounding of the means
    want = O.frechet_distance(*O.statistics_online(blocks_b), *O.statistics_online(blocks_e), run_sqrtm=False)
    assert abs(float(cols[3]) - want) / abs(want) < 1e-4
    r = subprocess.run([sys.executable, "-m", "fadtk", "vggish", str(base), str(evl), str(tmp_path / "indiv.csv"), "--indiv"],
                       capture_output=True, text=True, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = (tmp_path / "indiv.csv").read_text().strip().split("\n")
    assert len(rows) == 4 and all(str(evl) in ln for ln in rows)
    vals = [abs(float(ln.rsplit(",", 1)[1])) for ln in rows]
    assert vals == sorted(vals)


@pytest.mark.parametrize("orig_sr,new_sr", [(48000, 16000), (44100, 16000), (8000, 16000), (22050, 24000), (44100, 48000), (32000, 48000)])
def test_resampler_kernel_matches_oracle(orig_sr, new_sr):
    """fad_resample_kaiser (fadtk's torchaudio Kaiser-sinc parameters, fad.py:151-159) against the float64 restatement:
    fp32 accumulation over <= 815 taps, so 2e-6 absolute on a signal of amplitude ~0.5; after the 16-bit round trip
    the samples are identical except where the unquantised value sits within that error of a rounding boundary."""
    import torch
    from oracle import audio_oracle as AO
    from fadtk_amd import hip
    x = R.audio_clip(31, orig_sr // 2 + 123, orig_sr)
    want = AO.resample_kaiser(x, orig_sr, new_sr)
    got = hip.resample_kaiser(x, orig_sr, new_sr)
    assert got.dtype == np.float32 and got.shape == want.shape
    assert np.abs(got - want).max() < 2e-6
    got_dev = hip.resample_kaiser(torch.from_numpy(x).cuda(), orig_sr, new_sr)          # device in, device out
    assert got_dev.is_cuda and np.array_equal(got_dev.cpu().numpy(), got)
    q = hip.resample_kaiser(x, orig_sr, new_sr, quantize_pcm16=True)
    qw = AO.pcm16_roundtrip(want)
    lsb = np.abs(q - qw) * 32768
    assert lsb.max() <= 1.0 and (lsb > 0).mean() < 1e-2       # 2e-6 * 32768 = 0.07 LSB around every .5 boundary
    near_half = np.abs((want * 32768) % 1.0 - 0.5) < 1e-1
    assert not np.any((lsb > 0) & ~near_half)


def test_resampler_edges_and_audio_normalisation(tmp_path):
    from oracle import audio_oracle as AO
    from fadtk_amd import audio, hip
    assert hip.resample_kaiser(np.zeros(0, np.float32), 44100, 16000).shape == (0,)
    short = R.audio_clip(5, 37, 44100)                                   # shorter than the filter half-width
    np.testing.assert_allclose(hip.resample_kaiser(short, 44100, 16000), AO.resample_kaiser(short, 44100, 16000), atol=2e-6)
    same = R.audio_clip(6, 1000, 16000)
    np.testing.assert_array_equal(hip.resample_kaiser(same, 16000, 16000), same)        # torchaudio returns the input as is
    np.testing.assert_array_equal(hip.resample_kaiser(same, 16000, 16000, quantize_pcm16=True), AO.pcm16_roundtrip(same).astype(np.float32))
    with pytest.raises(Exception):
        hip.resample_kaiser(same, 44101, 16000)                          # coprime rates: filter table far too large
    # load_audio's normalisation step (fad.py:148-160): decode, mono mix, resample, PCM16 cache file
    x = R.audio_clip(400, 24000, 48000)
    audio.write_pcm16(tmp_path / "a.wav", x, 48000)
    audio.convert_to_model_rate(tmp_path / "a.wav", tmp_path / "convert" / "16000" / "a.wav", 16000)
    pcm, fs = audio.read_pcm16(tmp_path / "convert" / "16000" / "a.wav")
    assert fs == 16000 and pcm.shape == (8000,)
    src = audio.read_pcm16(tmp_path / "a.wav")[0] / 32768.0
    want = np.rint(AO.resample_kaiser(src.astype(np.float32), 48000, 16000) * 32768)
    assert np.abs(pcm - want).max() <= 1
