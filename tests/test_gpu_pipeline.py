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


@pytest.mark.parametrize("which", ["whisper-tiny", "encodec-emb", "clap-laion-audio", "hubert-base", "MERT-v1-95M-6", "clap-2023"])
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
              "MERT-v1-95M-6": 75 * secs - 1, "clap-2023": secs}[which]            # clap-2023: one 7 s window per second of audio
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


# --------------------------------------------------------------------------------- reference surface not exercised elsewhere
def _toy_fad(name="toy-model", d=32):
    import types
    import fadtk_amd
    ml = types.SimpleNamespace(name=name, sr=16000, num_features=d, load_model=lambda: None)
    return fadtk_amd.FrechetAudioDistance(ml, audio_load_worker=2, load_model=False)


def test_load_embeddings_max_count_and_concat_false(tmp_path):
    """fad.py:211-243: load_embeddings reads every cached .npy of a directory; max_count stops once MORE than max_count
    frames are in; concat=False returns (list of matrices, files)."""
    fad = _toy_fad()
    root = tmp_path / "set"
    (root / "embeddings" / "toy-model").mkdir(parents=True)
    blocks = R.songs(300, 7, [5, 9, 2, 33, 12, 7, 4], 32)
    names = [f"s{i}.wav" for i in range(7)]
    for nm, blk in zip(names, blocks):
        (root / nm).write_bytes(b"")
        np.save(root / "embeddings" / "toy-model" / (Path(nm).stem + ".npy"), blk)
    files = sorted(root.glob("*.wav"))
    allrows = fad._load_embeddings(files, concat=True)
    assert allrows.shape == (sum(b.shape[0] for b in blocks), 32) and allrows.dtype == np.float16
    np.testing.assert_array_equal(allrows, np.concatenate(blocks))
    got = fad.load_embeddings(root, concat=True)                        # glob order of the directory, same rows overall
    assert got.shape == allrows.shape and np.isclose(got.astype(np.float64).sum(), allrows.astype(np.float64).sum())
    part = fad._load_embeddings(files, max_count=15, concat=True)       # 5 + 9 = 14 <= 15, + 2 = 16 > 15 -> stops after 3 files
    assert part.shape[0] == 16
    lst, fs = fad._load_embeddings(files, max_count=15, concat=False)
    assert [b.shape[0] for b in lst] == [5, 9, 2] and fs == files
    with pytest.raises(ValueError):
        fad._load_embeddings([], concat=True)


from pathlib import Path     # noqa: E402


def test_package_stats_round_trip_into_load_stats(tmp_path, monkeypatch):
    """`python -m fadtk.package <dir> <out.npz>` (package.py:34-42) writes {model}.mu / {model}.cov that load_stats
    (fad.py:262-268) reads back; the statistics equal the directory's own."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    base = _make_set(tmp_path / "base", 4, 3.0, 16000, 1200)
    env = dict(os.environ, FADTK_AMD_RANDOM_WEIGHTS="1", PYTHONPATH=str(root))
    out = tmp_path / "packed.npz"
    r = subprocess.run([sys.executable, "-m", "fadtk.package", str(base), str(out), "-m", "vggish", "-w", "2"],
                       capture_output=True, text=True, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    with np.load(out) as z:
        assert sorted(z.files) == ["vggish.cov", "vggish.mu"]
        mu_p, cov_p = z["vggish.mu"], z["vggish.cov"]
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    import fadtk_amd
    from fadtk_amd.model_loader import VGGishModel
    fad = fadtk_amd.FrechetAudioDistance(VGGishModel(), load_model=False)
    mu_f, cov_f = fad.load_stats(str(out))                              # an .npz file
    assert np.array_equal(mu_f, mu_p) and np.array_equal(cov_f, cov_p)
    mu_d, cov_d = fad.load_stats(base)                                  # the directory (its stats cache was written by package)
    assert np.array_equal(mu_d, mu_p) and np.array_equal(cov_d, cov_p)
    blocks = [np.load(p) for p in (base / "embeddings" / "vggish").glob("*.npy")]
    mu_o, cov_o = O.statistics_online(blocks)
    np.testing.assert_allclose(cov_p, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
    with pytest.raises(ValueError):                                     # missing key (fad.py:265)
        _toy_fad("other-model").load_stats(str(out))


def test_cli_inf_end_to_end(tmp_path):
    """`python -m fadtk <model> <baseline> <eval> <csv> --inf` (__main__.py:45-49): FAD-inf over the eval directory,
    score and r2 appended to the CSV; the extrapolation equals the oracle's on the same cached embeddings and seed."""
    import os
    import subprocess
    import sys
    root = Path(__file__).resolve().parent.parent
    base = _make_set(tmp_path / "base", 12, 6.0, 16000, 1300)
    evl = _make_set(tmp_path / "eval", 30, 10.0, 16000, 1400, gain=0.8)       # 30 x 10 examples = 300 eval frames
    env = dict(os.environ, FADTK_AMD_RANDOM_WEIGHTS="1", PYTHONPATH=str(root))
    csv = tmp_path / "inf.csv"
    code = ("import sys, numpy as np; np.random.seed(0); sys.argv = ['fadtk', 'vggish', %r, %r, %r, '--inf', '-w', '2'];"
            "import fadtk_amd.cli as c; c.score_main()" % (str(base), str(evl), str(csv)))
    # score_inf's default min_n is 500 > 300 frames: numpy then draws MORE than the set holds (with replacement), like the reference
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=tmp_path)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "FAD-inf Information:" in r.stdout
    cols = csv.read_text().strip().split("\n")[1].split(",")
    assert cols[0] == "vggish" and np.isfinite(float(cols[3])) and cols[4] != "None" and 0.0 <= float(cols[4]) <= 1.0 + 1e-9
    # oracle on the same embeddings, same RNG stream, same file order as the CLI's glob
    mu_b, cov_b = O.statistics_online([np.load(p) for p in (base / "embeddings" / "vggish").glob("*.npy")])
    files = [evl / "embeddings" / "vggish" / (p.stem + ".npy") for p in evl.glob("*.*")]
    embeds = np.concatenate([np.load(f) for f in files], axis=0)
    np.random.seed(0)
    want = O.score_inf(mu_b, cov_b, embeds)
    assert abs(float(cols[3]) - want.score) / abs(want.score) < 5e-4
    assert abs(float(cols[4]) - want.r2) < 1e-3


def test_encodec_48k_segments_of_one_second(monkeypatch, tmp_path):
    """model_loader.py:139-152: the 48 kHz Encodec embeds 1-second stereo segments independently and concatenates the
    frames (150 per full second; the tail segment gives ceil(samples / 320) frames)."""
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    import torch
    from fadtk_amd import audio
    from fadtk_amd.model_loader import EncodecEmbModel
    ml = EncodecEmbModel("48k")
    ml.load_model()
    sr = 48000
    audio.write_pcm16(tmp_path / "x.wav", R.audio_clip(1500, int(2.5 * sr), sr), sr)
    wav = ml.load_wav(tmp_path / "x.wav")
    assert tuple(wav.shape) == (1, 2, int(2.5 * sr))                     # mono file -> the model's two channels
    emb = ml.get_embedding(wav)
    assert emb.dtype == np.float16 and emb.shape == (150 + 150 + 75, 128) and np.isfinite(emb).all()
    seg0 = ml._get_frame(wav[:, :, :sr]).cpu().numpy()                   # segments do not see each other
    np.testing.assert_allclose(emb[:150].astype(np.float32), seg0, rtol=0, atol=4e-3)   # (conv algorithm choice may differ by an fp16 ulp)
    other = ml.get_embedding(wav[:, :, :2 * sr])                          # the first two segments of a shorter file: same frames
    np.testing.assert_allclose(emb[:300].astype(np.float32), other.astype(np.float32), rtol=0, atol=4e-3)


def test_config4_many_files_device_resident_matches_online_oracle():
    """Config 4, pure-moments variant: 1000 Encodec-shaped files (~2250 frames x 128, float16) fed from HBM in groups of
    256 files -- one update per group, per-file sums and the per-file mean terms on the device -- against the
    reference's file-by-file online merge (utils.py:19-46) of the same files."""
    import torch
    from fadtk_amd.utils import OnlineStats
    rng = np.random.default_rng(44)
    sizes = rng.integers(1500, 3001, size=1000)
    sizes[17], sizes[500] = 2, 9000                                       # a 2-frame file and one longer than a split
    shift = rng.standard_normal(128) * 0.3
    blocks = [((0.6 + 0.8 * rng.random()) * rng.standard_normal((int(n), 128)) + shift + 0.05 * rng.standard_normal(128)).astype(np.float16)
              for n in sizes]
    stats = OnlineStats(128, 0, compat=True)
    for g0 in range(0, 1000, 256):
        grp = blocks[g0:g0 + 256]
        stats.add_group(torch.from_numpy(np.concatenate(grp)).cuda(), [b.shape[0] for b in grp])
    mu, cov = stats.finish()
    assert stats.n_files == 1000 and stats.n_short == 0
    stats.close()
    mu_o, cov_o = O.statistics_online(blocks)
    # mu is the weighted mean of the float16-ROUNDED file means; numpy's own float32 running sum puts a handful of the
    # 128000 file-mean entries one float16 ulp away from the exactly rounded value: 1.2e-4 / 1000 files each
    np.testing.assert_allclose(mu, mu_o, rtol=0, atol=5e-6)
    np.testing.assert_allclose(cov, cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
    # the float16 rounding of the per-file means is visible: the plain raw-moment estimate differs
    x = np.concatenate(blocks).astype(np.float64)
    plain = np.cov(x, rowvar=False)
    assert np.abs(cov - plain).max() > 10 * np.abs(cov - cov_o).max()


def test_batched_loaders_give_each_file_what_it_gets_alone(monkeypatch):
    """``ModelLoader._get_embedding_batch`` (round 6: ONE front-end launch and ONE forward for the files of a group, fad_batch.py)
    returns, file by file, what ``_get_embedding`` returns for the file alone -- frame counts exactly, values to the accuracy a
    convolution's choice of algorithm by batch size leaves (the embeddings are stored as float16)."""
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    import torch
    from fadtk_amd.model_loader import EncodecEmbModel, VGGishModel, WhisperModel
    ml = VGGishModel(); ml.load_model()
    clips = [R.audio_clip(40 + i, int(s * 16000), 16000) for i, s in enumerate((10.0, 3.3, 1.0, 7.9, 10.0))]
    both = ml._get_embedding_batch(clips)
    assert [int(e.shape[0]) for e in both] == [10, 3, 1, 8, 10]
    for c, e in zip(clips, both):
        one = ml._get_embedding(c)
        assert one.shape == e.shape
        torch.testing.assert_close(e, one, rtol=2e-3, atol=2e-3 * float(one.abs().max()))
    ml = WhisperModel("tiny"); ml.load_model()
    clips = [R.audio_clip(50 + i, int(s * 16000), 16000) for i, s in enumerate((4.0, 31.0, 12.5))]
    both = ml._get_embedding_batch(clips)
    for c, e in zip(clips, both):
        one = ml._get_embedding(c)
        assert tuple(e.shape) == (2, 384)
        torch.testing.assert_close(e, one, rtol=2e-3, atol=2e-3 * float(one.abs().max()))
    ml = EncodecEmbModel("24k"); ml.load_model()
    clips = [torch.from_numpy(R.audio_clip(60 + i, 2 * 24000, 24000).astype(np.float32))[None, None, :] for i in range(3)]
    both = ml._get_embedding_batch(clips)
    for c, e in zip(clips, both):
        one = ml._get_embedding(c)
        assert tuple(e.shape) == (150, 128)
        torch.testing.assert_close(e, one, rtol=2e-3, atol=2e-3 * float(one.abs().max()))
    ragged = clips[:2] + [clips[2][:, :, :30000]]                   # clips of different lengths: the per-file loop
    assert [int(e.shape[0]) for e in ml._get_embedding_batch(ragged)] == [150, 150, 94]


def test_batch_driver_drops_only_the_file_that_fails(tmp_path, monkeypatch):
    """A group of files shares one forward; a file the loader cannot embed costs that file alone, the others of its group are embedded
    one by one (the reference's loop loses only the failing file as well: fad_batch.py:18-22 catches per file)."""
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    from fadtk_amd.fad_batch import cache_embedding_files
    from fadtk_amd.model_loader import VGGishModel
    root = _make_set(tmp_path / "set", 7, 2.0, 16000, 900)

    class Picky(VGGishModel):
        def _get_embedding_batch(self, audios):
            if any(abs(float(np.asarray(a)[0]) - self.bad) < 1e-12 for a in audios):
                raise RuntimeError("cannot embed this one")
            return super()._get_embedding_batch(audios)

        def _get_embedding(self, audio):
            return self._get_embedding_batch([audio])[0]

    ml = Picky()
    ml.bad = float(ml.load_wav(root / "clip003.wav")[0])            # (load_wav of the file as stored: what the loop will see)
    cache_embedding_files(root, ml, workers=3)
    got = sorted(p.stem for p in (root / "embeddings" / "vggish").glob("*.npy"))
    assert got == [f"clip{i:03d}" for i in range(7) if i != 3]
    assert np.load(root / "embeddings" / "vggish" / "clip000.npy").shape == (2, 128)
