"""CPU-only checks of the host-side mirror: plugin surface, cache layout, audio plumbing, CLI parsing."""
import subprocess
import sys
import wave
from pathlib import Path

import numpy as np
import pytest

import recipes as R

ROOT = Path(__file__).resolve().parent.parent


def test_cache_path_layout():
    from fadtk_amd import get_cache_embedding_path
    assert get_cache_embedding_path("vggish", "/data/set/song.one.flac") == Path("/data/set/embeddings/vggish/song.one.npy")
    assert get_cache_embedding_path("clap-laion-audio", Path("rel/a.wav")) == Path("rel/embeddings/clap-laion-audio/a.npy")


def test_registry_names_and_plugin_contract():
    from fadtk_amd.model_loader import ModelLoader, get_all_models
    import pickle
    models = get_all_models()
    names = [m.name for m in models]
    # the reference's registry (model_loader.py:676-701) minus the optional DAC / CDPAM
    expect = ["clap-2023", "clap-laion-audio", "clap-laion-music", "vggish"]
    expect += [f"MERT-v1-95M-{v}" for v in range(1, 12)] + ["MERT-v1-95M"]
    expect += ["encodec-emb", "encodec-emb-48k"]
    for fam, sizes in (("w2v2", ("base", "large")), ("hubert", ("base", "large")), ("wavlm", ("base", "base-plus", "large"))):
        for size in sizes:
            last = 24 if size == "large" else 12
            expect += [f"{fam}-{size}-{v}" for v in range(1, last)] + [f"{fam}-{size}"]
    expect += ["whisper-tiny", "whisper-small", "whisper-base", "whisper-medium", "whisper-large"]
    assert names == expect
    dims = {m.name: (m.num_features, m.sr) for m in models}
    assert dims["MERT-v1-95M-4"] == (768, 24000) and dims["w2v2-large-7"] == (1024, 16000) and dims["wavlm-base-plus"] == (768, 16000)
    assert dims["vggish"] == (128, 16000) and dims["encodec-emb"] == (128, 24000)
    assert dims["clap-laion-audio"] == (512, 48000) and dims["whisper-small"] == (768, 16000) and dims["clap-2023"] == (1024, 44100)
    for m in models:                                   # loaders cross process boundaries before load_model()
        assert pickle.loads(pickle.dumps(m)).name == m.name and m.model is None

    class Custom(ModelLoader):                         # README-style plugin
        def __init__(self):
            super().__init__("my-model", 7, 8000, min_len=2)

        def load_model(self):
            self.model = "ready"

        def _get_embedding(self, audio):
            import torch
            return torch.ones((3, 7), dtype=torch.float32) * float(len(audio))

    c = Custom()
    assert c.enforce_min_len(np.zeros(10)).shape == (16000,)        # padded to min_len * sr
    e = c.get_embedding(np.zeros(5))
    assert e.dtype == np.float16 and e.shape == (3, 7) and (e == 5).all()


def test_wav_roundtrip_and_load_wav(tmp_path):
    from fadtk_amd import audio
    from fadtk_amd.model_loader import VGGishModel
    sr = 16000
    x = R.audio_clip(400, sr // 2, sr)
    audio.write_pcm16(tmp_path / "a.wav", x, sr)
    with wave.open(str(tmp_path / "a.wav")) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, sr, sr // 2)
    pcm, fs = audio.read_pcm16(tmp_path / "a.wav")
    assert fs == sr and pcm.dtype == np.int16
    np.testing.assert_array_equal(pcm, np.clip(np.rint(x.astype(np.float64) * 32768), -32768, 32767).astype(np.int16))
    wav = VGGishModel().load_wav(tmp_path / "a.wav")              # int16 / 32768, zero padded to 1 s
    assert wav.dtype == np.float64 and wav.shape == (sr,) and np.all(wav[sr // 2:] == 0)
    np.testing.assert_array_equal(wav[: sr // 2], pcm / 32768.0)



def test_resampler_oracle_properties():
    """The CPU restatement of fadtk's Kaiser-sinc resampler (oracle/audio_oracle.py; the GPU kernel is checked against
    it in test_gpu_pipeline.py): a 1 kHz tone survives 48k -> 16k with its amplitude, an above-Nyquist tone is removed,
    lengths follow ceil(new * n / orig), equal rates pass through, the table has torchaudio's shape."""
    from oracle import audio_oracle as AO
    t = np.arange(48000) / 48000.0
    low, high = np.sin(2 * np.pi * 1000 * t), np.sin(2 * np.pi * 15000 * t)
    y = AO.resample_kaiser((low + high).astype(np.float32), 48000, 16000)
    assert y.shape == (16000,)
    ref = np.sin(2 * np.pi * 1000 * np.arange(16000) / 16000.0)
    assert np.abs(y[500:-500] - ref[500:-500]).max() < 2e-3
    assert AO.resample_kaiser(low[:1001].astype(np.float32), 44100, 16000).shape == (364,)        # ceil(160 * 1001 / 441) = ceil(363.2)
    np.testing.assert_array_equal(AO.resample_kaiser(low.astype(np.float32), 16000, 16000), low.astype(np.float32).astype(np.float64))
    k, width, orig, new = AO.sinc_kernel(44100, 16000)
    assert (orig, new, width) == (441, 160, 187) and k.shape == (160, 2 * 187 + 441) and k.dtype == np.float32
    q = AO.pcm16_roundtrip(np.array([0.5, -1.5, 1.0, 1e-6]))
    np.testing.assert_array_equal(q, [0.5, -1.0, 32767 / 32768, 0.0])


def test_cli_surface():
    out = subprocess.run([sys.executable, "-m", "fadtk", "--help"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0
    for flag in ("--workers", "--sox-path", "--inf", "--indiv", "baseline", "eval", "csv"):
        assert flag in out.stdout
    out = subprocess.run([sys.executable, "-m", "fadtk.embeds", "--help"], capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "--models" in out.stdout and "--dirs" in out.stdout
    bad = subprocess.run([sys.executable, "-m", "fadtk", "not-a-model", "a", "b"], capture_output=True, text=True, cwd=ROOT)
    assert bad.returncode == 2 and "invalid choice" in bad.stderr


def test_shard_matches_array_split():
    from fadtk_amd import dist
    items = list(range(23))
    for w in (1, 2, 3, 8, 30):
        got = [dist.shard(items, r, w) for r in range(w)]
        want = [list(a) for a in np.array_split(items, w)]
        assert got == want


def test_wav2vec2_family_loader_layer_and_truncation(monkeypatch):
    """HuBERT-style loaders return hidden state `layer` as [frames, D] fp16 (50 frames/s at 16 kHz) and cut the
    audio at `limit_minutes` (reference model_loader.py:580-595).  Seeded random weights: shapes only."""
    monkeypatch.setenv("FADTK_AMD_RANDOM_WEIGHTS", "1")
    from fadtk_amd.model_loader import HuBERTModel
    m = HuBERTModel("base", layer=2, limit_minutes=1.0 / 60.0)          # one second
    assert m.name == "hubert-base-2" and m.limit == 16000
    m.load_model()
    x = R.audio_clip(5, 3 * 16000, 16000)
    e_long, e_cut = m.get_embedding(x), m.get_embedding(x[:16000])
    assert e_long.dtype == np.float16 and e_long.shape == (49, 768)
    assert np.array_equal(e_long, e_cut)                                # 3 s were truncated to the same 1 s


def test_bench_gpus_n_builds_the_torchrun_command_itself():
    """`python bench.py --gpus 2` from a plain shell (no WORLD_SIZE) must start the ranks itself -- here only
    the command is printed (FAD_BENCH_PRINT_LAUNCH=1): torch.distributed.run, one process per GPU, loopback rendezvous, the
    original flags passed through.  Under torchrun (WORLD_SIZE set) the same flags must NOT relaunch."""
    import json
    import os
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["FAD_BENCH_PRINT_LAUNCH"] = "1"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    cmd = json.loads(r.stdout.strip().split("\n")[-1])
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=2" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    i = cmd.index(str(ROOT / "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "2", "--steps", "5", "--warmup", "1"]
    sys.path.insert(0, str(ROOT))
    import bench
    assert bench.torchrun_command(8, ["--gpus", "8"])[4] == "--nproc-per-node=8"


def test_tile256_role_tables_cover_the_upper_triangle(tmp_path):
    """fadtk_amd/csrc/tile256_roles.h (shared by the 256-column-slab moments kernel, its planner and its reduce): every 32 x 32
    block on or above the diagonal is owned by exactly one wave of one work item (two on a Z item's triangle), for 1..8 superblocks,
    with the same matrix-pipe load on every SIMD.  Plain C++, checked with g++ (no GPU)."""
    exe = tmp_path / "tile256_cover"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), str(ROOT / "tests" / "native_cpu" / "tile256_cover.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "nsb=2: 2 item types, 136 blocks per split, 136 output blocks ok" in r.stdout
    assert "nsb=3: 5 item types" in r.stdout and "nsb=8: 32 item types" in r.stdout


def test_batched_chain_workgroups_take_every_tile_once_and_share_the_xcds(tmp_path):
    """fadtk_amd/csrc/big_slots.h (the batched square-root chain's workgroup -> (problem, tile) map, shared with the host's grid sizes):
    every tile of every problem once, a problem of a whole group of eight on one XCD, the leftover problems cut evenly over all eight
    -- for 1..70 problems.  Plain C++, checked with g++ (no GPU)."""
    exe = tmp_path / "big_slots_cover"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), str(ROOT / "tests" / "native_cpu" / "big_slots_cover.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "20 problems x 16 items -> grid 320, 40 per XCD" in r.stdout


def test_adaptive_step_scales_never_give_up_and_need_fewer_iterations_than_plain_steps():
    """The rule that sets the per-song step scale of the batched low-precision chain (csrc/ns_check.h, csrc/ns_fast.h: nsf_check), emulated
    on eigenvalues (scripts/ns_emulate_adaptive.py): from a thirtieth to three times the x_min estimate the capped rule closes every problem
    (a negative count = the chain's 'residual grows' rule would have handed the song to the float64 routes), at the shipped start (half the
    estimate; the resident D = 128 kernel: the estimate) in no more iterations than the plain step from the old scale c = u / 2.9 needs."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ns_emulate_adaptive", ROOT / "scripts" / "ns_emulate_adaptive.py")
    E = importlib.util.module_from_spec(spec); spec.loader.exec_module(E)
    rng = np.random.default_rng(7)
    for d, n, lo, hi, shipped in ((256, 600, 0.5, 1.5, 0.5), (128, 2250, 0.6, 1.4, 1.0)):
        x, pr = E.song_spectrum(rng, d, n, lo, hi)
        est = E.l0_estimate(pr, d)
        counts = {sc: E.run(x, min(est * sc, 0.5), cap=True) for sc in (0.03, 0.1, 0.25, 0.5, 1.0, 2.0, 3.0)}
        assert all(0 < c < 16 for c in counts.values()), counts
        # plain steps from c = u / 2.9 (x larger by sqrt(2.9), every mu = 1): the round-3 chain
        xp = x * np.sqrt(2.9)
        k, res = 0, 1e300
        while res > 1e-3 and k < 40:
            xp = xp * (3 - xp * xp) / 2
            res = float(np.sqrt(np.sum((1 - xp * xp) ** 2)))
            k += 1
        assert counts[shipped] <= k, (counts, k)


def test_wide_chain_acceptance_rule_emulated():
    """Round 5's acceptance rule for pairs with a DECAYING spectrum (csrc/ns_fast.h: SP_V2 / SP_V3; frechet.hip: fast_decide_one), emulated in
    numpy on split-float16 arithmetic (scripts/ns_emulate_verify.py), D = 256: a k^-1 pair finishes on the chain, the norm bound says nothing
    (it is orders of magnitude above the bar), the verification products accept it and the TRUE error of the corrected trace is inside
    the estimate and far inside 1e-5 of the distance; a k^-2 pair is declined before it starts (its x_min estimate is below what float16 can
    carry) -- the float64 route."""
    import importlib.util
    import warnings
    spec = importlib.util.spec_from_file_location("ns_emulate_verify", ROOT / "scripts" / "ns_emulate_verify.py")
    E = importlib.util.module_from_spec(spec); spec.loader.exec_module(E)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        flat = E.emulate(*E.bench_pair(0.0, d=256, n=20000))
        k1 = E.emulate(*E.bench_pair(1.0, d=256, n=20000))
        k2 = E.emulate(*E.bench_pair(2.0, d=256, n=20000))
    assert flat["route"] == "chain" and flat["accepted_by"] == "norm bound" and flat["fad_rel_err"] < 1e-8, flat
    assert k1["route"] == "chain" and k1["scaled"] and k1["accepted_by"] == "verification", k1
    assert k1["norm_bound_fad"] > 1e-3 and k1["iters"] <= 14, k1
    assert k1["fad_rel_err"] <= k1["verify_est_fad"] <= 1e-5, k1
    assert k2["route"] == "declined", k2


def test_out_of_memory_is_a_type_not_a_wording():
    """score_inf falls back to the point-by-point route on an out-of-memory error of the device route -- recognised by TYPE
    (the library's FadOutOfMemory for FAD_ERR_ALLOC, torch's OutOfMemoryError), not by what the message happens to say."""
    import torch
    from fadtk_amd import _capi as K
    with pytest.raises(K.FadOutOfMemory):
        K.check(K.FAD_ERR_ALLOC, "somewhere")
    assert issubclass(K.FadOutOfMemory, RuntimeError)
    assert K.is_out_of_memory(K.FadOutOfMemory("any wording at all"))
    assert K.is_out_of_memory(torch.cuda.OutOfMemoryError("x"))
    assert not K.is_out_of_memory(RuntimeError("out of memory"))        # a RuntimeError that merely SAYS so is not one
    with pytest.raises(RuntimeError) as ei:
        K.check(K.FAD_ERR_HIP, "somewhere")
    assert not isinstance(ei.value, K.FadOutOfMemory) or "memory" in str(ei.value).lower()


def test_convert_to_model_rate_copies_pcm16_mono_at_the_models_rate(tmp_path):
    """fad.py:139-186 normalises every file to mono PCM16 at the model's rate; for a file that already is one the decode -> float -> quantise
    round trip returns the very samples, so the cache file is written from the input's frames directly (no GPU needed: this runs on CPU).
    Full-scale and odd values included: the copy must equal what the general path's arithmetic gives."""
    from fadtk_amd import audio
    rng = np.random.default_rng(5)
    pcm = rng.integers(-32768, 32768, size=16000, dtype=np.int64).astype("<i2")
    pcm[:4] = [-32768, 32767, 0, -1]
    src = tmp_path / "in.wav"
    with wave.open(str(src), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    audio.convert_to_model_rate(src, tmp_path / "convert" / "16000" / "in.wav", 16000)
    got, sr = audio.read_pcm16(tmp_path / "convert" / "16000" / "in.wav")
    assert sr == 16000 and np.array_equal(got, pcm)
    x, _ = audio.read_audio(src)                                    # the general path's arithmetic on the same samples
    q = np.clip(np.rint(x.mean(axis=0).astype(np.float64) * 32768.0), -32768, 32767).astype("<i2")
    assert np.array_equal(q, pcm)
