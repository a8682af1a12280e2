#!/usr/bin/env python3
"""bench.py -- FAD hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Workload (BASELINE.json configs[2], "C3"): two synthetic float16 embedding matrices of
[100000 x 512] per GPU, RESIDENT IN HBM when the timed region starts.  One *step* is one pass of the
hot path over that batch -- one FAD score:

    moments(A, B)      hand-written HIP, fp16 MFMA E^T E + column sums -> (n, sum x, sum xxT) of BOTH sets in one launch
                       of each kernel (fad_moments_update_multi; D = 512: the 256-column-slab kernel, moments_tile256.h)
    [N>1]              ONE in-place all-reduce (RCCL/xGMI) over the buffer that holds both sets' packed float64
                       statistics (fadtk_amd.dist.SharedStats -- the same object the product's --gpus path uses)
    frechet(A, B)      finalise (mu, Sigma) x2, Newton-Schulz sqrt(S1 S2): split-float16 MFMA iterations + exact int8-MFMA products +
                       one float64-accurate correction (float64 throughout when the product is ill-conditioned), the reference's
                       float16 mean term.  The chains of the scores in flight run as ONE batch of 16 (--batch; the steps left at the end of a call join the last batch).

i.e. exactly one FAD score over the union of all ranks' rows.  With N GPUs every rank holds its own
100k-row shard of both sets (weak scaling: rows grow with N), so `value` is reported in
config-3-sized score workloads per second:  value = N * K / seconds  (at N=1: plain FAD scores/s).

Rank 0 prints ONE JSON line with the driver's fields plus
  roofline          dominant kernel (moments tile kernel): achieved TFLOP/s from ALGORITHMIC flops
                    2 sets x 2*N*D^2 per launch / mean launch duration (HIP events on the launch stream,
                    recorded inside the timed region by the library), vs the dense fp16 MFMA peak
  roofline_frechet  the square-root chain: GEMMs x 2 D^3 / its duration (events on the same stream)
  cpu_baseline      the numpy/scipy oracle (a line-by-line restatement of fadtk's CPU path, both
                    sqrtm and eig as in fad.py:88-92) timed on this node's host cores, rank 0, N=1
  extra             config-4 pure-moments pass (files of [2250 x 128] frames, per-file mean terms included) and the
                    config-5 per-song pass (10k two-frame songs at D=768) with its own CPU baseline and roofline
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

N_ROWS = 100_000
DIM = 512
SETS = 2                                # frame matrices per launch of the tile kernel (the two sets of a score)
MFMA_F16_PEAK_TFLOPS = 2500.0           # dense, /opt/skills/guides/MI355X_MICROARCH.md
MFMA_F32_PEAK_TFLOPS = 157.3            # f32-input MFMA = the fp32 vector rate, same guide
HBM_PEAK_GBS = 8000.0
FAD_F16 = 0                             # fad_dtype code: the reference's float16 mean term
DEFAULT_INFLIGHT = 3


N_PAIRS = 4                              # distinct (A, B) pairs rotated through the timed loop: 4 x 204.8 MB = 819 MB > the 256 MiB
                                         # Infinity Cache, so every step streams its frames from HBM (a caller scores each set once)


def golden_g7():
    """The reference's own scalar for the C3 recipe (tests/golden/golden.json: g7, written by tests/golden/make_golden.py from the imported
    reference) and the recipe's generator (tests/golden/recipes.py: numpy default_rng seeds 10 / 11) -- fixtures, not the oracle."""
    try:
        sys.path.insert(0, str(ROOT / "tests" / "golden"))
        import recipes
        g7 = json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())["g7"]
        return recipes, g7
    except Exception:       # noqa: BLE001  (a checkout without tests/: the pair falls back to the device generator)
        return None, None
    finally:
        if sys.path and sys.path[0].endswith("golden"):
            sys.path.pop(0)


def make_sets(torch, device, rank, pair=0):
    """C3 recipe: A ~ N(0,1), B ~ 1.02 N(0,1) + 0.01, float16.  Pair 0 of rank 0 is LITERALLY the golden G7 pair --
    `recipes.c3_pair()` (numpy default_rng, seeds 10 / 11), generated on the host and uploaded once -- so that the line can hold the
    reference's own scalar against the GPU's (`parity_rel_err_vs_golden_g7`); further pairs and ranks come from the device generator."""
    if pair == 0 and rank == 0:
        recipes, g7 = golden_g7()
        if recipes is not None and (N_ROWS, DIM) == (int(g7["n"]), int(g7["d"])):
            a, b = recipes.c3_pair()
            return torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)
    g = torch.Generator(device=device)
    g.manual_seed(10 + 100 * pair + 1000 * rank)
    a = torch.randn((N_ROWS, DIM), generator=g, device=device, dtype=torch.float32).to(torch.float16)
    g.manual_seed(11 + 100 * pair + 1000 * rank)
    b = (1.02 * torch.randn((N_ROWS, DIM), generator=g, device=device, dtype=torch.float32) + 0.01).to(torch.float16)
    return a, b


def make_realistic_sets(torch, device, pair=0):
    """What a pair of real embedding sets looks like next to the C3 recipe: covariance spectra k^-1 in a shared random basis
    (extra_decaying's recipe: the product Sigma_1 Sigma_2 then has condition ~3e5) and frames that are NOT centred -- every dimension
    carries an offset of half the mean standard deviation (post-activation features do) -- so that the reference's own float32
    running-sum mean (fad.py:48) differs from the rounded exact mean.  Same size as C3: 2 x [100000 x 512] float16."""
    g = torch.Generator(device=device); g.manual_seed(77)
    q, _ = torch.linalg.qr(torch.randn((DIM, DIM), generator=g, device=device, dtype=torch.float64))
    q = q.to(torch.float32)
    lam = (torch.arange(1, DIM + 1, device=device, dtype=torch.float64) ** -0.5).to(torch.float32)
    off = 0.5 * float(torch.sqrt((lam.double() ** 2).mean()))
    g.manual_seed(770 + 10 * pair)
    a = ((torch.randn((N_ROWS, DIM), generator=g, device=device, dtype=torch.float32) * lam) @ q.T + off).to(torch.float16)
    g.manual_seed(771 + 10 * pair)
    b = ((1.05 * torch.randn((N_ROWS, DIM), generator=g, device=device, dtype=torch.float32) * lam) @ q.T + off + 0.01).to(torch.float16)
    return a, b


def spread(ms):
    """median / min / max of repeated timings (ms)."""
    ms = np.asarray(ms, dtype=np.float64)
    return {"median": float(np.median(ms)), "min": float(ms.min()), "max": float(ms.max()), "runs": int(ms.size)}


def _blas_threads():
    threads, blas = os.cpu_count(), "unknown"
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = info[0].get("num_threads", threads)
            blas = f"{info[0].get('internal_api')} {info[0].get('version')}"
    except Exception:       # noqa: BLE001
        pass
    return int(threads), blas


def cpu_quota_cores():
    """CPUs' worth of time the process's CPU-bandwidth cgroup grants per period (cgroup v2 cpu.max, v1 cfs_quota / cfs_period), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:       # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:       # noqa: BLE001
        return None


def cpu_baseline(a_host, b_host):
    """Reference CPU path (oracle port) on the same arrays as the GPU run, best of 3 after one warm-up -- with the BLAS pool as it comes
    (one thread per physical core the library sees) and, where the process sits in a CPU-bandwidth cgroup, once more with as many threads as
    the cgroup grants CPUs: on the GPU boxes that is 16 of 256 logical CPUs (cpu.max = 1600000 100000), and 64 threads spend most of every
    period frozen (cpu.stat nr_throttled).  `value` is the better of the two, `cores` the threads it used."""
    from oracle import fad_oracle as O
    threads, blas = _blas_threads()

    def best_of(n=4):
        times, f = [], None
        for it in range(n):
            t0 = time.perf_counter()
            f = O.fad_between(a_host, b_host)
            dt = time.perf_counter() - t0
            if it > 0:
                times.append(dt)
        return min(times), f
    t_default, fad = best_of()
    runs = {str(threads): 1.0 / t_default}
    best_t, best_threads = t_default, threads
    quota = cpu_quota_cores()
    if quota and int(quota) >= 1 and int(quota) < threads:
        try:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=int(quota), user_api="blas"):
                t_q, _ = best_of()
            runs[str(int(quota))] = 1.0 / t_q
            if t_q < best_t:
                best_t, best_threads = t_q, int(quota)
        except Exception:       # noqa: BLE001
            pass
    import scipy
    return {"value": 1.0 / best_t, "unit": "FAD scores/s", "cores": best_threads, "kind": "port",
            "sample": f"full config-3 workload (2 x [{N_ROWS}x{DIM}] fp16 -> 1 score), best of 3 after 1 warm-up; "
                      f"{os.cpu_count()} logical CPUs, cgroup CPU quota {quota}, BLAS {blas}, numpy {np.__version__}, scipy {scipy.__version__}",
            "scores_per_s_by_blas_threads": runs, "seconds_best": best_t}, float(fad)


def cpu_quiet(seconds=0.35):
    """Every side measurement ends with its oracle -- seconds of BLAS on all host cores.  On the GPU boxes the process sits in a CPU-bandwidth
    cgroup: such a burst exhausts the quota and the cgroup's threads are frozen for the rest of the 100 ms periods that follow (cpu.stat
    `nr_throttled`; scripts/probe_stall.py shows it), the BLAS pool keeps spinning for a while on top.  A short sleep lets both pass before the
    next measurement starts its clock (r05i -> r05q: `host_resident` 184 scores/s inside the line, 202 in a process of its own)."""
    time.sleep(seconds)


def extra_c4(torch, hip, device, local_rank):
    """Config 4, pure-moments variant: Encodec-shaped files of [2250 x 128] float16 frames fed the way
    the product feeds them -- groups of files, ONE update per group (fad_moments_update_segmented: tile kernel + reduce
    + per-file column sums), plus the per-file mean terms of the reference's online path
    (fad_moments_update_file_means) -- all on data resident in HBM.  Wall time of the whole pass (four groups of 4096 files),
    five passes, median reported."""
    from fadtk_amd.utils import OnlineStats
    files_per_group, rows_per_file, d, groups, passes = 4096, 2250, 128, 4, 5
    x = torch.randn((files_per_group * rows_per_file, d), device=device, dtype=torch.float16)
    sizes = np.full(files_per_group, rows_per_file, dtype=np.int64)
    # (ref_means=False: the per-file float16 means are the rounded EXACT ones -- rounds 1-4's pass, one read of the frames; the
    #  reference-order means of round 5 cost a second walk over every group: `with_reference_order_file_means` below)
    stats = OnlineStats(d, local_rank, compat=True, ref_means=False)
    stats.add_group(x, sizes)                                    # warm-up (allocations)
    torch.cuda.synchronize()
    ms = []
    for _ in range(passes):
        t0 = time.perf_counter()
        for _ in range(groups):
            stats.add_group(x, sizes)
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
    dt = float(np.median(ms)) * 1e-3
    ref_ms = None
    try:
        sref = OnlineStats(d, local_rank, compat=True, ref_means=True)
        sref.add_group(x, sizes)
        torch.cuda.synchronize()
        tr = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(groups):
                sref.add_group(x, sizes)
            torch.cuda.synchronize()
            tr.append((time.perf_counter() - t0) * 1e3)
        ref_ms = float(np.median(tr))
        sref.close()
    except Exception:       # noqa: BLE001
        ref_ms = None
    stats.frames.set_timing(True)
    for _ in range(3):
        stats.frames.update(x)
    k_ms, r_ms, _ = stats.frames.last_timing()
    stats.frames.set_timing(False)
    mu, cov = stats.finish()
    stats.close()
    # the same 16384 files as ONE update (9.4 GB resident): what the ~0.17 ms of small kernels and launch gaps per update cost the pass above
    one_ms = None
    try:
        x4 = x.repeat(groups, 1)
        sizes4 = np.full(groups * files_per_group, rows_per_file, dtype=np.int64)
        st4 = OnlineStats(d, local_rank, compat=True, ref_means=False)
        st4.add_group(x4, sizes4)
        torch.cuda.synchronize()
        t4 = []
        for _ in range(3):
            t0 = time.perf_counter(); st4.add_group(x4, sizes4); torch.cuda.synchronize(); t4.append((time.perf_counter() - t0) * 1e3)
        one_ms = float(np.median(t4))
        st4.close()
        del x4
    except Exception:       # noqa: BLE001  (memory on a shared box ...)
        one_ms = None
    nbytes = groups * x.numel() * 2
    return {"files": groups * files_per_group, "frames_per_file": rows_per_file, "dim": d, "files_per_update": files_per_group,
            "ms": dt * 1e3, "ms_spread": spread(ms), "frames_per_s": groups * x.shape[0] / dt, "GBps_algorithmic": nbytes / dt / 1e9,
            "frac_of_8TBps": nbytes / dt / 1e9 / HBM_PEAK_GBS, "includes": "tile kernel + reduce + per-file sums + per-file mean terms",
            "tile_kernel_ms_per_update": k_ms, "tile_kernel_frac_of_8TBps": x.numel() * 2 / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "one_update_of_all_files": ({"ms": one_ms, "frac_of_8TBps": nbytes / (one_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if one_ms else None),
            "with_reference_order_file_means": ({"ms": ref_ms, "frac_of_8TBps_one_read": nbytes / (ref_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "frac_of_8TBps_two_reads": 2 * nbytes / (ref_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "per-file means as np.mean forms them (utils.py:16: float32 running sum per file) -- what OnlineStats does by "
                                                         "default.  Round 6: the tile kernel's diagonal workgroups walk them in the SAME pass (files of one run each, "
                                                         "fad_moments_update_segmented_ref): the frames cross HBM once; `frac_of_8TBps_two_reads` is kept for comparison with "
                                                         "round 5, when a second walk read every group again (4.1 ms)"} if ref_ms else None),
            "cov_trace_per_dim": float(np.trace(cov)) / d}


def extra_c2_vggish(torch, hip, device, local_rank, n_files=1000):
    """Config 2 END TO END: two sets of `n_files` synthetic ten-second 16 kHz PCM16 wavs on disk -> normalised-audio cache
    (fad.py:139-186) -> HIP log-mel front end + VGGish forward (seeded random weights: no checkpoint exists offline), 64 files per launch /
    forward -> float16 embedding cache (.npy per file, written behind the loop) + statistics while the frames are in HBM (the online
    path's per-file float16 means included) -> FAD from the cached statistics.  Checked against the oracle on the very same cached
    embeddings.  Reference path: fadtk/fad_batch.py:15-48, fad.py:188-201, model_loader.py:99-108."""
    import shutil
    import tempfile
    import fadtk_amd
    from fadtk_amd import audio
    from fadtk_amd.fad_batch import embed_and_accumulate
    from fadtk_amd.model_loader import VGGishModel
    from oracle import fad_oracle as O
    os.environ["FADTK_AMD_RANDOM_WEIGHTS"] = "1"
    root = Path(tempfile.mkdtemp(prefix="fad_c2_"))
    sr, secs = 16000, 10
    t = np.arange(sr * secs) / sr
    try:
        t0 = time.perf_counter()
        for name, seed0, gain in (("base", 0, 1.0), ("eval", 100000, 0.8)):
            (root / name).mkdir()
            for i in range(n_files):
                rng = np.random.default_rng(seed0 + i)
                x = 0.1 * rng.standard_normal(sr * secs)
                for f0, a in zip(rng.uniform(80.0, 4000.0, 3), rng.uniform(0.02, 0.15, 3)):
                    x += a * np.sin(2.0 * np.pi * f0 * t + rng.uniform(0, 6.28))
                audio.write_pcm16(root / name / f"clip{i:04d}.wav", gain * x, sr)
        gen_s = time.perf_counter() - t0
        ml = VGGishModel()
        fad = fadtk_amd.FrechetAudioDistance(ml, audio_load_worker=8)          # loads the model once
        # (warm-up on clips that are not the measured files: MIOpen searches its convolution algorithms the first time it sees a batch shape --
        #  seconds, once per process)
        warm = [0.1 * np.random.default_rng(7).standard_normal(sr * secs).astype(np.float32) for _ in range(VGGishModel.batch_files)]
        ml._get_embedding_batch(warm); ml._get_embedding_batch(warm[:1])
        torch.cuda.synchronize()
        out = {}
        t0 = time.perf_counter()
        for name in ("base", "eval"):
            t1 = time.perf_counter()
            embed_and_accumulate(root / name, ml, workers=8)
            torch.cuda.synchronize()
            out[name + "_s"] = time.perf_counter() - t1
        embed_s = time.perf_counter() - t0
        t1 = time.perf_counter()
        score = float(fad.score(root / "base", root / "eval"))
        score_s = time.perf_counter() - t1
        # the same files once more from the normalised-audio cache, embeddings removed: the loop without the one-off PCM conversion
        shutil.rmtree(root / "eval" / "embeddings"); shutil.rmtree(root / "eval" / "stats")
        t1 = time.perf_counter()
        embed_and_accumulate(root / "eval", ml, workers=8)
        torch.cuda.synchronize()
        warm_s = time.perf_counter() - t1
        # ... and the reference's shape of the loop (one file per launch / forward) on a 100-file sample
        shutil.rmtree(root / "eval" / "embeddings"); shutil.rmtree(root / "eval" / "stats")
        some = sorted((root / "eval").glob("*.wav"))[:100]
        sub = root / "sub"; (sub / "convert" / str(sr)).mkdir(parents=True)
        for pth in some:                                                   # (with their normalised-audio cache: the loop alone is compared)
            shutil.copy(pth, sub / pth.name)
            shutil.copy(root / "eval" / "convert" / str(sr) / pth.name, sub / "convert" / str(sr) / pth.name)
        ml.batch_files = 1
        t1 = time.perf_counter()
        embed_and_accumulate(sub, ml, workers=8)
        torch.cuda.synchronize()
        per_file_s = time.perf_counter() - t1
        del ml.batch_files
        blocks = {name: [np.load(q) for q in sorted((root / name / "embeddings" / "vggish").glob("*.npy"))] for name in ("base",)}
        shutil.rmtree(root / "eval" / "embeddings", ignore_errors=True)
        embed_and_accumulate(root / "eval", ml, workers=8)
        blocks["eval"] = [np.load(q) for q in sorted((root / "eval" / "embeddings" / "vggish").glob("*.npy"))]
        t1 = time.perf_counter()
        want = float(O.frechet_distance(*O.statistics_online(blocks["base"]), *O.statistics_online(blocks["eval"]), run_sqrtm=False))
        oracle_s = time.perf_counter() - t1
        frames = sum(b.shape[0] for v in blocks.values() for b in v)
        return {"files": 2 * n_files, "seconds_per_file": secs, "frames": int(frames), "dim": 128, "files_per_forward": VGGishModel.batch_files,
                "embed_and_accumulate_s": embed_s, "first_set_s": out["base_s"], "second_set_s": out["eval_s"],
                "files_per_s": 2 * n_files / embed_s, "frames_per_s": frames / embed_s,
                "files_per_s_from_the_audio_cache": n_files / warm_s,
                "files_per_s_from_the_audio_cache_one_file_per_forward": 100 / per_file_s, "score_s": score_s, "fad": score, "fad_oracle_on_the_cached_embeddings": want,
                "parity_rel_err_vs_oracle": abs(score - want) / abs(want), "oracle_s": oracle_s, "wav_generation_s": gen_s,
                "includes": "wav decode + PCM16 normalisation cache + HIP log-mel + VGGish forward (random weights) + .npy cache + online statistics; "
                            "host side: 8 decode threads, one writer thread"}
    finally:
        shutil.rmtree(root, ignore_errors=True)


def extra_c4_encodec(torch, hip, device, local_rank, clips_per_set=32):
    """Config 4 WITH the embedder on a bounded sample: thirty-second 24 kHz clips generated on the device (72 GB of PCM for
    the full 50k files would swamp any disk), Encodec's SEANet encoder (seeded random weights) over 8 clips per forward, the frames rounded
    to float16 as model_loader.py:47-48 stores them and folded into the online statistics while still in HBM (one fad_moments_update_segmented
    per group).  Two sets of `clips_per_set`; the FAD between them against the oracle on the same frames copied to the host."""
    import fadtk_amd
    from fadtk_amd.model_loader import EncodecEmbModel
    from fadtk_amd.utils import OnlineStats
    from oracle import fad_oracle as O
    os.environ["FADTK_AMD_RANDOM_WEIGHTS"] = "1"
    ml = EncodecEmbModel("24k"); ml.load_model()
    sr, secs, B = 24000, 30, ml.batch_files
    g = torch.Generator(device=device).manual_seed(4)
    tt = torch.arange(sr * secs, device=device, dtype=torch.float32) / sr

    def clips(n, gain):
        x = 0.1 * torch.randn((n, 1, sr * secs), device=device, generator=g)
        f0 = (100.0 if gain == 1.0 else 2000.0) + 3000.0 * torch.rand((n, 1, 1), device=device, generator=g)
        return gain * (x + 0.2 * torch.sin(2.0 * np.pi * f0 * tt))

    def run(gain, keep):
        st = OnlineStats(128, local_rank, compat=True)
        host = []
        enc_s = 0.0
        for o in range(0, clips_per_set, B):
            wav = clips(min(B, clips_per_set - o), gain)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            embs = ml._get_embedding_batch(list(wav.split(1)))
            rows = torch.cat([e.to(torch.float16) for e in embs], dim=0).contiguous()
            torch.cuda.synchronize(); enc_s += time.perf_counter() - t1
            st.add_group(rows, [int(e.shape[0]) for e in embs])
            if keep:
                host.extend(np.ascontiguousarray(e.to(torch.float16).cpu().numpy()) for e in embs)
        mu, cov = st.finish()
        n = st.frames.count
        st.close()
        return mu, cov, n, host, enc_s

    run(1.0, False)                                  # warm-up: MIOpen picks its convolution algorithms on the first forward
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mu_a, cov_a, n_a, host_a, enc_a = run(1.0, True)
    mu_b, cov_b, n_b, host_b, enc_b = run(0.5, True)
    torch.cuda.synchronize()
    total_s = time.perf_counter() - t0
    score = float(fadtk_amd.calc_frechet_distance(mu_a, cov_a, mu_b, cov_b))
    want = float(O.frechet_distance(*O.statistics_online(host_a), *O.statistics_online(host_b), run_sqrtm=False))
    return {"clips": 2 * clips_per_set, "seconds_per_clip": secs, "frames": int(n_a + n_b), "dim": 128, "clips_per_forward": B,
            "s": total_s, "clips_per_s": 2 * clips_per_set / total_s, "frames_per_s": (n_a + n_b) / total_s, "audio_seconds_per_s": 2 * clips_per_set * secs / total_s,
            "encoder_forward_s": enc_a + enc_b, "statistics_and_copies_s": total_s - enc_a - enc_b,
            "fad": score, "fad_oracle_on_the_same_frames": want, "parity_rel_err_vs_oracle": abs(score - want) / abs(want),
            "full_config4_at_this_rate_s_per_gpu": 50000 / 8 / (2 * clips_per_set / total_s),
            "includes": "device-generated clips -> SEANet encoder (random weights, PyTorch-ROCm / MIOpen) -> float16 frames -> fused online statistics (HIP); "
                        "the host copies of the frames are for the oracle only"}


def extra_c5(torch, hip, device):
    """Config 5 shape (Whisper-small): 10k two-frame songs at D=768 against one baseline, one batched
    call (seven calls timed, median); CPU baseline = 16 of the same songs through the oracle on a pool of 8 threads
    (fad.py:387) -- SURVEY 8-d4 asks for 64, which is ~2 minutes of host time at the ~2 s a 768-dimensional eig + sqrtm takes:
    the cost per song does not depend on the data (dense LAPACK on D x D), the per-song seconds of the sample are reported
    so that the extrapolation can be judged; the GPU scores of those songs are checked against the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import fad_oracle as O
    nsongs, d5 = 10_000, 768
    g5 = torch.Generator(device=device); g5.manual_seed(5)
    songs = torch.randn((2 * nsongs, d5), generator=g5, device=device).to(torch.float16)
    base = torch.randn((3 * d5, d5), generator=g5, device=device, dtype=torch.float64)
    mu5 = base.mean(0).cpu().numpy(); cov5 = torch.cov(base.T).cpu().numpy()
    offs = np.arange(0, 2 * nsongs + 1, 2)
    mu5_d, cov5_d = torch.from_numpy(mu5).to(device), torch.from_numpy(cov5).to(device)      # the baseline is resident in HBM as well
    hip.frechet_batched(mu5_d, cov5_d, songs, offs)
    ms = []
    for _ in range(7):
        torch.cuda.synchronize(); t5 = time.perf_counter()
        sc5, st5 = hip.frechet_batched(mu5_d, cov5_d, songs, offs)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t5) * 1e3)
    dt5 = float(np.median(ms)) * 1e-3
    n_cpu = 16
    sample = songs[:2 * n_cpu].cpu().numpy()
    blocks = [sample[2 * i:2 * i + 2] for i in range(n_cpu)]
    per_song = []

    def one(sg):
        t = time.perf_counter()
        r = O.individual_scores(mu5, cov5, [sg], run_sqrtm=True)[0]
        per_song.append(time.perf_counter() - t)
        return r

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=8) as ex:
        want = list(ex.map(one, blocks))
    dt_cpu = time.perf_counter() - t0
    want = np.array([np.nan if w is None else float(w) for w in want])
    rel = float(np.nanmax(np.abs(sc5[:n_cpu] - want) / np.abs(want)))
    t64 = -(-d5 // 64)
    flops = 2.0 * (-(-nsongs // d5) * d5) * d5 * d5 * (t64 + 1) / (2 * t64)       # issued: W = Dm U, U upper triangular in 64-wide column tiles
    return {"songs": nsongs, "dim": d5, "frames_per_song": 2, "ms": dt5 * 1e3, "ms_spread": spread(ms), "songs_per_s": nsongs / dt5,
            "ok": int((st5 == 0).sum()), "max_rel_err_vs_oracle_sample": rel,
            "cpu_baseline": {"value": n_cpu / dt_cpu, "unit": "songs/s", "cores": 8, "kind": "port",
                             "sample": f"{n_cpu} of the same songs through the oracle (eig + sqrtm per song, fad.py:373-378) on a "
                                       "thread pool of 8 (fad.py:387, BLAS threads as numpy finds them), one pass; SURVEY 8-d4's 64 songs "
                                       "would take ~4x as long for the same rate (data-independent dense LAPACK per song)",
                             "seconds": dt_cpu, "per_song_seconds": spread(per_song)},
            "roofline": {"kernel": "gemm_f64_kernel<64> (W = Dm U, 14 problems of 768 x 768 x 768 against the upper-triangular half of Sigma_b; whole batched call timed)",
                         "bound": "fp64 mfma", "achieved": flops / dt5 / 1e12,
                         "peak": 78.6, "unit": "TFLOP/s", "frac": flops / dt5 / 1e12 / 78.6,
                         "note": "flops ISSUED = 2 rows D^2 (t+1)/(2t), t = D/64 (d^T Sigma d = 2 d^T U d skips the zero half: 13/24 of 2 n_songs D^2 "
                                 "at D = 768); peak = fp64 matrix datasheet figure (the guide lists none); measured v_mfma_f64_16x16x4 ceiling on "
                                 "this chip 45-47 TFLOP/s (scripts/probes/mfma_rate.hip); the whole call is timed (pair_stats_diff, the row dots, two "
                                 "small kernels and the host's offsets-up / scores-down round trip included; rows and baseline resident in HBM)"}}


def _song_chain_roofline(nsongs, frames, d, seconds, iters):
    """Matrix-pipe work ISSUED by the batched per-song chain over the whole call time: covariances (float16 MFMA: x' x' + x' e + e x' on the
    upper-triangular 128 x 128 tiles), the two exact products (30 + 26 int8 digit pairs), `iters` split-float16 iterations of three
    products x three MFMA terms.  The call also holds the statistics pass, digit planes and the host's decision: `frac` prices ALL of its time
    against the time the issued matrix work needs at the dense peaks."""
    nt = -(-d // 128)
    cov_flops = nsongs * 3.0 * 2.0 * frames * 128 * 128 * (nt * (nt - 1) // 2 + nt * 20.0 / 32.0)
    it_flops = nsongs * (1 + 3 * (iters - 1)) * 3 * 2.0 * d ** 3
    i8_ops = nsongs * (30 + 26) * 2.0 * d ** 3
    ideal = (cov_flops + it_flops) / (MFMA_F16_PEAK_TFLOPS * 1e12) + i8_ops / (2 * MFMA_F16_PEAK_TFLOPS * 1e12)
    return {"bound": "mfma", "achieved": (cov_flops + it_flops + i8_ops) / seconds / 1e12, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "T(FL)OP/s issued (f16 + int8)",
            "frac": ideal / seconds, "issued": {"f16_cov_flops": cov_flops, "f16_iteration_flops": it_flops, "i8_ops": i8_ops, "iterations_assumed": iters},
            "note": "frac = time the issued MFMA work needs at the dense peaks (f16 2.5 PFLOP/s, int8 5 POP/s) / whole call time; iterations as measured "
                    "for these songs (condition numbers of a few thousand: 6-8 with the scaled steps of round 4, 7-12 before)"}


def extra_c5_frames(torch, hip, device):
    """Config 5, encoder-frame variant: songs of [1500 x 768] float16 frames (Whisper-small's encoder output per
    clip) against a baseline from a synthetic [20000 x 768]; every song is a full D x D problem (n - 1 >= D).  Five calls timed,
    median; CPU baseline = 4 of the songs through the oracle, one after the other."""
    from oracle import fad_oracle as O
    nsongs, frames, d5 = 32, 1500, 768
    g5 = torch.Generator(device=device); g5.manual_seed(55)
    scale = 0.5 + torch.rand((d5,), generator=g5, device=device)
    songs = (torch.randn((nsongs * frames, d5), generator=g5, device=device) * scale).to(torch.float16)
    base = torch.randn((20000, d5), generator=g5, device=device, dtype=torch.float64) * scale.double() * 1.05 + 0.01
    mu5 = base.mean(0).cpu().numpy(); cov5 = torch.cov(base.T).cpu().numpy()
    offs = np.arange(0, nsongs * frames + 1, frames)
    for _ in range(2):                                                   # (allocations and first-use kernel loads stay out of the timed calls)
        hip.frechet_batched(mu5, cov5, songs, offs)
    ms = []
    for _ in range(5):
        torch.cuda.synchronize(); t5 = time.perf_counter()
        sc5, st5 = hip.frechet_batched(mu5, cov5, songs, offs)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t5) * 1e3)
    dt5 = float(np.median(ms)) * 1e-3
    n_cpu = 4
    sample = songs[: n_cpu * frames].cpu().numpy()
    t0 = time.perf_counter()
    want = [O.individual_scores(mu5, cov5, [sample[i * frames:(i + 1) * frames]], run_sqrtm=True)[0] for i in range(n_cpu)]
    dt_cpu = time.perf_counter() - t0
    want = np.array([np.nan if w is None else float(w) for w in want])
    rel = float(np.nanmax(np.abs(sc5[:n_cpu] - want) / np.abs(want)))
    return {"songs": nsongs, "dim": d5, "frames_per_song": frames, "ms": dt5 * 1e3, "ms_spread": spread(ms), "songs_per_s": nsongs / dt5,
            "ok": int((st5 == 0).sum()), "max_rel_err_vs_oracle_sample": rel,
            "roofline": _song_chain_roofline(nsongs, frames, d5, dt5, iters=8),
            "cpu_baseline": {"value": n_cpu / dt_cpu, "unit": "songs/s", "cores": "BLAS threads as numpy finds them", "kind": "port",
                             "sample": f"{n_cpu} of the same songs through the oracle (np.cov + eig + sqrtm per song, fad.py:373-378), "
                                       "one after the other", "seconds": dt_cpu},
            "note": "per call: one-pass float64 statistics, covariances on the float16 tile kernel (shifted by the song's mean), A = Sigma_b Sigma_s "
                    "exact on the int8 MFMA, split-float16 Newton-Schulz on 128 x 128 tiles (one XCD per song, 8-12 iterations), digits of "
                    "the final Y, exact correction, decision per song on the host; a song the chain does not accept goes to the float64 routes"}


def extra_c4_songs(torch, hip, device):
    """Config-4 shape per song (Encodec: D = 128, 30 s at 75 frames/s): 2000 songs of [2250 x 128] float16 frames against one baseline,
    one batched call (five timed, median); every song is a full 128 x 128 problem.  CPU baseline = 16 of the songs through the oracle."""
    from oracle import fad_oracle as O
    nsongs, frames, d4 = 2000, 2250, 128
    g4 = torch.Generator(device=device); g4.manual_seed(44)
    scale = 0.6 + 0.8 * torch.rand((d4,), generator=g4, device=device)
    songs = (torch.randn((nsongs * frames, d4), generator=g4, device=device) * scale).to(torch.float16)
    base = torch.randn((50000, d4), generator=g4, device=device, dtype=torch.float64) * scale.double() * 1.03 + 0.02
    mu4 = base.mean(0).cpu().numpy(); cov4 = torch.cov(base.T).cpu().numpy()
    offs = np.arange(0, nsongs * frames + 1, frames)
    for _ in range(2):                                                   # (first-call allocations measured 68 ms inside a timed run in round 3)
        hip.frechet_batched(mu4, cov4, songs, offs)
    ms = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sc, stt = hip.frechet_batched(mu4, cov4, songs, offs)
        torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) * 1e3)
    dt = float(np.median(ms)) * 1e-3
    n_cpu = 16
    sample = songs[: n_cpu * frames].cpu().numpy()
    t0 = time.perf_counter()
    want = [O.individual_scores(mu4, cov4, [sample[i * frames:(i + 1) * frames]], run_sqrtm=True)[0] for i in range(n_cpu)]
    dt_cpu = time.perf_counter() - t0
    want = np.array([np.nan if w is None else float(w) for w in want])
    rel = float(np.nanmax(np.abs(sc[:n_cpu] - want) / np.abs(want)))
    return {"songs": nsongs, "dim": d4, "frames_per_song": frames, "ms": dt * 1e3, "ms_spread": spread(ms), "songs_per_s": nsongs / dt,
            "ok": int((stt == 0).sum()), "max_rel_err_vs_oracle_sample": rel,
            "GBps_frames": songs.numel() * 2 / dt / 1e9,
            "roofline": _song_chain_roofline(nsongs, frames, d4, dt, iters=6),
            "note": "as per_song_config5_encoder_frames, but D = 128: the whole Newton-Schulz iteration of a song runs in ONE workgroup, iterates in "
                    "LDS and registers (ns_fast_res.h)",
            "cpu_baseline": {"value": n_cpu / dt_cpu, "unit": "songs/s", "cores": "BLAS threads as numpy finds them", "kind": "port",
                             "sample": f"{n_cpu} of the same songs through the oracle (np.cov + eig + sqrtm per song), one after the other",
                             "seconds": dt_cpu}}


def extra_score_inf(fadtk_amd, a_host, b_host):
    """FAD-inf at config-3 size (fad.py:304-351): 25 resampled sizes of the [100000 x 512] float16 eval set against the
    baseline's statistics.  Batched device route (frames in HBM, eight resamples per moments launch, square-root chains in
    flight) vs the point-by-point route through the public functions (host gather, PCIe per point: what round 2 shipped),
    and 2 of the 25 points through the CPU oracle."""
    from oracle import fad_oracle as O

    class _M:
        name = "bench"
    fad = fadtk_amd.FrechetAudioDistance(_M(), load_model=False)
    mu_b, cov_b = fadtk_amd.calc_embd_statistics(a_host)
    mu_b = mu_b.astype(np.float64)
    ns = [int(n) for n in np.linspace(500, b_host.shape[0], 25)]
    rng = np.random.default_rng(25)
    picks = [rng.integers(0, b_host.shape[0], size=n) for n in ns]
    fad._score_inf_points_on_device(mu_b, cov_b, b_host, picks[:3])        # warm-up (allocations)
    ms_dev, vals = [], None
    for _ in range(3):
        t0 = time.perf_counter(); vals = fad._score_inf_points_on_device(mu_b, cov_b, b_host, picks); ms_dev.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); seq = fad._score_inf_points_sequential(mu_b, cov_b, b_host, picks); ms_seq = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    want = []
    for k in (0, 2, 12):
        mu_e, cov_e = O.embd_statistics(b_host[picks[k]])
        want.append(O.frechet_distance(mu_b, cov_b, mu_e, cov_e, run_sqrtm=True))
    dt_cpu = time.perf_counter() - t0
    rel = max(abs(vals[k] - w) / abs(w) for k, w in zip((0, 2, 12), want))
    xs = 1.0 / np.array(ns)
    slope, intercept = np.polyfit(xs, np.array(vals), 1)
    return {"points": 25, "rows": int(b_host.shape[0]), "dim": int(b_host.shape[1]), "resampled_rows_total": int(sum(ns)),
            "ms_batched_device_route": float(np.median(ms_dev)), "ms_spread": spread(ms_dev), "points_per_s": 25e3 / float(np.median(ms_dev)),
            "ms_point_by_point_route": ms_seq, "speedup_vs_point_by_point": ms_seq / float(np.median(ms_dev)),
            "max_rel_diff_between_routes": float(np.max(np.abs(np.array(vals) - np.array(seq)) / np.abs(np.array(seq)))),
            "fad_inf": float(intercept), "max_rel_err_vs_oracle_sample": float(rel),
            "cpu_baseline": {"value": 3.0 / dt_cpu, "unit": "points/s", "cores": "BLAS threads as numpy finds them", "kind": "port",
                             "sample": f"3 of the 25 points (n = {ns[0]} -- fewer rows than dimensions --, {ns[2]} and {ns[12]} resampled rows) through the oracle: fancy-index gather, np.cov, eig + sqrtm",
                             "seconds": dt_cpu},
            "note": "includes the upload of the [100000 x 512] eval frames (102 MB) once per call and the 25 index vectors"}


def extra_decaying(torch, hip, device):
    """What a pair with a DECAYING spectrum costs (covariances of real embeddings decay; config 3's iid recipe is the flat best case):
    D = 512, N = 100000 per set, Sigma ~ k^-p for p = 0.5, 1, 2 (both sets share the eigenvectors, scripts/probe_illcond.py), so
    Sigma_1 Sigma_2 ~ k^-2p.  Per spectrum: route, iterations, blocking fad_frechet_from_moments time (median of 5) and the relative
    difference to the CPU oracle (eig + sqrtm, fad.py:88-92) on the same float16 frames."""
    from oracle import fad_oracle as O
    d, n = 512, N_ROWS
    out = {}
    g = torch.Generator(device=device); g.manual_seed(77)
    q, _ = torch.linalg.qr(torch.randn((d, d), generator=g, device=device, dtype=torch.float64))
    for p in (0.5, 1.0, 2.0):
        lam = torch.arange(1, d + 1, device=device, dtype=torch.float64) ** (-p / 2.0)
        a = ((torch.randn((n, d), generator=g, device=device, dtype=torch.float64) * lam) @ q.T).to(torch.float16)
        b = ((1.05 * torch.randn((n, d), generator=g, device=device, dtype=torch.float64) * lam) @ q.T + 0.01).to(torch.float16)
        with hip.Moments(d) as ma, hip.Moments(d) as mb:
            hip.Moments.update_multi([ma, mb], [a, b])
            for _ in range(2):
                fad, diag = hip.frechet_from_moments(ma, mb, mean_dtype=FAD_F16)
            ms = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                fad, diag = hip.frechet_from_moments(ma, mb, mean_dtype=FAD_F16)
                ms.append((time.perf_counter() - t0) * 1e3)
            # the same pair sixteen times over as ONE batch (fad_frechet_from_moments_multi_begin): what a score of this kind costs among its like
            batch = [(ma, mb)] * 16
            for _ in range(2):
                res = hip.FrechetMultiJob(batch, mean_dtype=FAD_F16).result()
            msb = []
            for _ in range(5):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                res = hip.FrechetMultiJob(batch, mean_dtype=FAD_F16).result()
                msb.append((time.perf_counter() - t0) * 1e3)
            fb, db = res[-1]
        t0 = time.perf_counter()
        ref = float(O.fad_between(a.cpu().numpy(), b.cpu().numpy()))
        out[f"k^-{p:g}"] = {"ms": float(np.median(ms)), "ms_spread": spread(ms), "route": int(diag.get("route", 0)) if diag["converged"] == 3 else 0,
                            "converged": int(diag["converged"]), "iterations": int(diag["iters"]), "fad": float(fad),
                            "rel_err_vs_oracle": abs(float(fad) - ref) / abs(ref),
                            "ms_per_score_in_a_batch_of_16": float(np.median(msb)) / 16.0, "batch_ms_spread": spread(msb),
                            "batch_route": int(db.get("route", 0)) if db["converged"] == 3 else 0, "batch_iterations": int(db["iters"]),
                            "batch_rel_err_vs_oracle": abs(float(fb) - ref) / abs(ref), "oracle_seconds": time.perf_counter() - t0}
        del a, b
        cpu_quiet()                 # (the oracle's BLAS burst must not reach into the next spectrum's timings)
    out["note"] = ("covariance spectra k^-p of both sets (product k^-2p); route 2 = eight-launch split-float16 chain, 1 = float32 chain, 0 = float64 "
                   "Newton-Schulz (scaled steps while the tracked lower bound of the spectrum is below 0.9)")
    return out


def extra_host(fadtk_amd, a_host, b_host, fad_ref):
    """SURVEY 8-d3 'with H2D copy': the same score from PAGEABLE numpy arrays through the reference's own call sequence --
    calc_embd_statistics(A), calc_embd_statistics(B), calc_frechet_distance (fad.py:42-120) -- i.e. 2 x 102.4 MB over PCIe inside
    the library (pinned, pipelined chunks: csrc/host_stage.cpp), (mu, Sigma) back to the host, both Sigma up again.  Never `value`."""
    fadtk_amd.calc_embd_statistics(a_host[:4096])
    ms, parts, f = [], [], None
    for _ in range(6):
        t0 = time.perf_counter(); m1, c1 = fadtk_amd.calc_embd_statistics(a_host); t1 = time.perf_counter()
        m2, c2 = fadtk_amd.calc_embd_statistics(b_host); t2 = time.perf_counter()
        f = float(fadtk_amd.calc_frechet_distance(m1, c1, m2, c2)); t3 = time.perf_counter()
        ms.append((t3 - t0) * 1e3); parts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    ms, parts = ms[1:], np.array(parts[1:])
    med = float(np.median(ms))
    set_ms = float(np.median(parts[:, :2]))
    # the FIRST call of a fresh process beside the warm one: what a one-shot CLI run pays -- the library's code objects
    # (one per translation unit, ~75 ms each the first time a device is used: csrc/common.cpp warm_code_objects), the handle, its
    # workspaces, the pinned staging area; then three more calls in that process
    cold = None
    try:
        import subprocess
        code = ("import time, sys, numpy as np; sys.path.insert(0, %r); import fadtk_amd; "
                "a = np.random.default_rng(3).standard_normal((100000, 512), dtype=np.float32).astype(np.float16); ts = []\n"
                "for _ in range(4):\n    t0 = time.perf_counter(); fadtk_amd.calc_embd_statistics(a); ts.append((time.perf_counter() - t0) * 1e3)\n"
                "print('COLD', *ts)") % str(ROOT)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=180)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("COLD")]
        if line:
            v = [float(x) for x in line[0].split()[1:]]
            cold = {"first_call_ms": v[0], "second_call_ms": v[1], "fourth_call_ms": v[3],
                    "note": "calc_embd_statistics([100000 x 512] float16, pageable) in a process of its own: the first call loads the library's code objects, "
                            "creates the handle and its workspaces and pins the staging area; nothing of it is paid again"}
    except Exception as e:      # noqa: BLE001
        cold = {"error": repr(e)}
    return {"scores_per_s": 1e3 / med, "ms_per_score": med, "ms_spread": spread(ms), "fresh_process": cold,
            "ms_statistics_per_set": set_ms, "ms_frechet": float(np.median(parts[:, 2])),
            "h2d_GBps_per_set": a_host.nbytes / (set_ms * 1e-3) / 1e9,
            "fad": f, "rel_err_vs_device_resident_path": abs(f - fad_ref) / abs(fad_ref),
            "note": "inputs: pageable numpy float16 [100000 x 512] x 2; per set: H2D + moments + finalize + D2H of (mu, Sigma); "
                    "h2d_GBps_per_set divides the set's bytes by the WHOLE calc_embd_statistics call"}


def torchrun_command(gpus, argv):
    """`python bench.py --gpus N ...` from a plain shell: the command that starts the N ranks (one per GPU, RCCL), exactly as
    fadtk_amd/cli.py:_relaunch does for the product's --gpus (reference: the spawn pool of fadtk/fad_batch.py:43-48)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", os.environ.get("MASTER_PORT", "29541"), str(Path(__file__).resolve()), *argv]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--inflight", type=int, default=DEFAULT_INFLIGHT,
                    help="scores in flight: consecutive steps alternate over this many pairs of accumulators (each with its own "
                         "Frechet job); the moments of step i and the Frechet chain of step i-1 are enqueued before the score of "
                         "step i-N is collected (3 or more keep the device busy); 1 = every step waits for its score")
    ap.add_argument("--single-stream", action="store_true",
                    help="all scores in flight on ONE stream: no two kernels ever overlap.  The default of the batched schedule (the tile "
                         "kernel's duration inside the timed loop is then the kernel alone: what `roofline` reports); for --batch 0 it "
                         "replaces one HIP stream per score in flight")
    ap.add_argument("--multi-stream", action="store_true",
                    help="batched schedule: one HIP stream per batch in flight (three): ~10 %% more scores/s -- the idle CUs of one "
                         "batch's small launches take another batch's kernels -- but a dispatch's duration then includes the time it "
                         "waits for another stream's workgroups to leave the CUs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--realistic-only", action="store_true",
                    help="profiling aid: the timed loop and the `value_realistic` block, nothing else (implies --timed-only for the other side blocks)")
    ap.add_argument("--no-realistic", action="store_true", help="skip the `value_realistic` block (k^-1 spectra, offset frames, reference-order means)")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed side measurements (used under rocprofv3 "
                                                             "so that the kernel statistics hold the config-3 launches only)")
    ap.add_argument("--timed-only", action="store_true",
                    help="warm-up + the timed loop and nothing else (no repeat / same-pair / other-layout / breakdown blocks, no extras, "
                         "no CPU baseline): the command every rocprofv3 pass under profiles/ runs, so that its kernel statistics hold the "
                         "timed layout's launches only")
    ap.add_argument("--group", type=int, default=0,
                    help="grouped schedule: the moments of G consecutive steps back to back on ONE stream, then their G square-root chains "
                         "side by side on G streams (two groups in flight: 2 G scores); 0 = the lane schedule (one stream per score)")
    ap.add_argument("--batch", type=int, default=16,
                    help="batched chains (the default schedule): the moments of B consecutive steps, then ONE square-root chain for the B "
                         "scores (fad_frechet_from_moments_multi_begin: nine launches carry all B); three such batches in flight, one stream "
                         "each; 0 = the lane schedule of round 3 (one stream and one chain per score, --inflight of them)")
    ap.add_argument("--moments-group", type=int, default=16,
                    help="batched schedule: the moments of this many consecutive steps in ONE launch of the tile kernel and ONE reduce "
                         "(fad_moments_update_multi over 2 x M frame matrices, 32 at most: a batch of 16 steps is one launch); 1 = one launch per step")
    ap.add_argument("--moments-stream", action="store_true",
                    help="batched schedule: every moments launch on ONE stream of its own, the chains on the batch streams (experiment)")
    ap.add_argument("--chain-cus", type=int, default=0,
                    help="experiment (profiles/r04*_streams.txt): confine the square-root chains to this many CUs per XCD (CU-masked "
                         "streams) and the moments kernels to the others; 0 = no masks")
    args = ap.parse_args()
    if int(args.batch) > 0 and not int(args.group) and not args.multi_stream:
        args.single_stream = True                                        # the batched schedule's default layout
    args.lane_streams = not args.single_stream
    if args.realistic_only:
        args.timed_only = True
    if args.timed_only:
        args.no_extras = True; args.no_cpu_baseline = True
    if (args.gpus > 1 or os.environ.get("FAD_BENCH_FORCE_DIST") == "1") and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # started from a plain shell: launch the ranks ourselves (the driver's N > 1 command goes through torch.distributed.run and
        # arrives here with WORLD_SIZE set)
        import subprocess
        cmd = torchrun_command(max(args.gpus, 1), sys.argv[1:])
        if os.environ.get("FAD_BENCH_PRINT_LAUNCH") == "1":      # tests: show the command instead of running it
            print(json.dumps(cmd))
            return
        raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))

    # Library banners (RCCL prints its version block to stdout) must not mix with the ONE JSON line: everything
    # written to fd 1 goes to stderr until the result is printed.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # before the HIP runtime starts: dmabuf IPC only
    if args.chain_cus > 0:                                               # before any handle is created (the knob is read there)
        os.environ["FAD_MOMENTS_CUS"] = str(8 * (32 - args.chain_cus))
    import torch
    import torch.distributed as dist
    from fadtk_amd import dist as fdist, hip

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch as `python bench.py --gpus {args.gpus}` (it starts the "
                         f"ranks itself) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    # FAD_BENCH_FORCE_DIST=1: take the multi-rank code path (process group, packed all-reduce) even with one rank,
    # so that it can be exercised on a single-GPU box
    distributed = world > 1 or os.environ.get("FAD_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", device_id=device, rank=rank, world_size=world)

    pairs = [make_sets(torch, device, rank, k) for k in range(N_PAIRS)]
    a, b = pairs[0]
    # both handles of a score keep their statistics in ONE device buffer: the exchange of the path -- the sum of the ranks'
    # sufficient statistics -- is a single in-place all-reduce over it (the product's --gpus path uses the same class).
    # One such buffer per score in flight ("lane"), and one HIP stream per lane: the host enqueues steps ahead of the score it
    # collects (run_steps), so the device never waits for the host between steps, and the latency-bound square-root chain of one
    # score (eight dependent launches, the matrix pipes idle most of the time) runs UNDER the moments kernel of the next -- 6787
    # instead of 5618 scores/s (profiles/r03r_streams.txt).  The price is in the per-kernel numbers: the tile kernel shares the
    # CUs with another score's chain and its HIP-event duration grows from 0.083 to 0.104 ms -- `roofline` reports that (it is
    # what the timed region ran) and `roofline.alone` the duration on a stream of its own (--single-stream times that loop).
    # (Tried and dropped: moments on a high-priority stream, chains on a low-priority one -- 5511 scores/s.)
    n_lanes = max(1, min(int(args.inflight), 8))

    comm_stream = torch.cuda.Stream(device=device) if distributed else None

    masked = args.chain_cus > 0 and args.lane_streams and n_lanes > 1

    class Lane:
        def __init__(self, k, own=None, mstream=None, shared=None):
            own = (args.lane_streams and n_lanes > 1) if own is None else own
            self.stream = torch.cuda.Stream(device=device) if own else torch.cuda.current_stream(device)
            self.cstream = self.stream                                   # where the square-root chain goes
            if mstream is not None:                                      # grouped schedule: shared moments stream, own chain stream
                self.cstream, self.stream = self.stream, mstream
            self.group_ev = None
            if masked and own:
                # --chain-cus C: moments on CUs [C, 32) of every XCD, chains on CUs [0, C) (FAD_MOMENTS_CUS tells the planner)
                self.stream = hip.cu_masked_stream(range(args.chain_cus, 32), local_rank)
                self.cstream = hip.cu_masked_stream(range(0, args.chain_cus), local_rank)
            if shared is not None:                                       # (a slot of a group's buffer: --moments-group)
                self.shared, slot = shared
                self.ma, self.mb = self.shared.moments[2 * slot], self.shared.moments[2 * slot + 1]
            else:
                self.shared = fdist.SharedStats(DIM, SETS, local_rank)
                self.ma, self.mb = self.shared.moments
            self.job = None
            self.fed = torch.cuda.Event(); self.reduced = torch.cuda.Event()

        def feed(self, pair):
            """Phase 1 of a step: moments of both sets (one launch of each kernel); with several ranks the exchange -- ONE in-place
            all-reduce over the buffer that holds both sets' statistics -- starts on a stream of its own as soon as they are
            there, so that it runs under the NEXT step's moments instead of in front of this step's square root."""
            with torch.cuda.stream(self.stream):
                self.ma.reset(); self.mb.reset()
                hip.Moments.update_multi([self.ma, self.mb], list(pair))
                if distributed:
                    self.fed.record()
                    with torch.cuda.stream(comm_stream):
                        comm_stream.wait_event(self.fed)
                        dist.all_reduce(self.shared.buffer)              # (SharedStats.allreduce minus the settle calls: both sets were just fed)
                        self.reduced.record()

        def score(self):
            """Phase 2: [wait for the exchange,] enqueue the whole Frechet chain; nothing is waited for on the host."""
            if self.cstream is not self.stream and not distributed and self.group_ev is None:
                with torch.cuda.stream(self.stream):
                    self.fed.record()
            with torch.cuda.stream(self.cstream):
                if self.group_ev is not None:
                    torch.cuda.current_stream().wait_event(self.group_ev)
                if distributed:
                    torch.cuda.current_stream().wait_event(self.reduced)
                elif self.cstream is not self.stream and self.group_ev is None:
                    torch.cuda.current_stream().wait_event(self.fed)
                self.job = hip.FrechetJob(self.ma, self.mb, mean_dtype=FAD_F16)

        def collect(self):
            job, self.job = self.job, None
            return job.result()

    G = max(0, min(int(args.group), 4))
    if G:
        NGRP = 3 if 3 * G <= 8 else 2                                     # groups in flight (8 score slots per thread)
        n_lanes = NGRP * G
        mstream = torch.cuda.Stream(device=device)
        lanes = all_lanes = [Lane(k, own=True, mstream=mstream) for k in range(n_lanes)]
    else:
        lanes = all_lanes = [Lane(k) for k in range(n_lanes)]
    plen = lanes[0].ma.packed_len
    ma, mb = lanes[0].ma, lanes[0].mb

    host_s = [0.0, 0.0, 0.0]            # host seconds spent enqueueing moments / enqueueing chains / waiting for + collecting scores

    def run_steps_lanes(count, marks=None, rotate=True, lanes=None):
        lanes = all_lanes if lanes is None else lanes
        n_lanes = len(lanes)
        """`count` steps, at most n_lanes of them in flight; every one of them is collected before this returns.  Order of the
        enqueues: feed(i), score(i-1), so the device sees  moments(i) | Frechet(i-1) | moments(i+1) | Frechet(i) ...  and the
        host collects score(i - n_lanes) before it reuses that lane -- with three lanes two more steps are queued behind the
        one it waits for, so the device never runs dry."""
        out, fed = None, None
        pc = time.perf_counter

        def collect(lane):
            nonlocal out
            t = pc()
            out = lane.collect()
            host_s[2] += pc() - t
            if marks is not None:
                marks.append(time.perf_counter())

        for i in range(count):
            lane = lanes[i % n_lanes]
            if lane.job is not None:
                collect(lane)
            t = pc()
            lane.feed(pairs[i % N_PAIRS] if rotate else pairs[0])       # consecutive steps stream DIFFERENT frames from HBM
            host_s[0] += pc() - t
            t = pc()
            if n_lanes == 1:
                lane.score()
            else:
                if fed is not None:
                    fed.score()
                fed = lane
            host_s[1] += pc() - t
        if fed is not None:
            fed.score()
        for k in range(n_lanes):                                         # drain, oldest first
            lane = lanes[(count + k) % n_lanes]
            if lane.job is not None:
                collect(lane)
        return out

    def run_steps_grouped(count, marks=None, rotate=True, lanes=None):
        """Grouped schedule (--group G): moments of G steps on the shared moments stream, one event behind the last of them, then the G
        chains on their own streams -- latency-bound launches of G different scores side by side instead of under a tile kernel that
        leaves them no room on a CU.  Two groups are in flight; a group's scores are collected before its lanes are fed again."""
        lanes = all_lanes if lanes is None else lanes
        out = None
        pc = time.perf_counter
        i = 0
        g = 0
        while i < count:
            grp = lanes[(g % NGRP) * G:(g % NGRP) * G + G]
            m = min(G, count - i)
            for lane in grp[:m]:
                if lane.job is not None:
                    t = pc(); out = lane.collect(); host_s[2] += pc() - t
                    if marks is not None:
                        marks.append(pc())
            t = pc()
            for k, lane in enumerate(grp[:m]):
                lane.feed(pairs[(i + k) % N_PAIRS] if rotate else pairs[0])
            ev = torch.cuda.Event()
            with torch.cuda.stream(grp[0].stream):
                ev.record()
            host_s[0] += pc() - t
            t = pc()
            for lane in grp[:m]:
                lane.group_ev = ev
                lane.score()
            host_s[1] += pc() - t
            i += m; g += 1
        for gg in range(g, g + NGRP):                                    # drain, oldest group first
            for lane in lanes[(gg % NGRP) * G:(gg % NGRP) * G + G]:
                if lane.job is not None:
                    t = pc(); out = lane.collect(); host_s[2] += pc() - t
                    if marks is not None:
                        marks.append(pc())
        return out

    run_steps = run_steps_lanes
    if G:
        run_steps = run_steps_grouped

    BATCH = 0 if G else max(0, min(int(args.batch), 32))
    MG = max(1, min(int(args.moments_group), 16, BATCH)) if BATCH else 1         # steps per moments launch (2 sets each, 32 at most)
    if BATCH:
        NB_FLY = 3
        # (--single-stream: all batches on ONE stream -- no two kernels ever overlap, the tile kernel's HIP-event time is the kernel alone)
        bstreams = [torch.cuda.current_stream(device)] * NB_FLY if args.single_stream else [torch.cuda.Stream(device=device) for _ in range(NB_FLY)]
        # the statistics of the steps that share a moments launch live in ONE device buffer (2 M packed accumulators): with several
        # ranks their exchange is ONE in-place all-reduce per launch (16.8 MB at M = 4) instead of one per step
        # (a batch slot holds up to LCAP pairs: the LAST batch of a call takes the steps that are left when they are no more than that -- 20 steps
        #  are one batch of 20, not 16 + 4: a chain of 4 pairs costs what a chain of 16 does)
        LCAP = min(32, (3 * BATCH) // 2)
        gshared = [[fdist.SharedStats(DIM, SETS * MG, local_rank) for _ in range(-(-LCAP // MG))] for _ in range(NB_FLY)] if MG > 1 else None
        blanes = [[Lane(k, own=False, shared=((gshared[q][k // MG], k % MG) if MG > 1 else None)) for k in range(LCAP)] for q in range(NB_FLY)]
        for q in range(NB_FLY):
            for ln in blanes[q]:
                ln.stream = ln.cstream = bstreams[q]
        bjobs = [None] * NB_FLY
        # --moments-stream: ALL moments launches on one stream of their own (no two tile kernels ever compete for the CUs), the chains on
        # the batch streams behind an event
        mstream_b = torch.cuda.Stream(device=device) if (args.moments_stream and not args.single_stream) else None
        bdone = [torch.cuda.Event() for _ in range(NB_FLY)]
        MGv = [MG]                          # (a side block below re-runs the loop with one launch per step)
        launch_sets = {}                    # id(leader handle) -> frame matrices of every launch recorded on it (cleared where timing starts)
        prepared = {}                       # (slot, first step of the group, steps, pair phase, which pairs) -> hip.PreparedMultiUpdate

        def run_steps_batched(count, marks=None, rotate=True, lanes=None):
            """Batched schedule (--batch B): per batch, the moments of B steps and then ONE chain for their B scores on the batch's
            stream; NB_FLY batches in flight.  Every score is collected inside the call."""
            out = None
            pc = time.perf_counter

            def collect(q):
                nonlocal out
                t = pc()
                vals, diags = bjobs[q][0].result_arrays()               # (every score of the batch; the diagnostics stay ctypes records)
                host_s[2] += pc() - t
                out = (float(vals[-1]), diags[len(vals) - 1].as_dict())
                if marks is not None:
                    now = pc()
                    marks.extend([now] * bjobs[q][1])
                bjobs[q] = None

            i, b = 0, 0
            while i < count:
                q = b % NB_FLY
                if bjobs[q] is not None:
                    collect(q)
                m = (count - i) if (count - i) <= LCAP else BATCH
                t = pc()
                MG = MGv[0]
                if MG > 1:
                    # the moments of MG steps in ONE launch of each kernel (fad_moments_update_multi takes up to 32 frame matrices): the
                    # 256 workgroups of the tile kernel then hold MG x longer row ranges -- one set of partial tiles per LAUNCH, not per
                    # step -- and the reduce runs once (scripts/probe_sets.py: 108 / 75 / 70 / 68 us per pair at 1 / 2 / 3 / 4 pairs)
                    for k0 in range(0, m, MG):
                        grp = blanes[q][k0:min(k0 + MG, m)]
                        launch_sets.setdefault(id(grp[0].ma), []).append(2 * len(grp))
                        with torch.cuda.stream(mstream_b if mstream_b is not None else bstreams[q]):
                            # the same handles fed from the same resident tensors as the last time this slot came round: the call's
                            # tables are built once (hip.PreparedMultiUpdate), feeding a launch is two calls into the library
                            key = (q, k0, len(grp), (i + k0) % N_PAIRS if rotate else -1, id(pairs[0][0]))
                            pu = prepared.get(key)
                            if pu is None:
                                pu = prepared[key] = hip.Moments.prepared_update_multi(
                                    [h for ln in grp for h in (ln.ma, ln.mb)],
                                    [x for kk in range(len(grp)) for x in (pairs[(i + k0 + kk) % N_PAIRS] if rotate else pairs[0])])
                            pu.run(reset=True)
                            if distributed:
                                grp[0].fed.record()
                                with torch.cuda.stream(comm_stream):
                                    comm_stream.wait_event(grp[0].fed)
                                    gs = grp[0].shared
                                    dist.all_reduce(gs.buffer[: 2 * len(grp) * gs.stride])        # the group's statistics: one collective
                                    for ln in grp:
                                        ln.reduced.record()
                else:
                    for k in range(m):
                        blanes[q][k].feed(pairs[(i + k) % N_PAIRS] if rotate else pairs[0])
                host_s[0] += pc() - t
                t = pc()
                if mstream_b is not None and MG > 1:                     # (--moments-stream: the chain waits for the batch's moments)
                    bdone[q].record(mstream_b)
                with torch.cuda.stream(bstreams[q]):
                    if mstream_b is not None and MG > 1:
                        torch.cuda.current_stream().wait_event(bdone[q])
                    if distributed:
                        for k in range(m):
                            torch.cuda.current_stream().wait_event(blanes[q][k].reduced)
                    bjobs[q] = (hip.FrechetMultiJob([(blanes[q][k].ma, blanes[q][k].mb) for k in range(m)], mean_dtype=FAD_F16), m)
                host_s[1] += pc() - t
                i += m; b += 1
            for qq in range(b, b + NB_FLY):
                if bjobs[qq % NB_FLY] is not None:
                    collect(qq % NB_FLY)
            return out

        run_steps = run_steps_batched
        lanes = blanes[0]

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    # Python's cyclic collector stays out of the measured blocks, as `timeit` keeps it out: a generation-2 pass over the objects of torch +
    # numpy + scipy takes ~70 ms here -- 30 x the timed region -- and falls wherever the allocation count says (r05p: a repeat block of 20
    # steps at 283 scores/s; the "30-70 ms stalls of single blocking calls" that no HIP trace ever showed: DESIGN.md 6.4).  Nothing in the
    # loops below creates reference cycles; the collector is switched on again before the line is printed.
    import gc
    gc.collect(); gc.disable()
    if os.environ.get("FAD_BENCH_PREWARM") == "1":       # diagnosis only: a block of K untimed steps in front of the warm-up
        run_steps(args.steps)
    if BATCH:
        # set-up, not warm-up: every one of the three job slots allocates its batch workspace (8 x 28 MB) on first use, and a warm-up of
        # W < 3 B steps would leave that to the timed region (measured: 0.77 instead of 0.12 ms per step at W = 3).  One full round here.
        run_steps(3 * BATCH)
        run_steps(LCAP)         # (... and the first slot the room of the largest batch a call can form: 1.3 GB for more than 16 pairs -- r05w: left
                                #  to the timed region, 8 740-8 980 scores/s where the repeats of the same 20 steps gave 10 100-10 250)
    run_steps(max(args.warmup, 0))
    # The tile kernel is timed by HIP events the library records around it on the launch stream -- on ONE lane (every
    # n_lanes-th step of the timed region) and without the third event behind the reduce: a timed event record between two
    # kernels costs the stream a few microseconds (scripts/probe_graph.py: 173 -> 182 -> 185 us per step with none / two /
    # three events on every step), and the benchmark should not throttle what it measures.
    # (batched schedule: two lanes of every batch stream are sampled -- the first and the middle update of a batch -- so that the
    # average covers launches that run beside another batch's chain as well as those that do not)
    # (... with --moments-group M > 1: every launch -- the leader of each group of M steps carries the events)
    sampled_slots = (sorted({0, BATCH // 2}) if MG == 1 else list(range(0, BATCH, MG))) if BATCH else []
    timed_handles = ([blanes[q][k].ma for q in range(3) for k in sampled_slots] if BATCH else [lanes[0].ma])
    for hnd in timed_handles:
        hnd.set_timing(2)
    if BATCH:
        launch_sets.clear()
    fence()
    marks = [time.perf_counter()]
    host_s[:] = [0.0, 0.0, 0.0]
    fad, diag = run_steps(args.steps, marks)                             # every score is delivered inside the timed region
    host_timed = list(host_s)
    fence()
    elapsed = time.perf_counter() - marks[0]
    per_rank_s, allreduce_alone = None, None
    if distributed:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_s = [float(v.item()) for v in every]                    # each rank's own clock over the K steps (barrier to barrier)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the exchange of the path by itself: the in-place all-reduce of one moments launch's packed statistics, nothing else on the GPU
        try:
            buf = (gshared[0][0].buffer if (BATCH and MG > 1) else lanes[0].shared.buffer) if BATCH else None
            if buf is not None:
                nb = int(buf.numel()) * buf.element_size()
                for _ in range(3):
                    dist.all_reduce(buf)
                torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    dist.all_reduce(buf)
                torch.cuda.synchronize()
                ar = (time.perf_counter() - t0) / 10
                allreduce_alone = {"bytes": nb, "ms": ar * 1e3, "GBps_bus": 2.0 * (world - 1) / world * nb / ar / 1e9,
                                   "per_step_ms": ar * 1e3 / max(1, (MG if BATCH else 1)),
                                   "note": "ONE in-place all-reduce (RCCL) of the buffer that holds the packed float64 (n, sum x, sum xxT) of the "
                                           "sets of one moments launch, alone on the GPUs; in the timed loop it runs on a stream of its own beside the next launch"}
        except Exception as e:      # noqa: BLE001
            allreduce_alone = {"error": repr(e)}
    step_ms = np.diff(np.array(marks)) * 1e3
    def collect_kernel_ms(handles):
        """-> (mean duration of the tile kernel per launch [ms], launches, mean frame matrices per launch, variant) over the launches
        the handles recorded since set_timing"""
        tot_ms, launches, sets_tot, var = 0.0, 0, 0, -1
        for hnd in handles:
            try:
                k_ms, _, var_h = hnd.last_timing()                       # mean over this handle's recorded launches
            except Exception:       # noqa: BLE001  (a sampled lane that saw no update in a short region)
                hnd.set_timing(False)
                continue
            var = var_h
            sets_h = launch_sets.get(id(hnd)) if (BATCH and MG > 1) else None
            n_h = len(sets_h) if sets_h else 1
            tot_ms += k_ms * n_h; launches += n_h; sets_tot += sum(sets_h) if sets_h else SETS * n_h
            hnd.set_timing(False)
        return (tot_ms / launches if launches else float("nan")), launches, (sets_tot / launches if launches else SETS), var

    kernel_ms, timed_launches_seen, launch_sets_mean, variant = collect_kernel_ms(timed_handles)   # (queried here: the side blocks record more launches)

    side = args.steps > 0 and not args.timed_only
    # ---- side blocks, outside the timed region: the same K steps five more times (median: the timed region above is a few
    # milliseconds long, one outlier moves it), and K steps that re-feed ONE pair -- 204.8 MB, which fit the 256 MiB Infinity
    # Cache: what rounds 1-2 reported
    def block(rotate):
        fence()
        t0 = time.perf_counter()
        run_steps(args.steps, None, rotate)
        fence()
        return time.perf_counter() - t0
    skip = set(os.environ.get("FAD_BENCH_SKIP", "").split(","))          # diagnostics: leave out side blocks (repeat, same, perstep, streams)
    repeat_s = [block(True) for _ in range(5)] if side and "repeat" not in skip else []
    same_pair_s = [block(False) for _ in range(3)] if side and "same" not in skip else []
    # ---- the REALISTIC score, same schedule: k^-1 spectra, frames with an offset, the reference's own float32 running-sum means ON
    # (measured right behind the flat loop's repeats and BEFORE the other side blocks: behind the per-step-launch / other-stream-layout blocks the
    #  same block ran at 4 150 / 2 830 scores/s (detached / attached walk) instead of 4 830 / 4 490 -- r05n against r05m, r05o, r05p; and the
    #  repeats of the flat loop lose ~8 % when they run behind this block -- r05o, r05p.  Why a block is slower for a while behind a different
    #  workload is not understood; the order keeps each of the two top-level numbers behind its own kind)
    realistic = None
    real_pending = []
    saved_pairs = list(pairs)
    if (side or args.realistic_only) and BATCH and rank == 0 and not distributed and not args.no_realistic:
        try:
            from oracle import fad_oracle as O
            rpairs = [make_realistic_sets(torch, device, k) for k in range(N_PAIRS)]
            all_h = [h for q in range(NB_FLY) for ln in blanes[q] for h in (ln.ma, ln.mb)]

            # blocks of at least twelve batches: a block of K = 20 steps is two and a half batches, and the first batch of a block has nothing
            # to hide its walks behind -- its chain waits 1.2 ms for the two walks of its own moments (the flat loop above pays the same fill
            # and drain, but its batches are a third as long); with six batches per block (r05i) the line showed 4 240 where the same loop
            # run on (scripts/probe_realistic.py pipelined) gives 4 700-4 950
            rsteps = max(args.steps, (int(os.environ.get("FAD_BENCH_REALISTIC_BATCHES", "12"))) * BATCH)

            def blocks(nb=5):
                run_steps(3 * BATCH)                                     # (the thread's launch-count hints settle on this kind of pair)
                ts = []
                for _ in range(nb):
                    fence(); t0 = time.perf_counter(); res = run_steps(rsteps); fence()
                    ts.append(time.perf_counter() - t0)
                return ts, res
            pairs[:] = rpairs
            # the frames of this loop are resident and never change: the walk of numpy's running sums may run DETACHED (it waits for nothing
            # and holds nothing up; the chain of a batch waits for it).  `attached` = the default contract (any caller): the walk starts
            # behind the stream's earlier work and the update returns the stream only when it is through.
            # (with the walk the moments go 8 matrices to a launch: the walk is paced per workgroup, 32 of them per matrix, and sixteen matrices'
            #  worth of them crowd the CUs -- r06j: 5 120 / 4 750 / 4 270 scores/s at 4 / 8 / 16 steps per launch; without it, as the flat loop)
            MG_walk = min(MGv[0], 4)
            for h in all_h:
                h.set_reference_mean(True, detached=True)
            MGv[0] = MG_walk
            ts_on, (fad_r, diag_r) = blocks()
            for h in all_h:
                h.set_reference_mean(True, detached=False)
            ts_att, _ = blocks(3)
            MGv[0] = MG
            for h in all_h:
                h.set_reference_mean(False)
            ts_off, _ = blocks(3)
            pairs[:] = saved_pairs
            # one blocking score of realistic pair 0 with reference-order means: latency, route, parity against the CPU oracle
            ra, rb = rpairs[0]
            with hip.Moments(DIM) as qa, hip.Moments(DIM) as qb:
                qa.set_reference_mean(True); qb.set_reference_mean(True)
                lat = []
                for _ in range(7):
                    qa.reset(); qb.reset()
                    torch.cuda.synchronize(); t0 = time.perf_counter()
                    hip.Moments.update_multi([qa, qb], [ra, rb])
                    f1, d1 = hip.frechet_from_moments(qa, qb, mean_dtype=FAD_F16)
                    lat.append((time.perf_counter() - t0) * 1e3)
                qa.set_reference_mean(False); qb.set_reference_mean(False)
                qa.reset(); qb.reset()
                hip.Moments.update_multi([qa, qb], [ra, rb])
                f_exact_means, _ = hip.frechet_from_moments(qa, qb, mean_dtype=FAD_F16)
            # (the oracle of this pair -- ~2 s of BLAS on every host core -- runs at the very end of the bench, next to the CPU baseline: on the
            #  GPU boxes the process sits in a CPU-bandwidth cgroup, a burst of 64 BLAS threads exhausts the quota and the host thread is
            #  frozen for tens of ms at a time afterwards -- `nr_throttled` of cpu.stat counts it, scripts/probe_stall.py prints it -- which is
            #  what slowed whatever block was measured next and what the "30-70 ms stalls" of single calls were: DESIGN.md 6.4)
            real_pending[:] = [ra.cpu().numpy(), rb.cpu().numpy(), float(f1), float(f_exact_means)]
            realistic = {
                "value": float(np.median([rsteps / t for t in ts_on])), "unit": "FAD scores/s", "steps_per_block": rsteps,
                "blocks": {"min": min(rsteps / t for t in ts_on), "max": max(rsteps / t for t in ts_on), "runs": len(ts_on)},
                "value_with_rounded_exact_means": float(np.median([rsteps / t for t in ts_off])),
                "value_with_attached_walk": float(np.median([rsteps / t for t in ts_att])),
                "reference_order_mean_cost": float(np.median(ts_on)) / float(np.median(ts_off)) - 1.0,
                "reference_order_mean_cost_attached": float(np.median(ts_att)) / float(np.median(ts_off)) - 1.0,
                "latency_ms_blocking": float(np.median(lat[2:])), "latency_ms_spread": spread(lat[2:]),
                "route": int(d1.get("route", 0)) if d1["converged"] == 3 else 0, "iterations": int(d1["iters"]),
                "fad": float(f1), "rel_err_vs_oracle": None, "rel_err_vs_oracle_with_rounded_exact_means": None,     # (filled in at the end)
                "batched_last_score": {"fad": float(fad_r), "route": int(diag_r.get("route", 0)) if diag_r["converged"] == 3 else 0, "iterations": int(diag_r["iters"])},
                "workload": f"{N_PAIRS} pairs of 2 x [{N_ROWS} x {DIM}] float16 rotated through the batched schedule of `value`: covariance spectra k^-1 in a shared "
                            "random basis (condition of Sigma_1 Sigma_2 ~3e5), every dimension offset by half the mean standard deviation, "
                            "fad_moments_set_reference_mean on, detached (numpy's float32 running-sum mean, fad.py:48; the frames are resident and "
                            "unchanged, so the walk runs beside everything), float16 mean term",
            }
            del rpairs
        except Exception as e:      # noqa: BLE001  a side block must never break the bench line
            realistic = {"error": repr(e)}
        pairs[:] = saved_pairs
        run_steps(6 * BATCH)            # (the flat pairs again: the thread's launch-count hints settle back before the side blocks below)

    # ... and the same batches with ONE moments launch per step (--moments-group 1: what the line reported before the launches were grouped)
    per_step_launch_s = []
    if side and BATCH and MG > 1 and not distributed and "perstep" not in skip:
        MGv[0] = 1
        run_steps(BATCH)
        per_step_launch_s = [block(True) for _ in range(3)]
        MGv[0] = MG
        run_steps(BATCH)
    # ... and K steps in the other stream layout: by default that is ONE stream for all scores in flight -- no two kernels overlap, the
    # tile kernel's duration there is the kernel alone
    per_stream_s, per_stream_kernel_ms, per_stream_sets = [], None, SETS
    if side and BATCH and "streams" not in skip:
        # batched schedule: the same batches, every one on the OTHER stream layout (ONE stream unless --single-stream was given)
        other = ([torch.cuda.current_stream(device)] * NB_FLY if not args.single_stream else [torch.cuda.Stream(device=device) for _ in range(NB_FLY)])
        saved = list(bstreams)
        bstreams[:] = other
        for q in range(NB_FLY):
            for ln in blanes[q]:
                ln.stream = ln.cstream = bstreams[q]
        run_steps(min(args.steps, BATCH))
        for rep in range(3):
            if rep == 2:                                     # what the overlap does to the tile kernel itself (last block)
                for hnd in timed_handles:
                    hnd.set_timing(2)
                launch_sets.clear()
            fence(); t0 = time.perf_counter(); run_steps(args.steps); fence()
            per_stream_s.append(time.perf_counter() - t0)
        per_stream_kernel_ms, _, per_stream_sets, _ = collect_kernel_ms(timed_handles)
        bstreams[:] = saved
        for q in range(NB_FLY):
            for ln in blanes[q]:
                ln.stream = ln.cstream = bstreams[q]
    elif side and n_lanes > 1 and not G:                    # the OTHER stream layout (one stream for all lanes / one per lane)
        lanes_s = [Lane(k, own=not args.lane_streams) for k in range(n_lanes)]
        run_steps_lanes(min(args.steps, 6), None, True, lanes_s)
        for rep in range(3):
            if rep == 2:
                lanes_s[0].ma.set_timing(2)          # what the overlap does to the tile kernel itself (last block)
            fence(); t0 = time.perf_counter(); run_steps_lanes(args.steps, None, True, lanes_s); fence()
            per_stream_s.append(time.perf_counter() - t0)
        per_stream_kernel_ms = lanes_s[0].ma.last_timing()[0]
        lanes_s[0].ma.set_timing(False)
        for ln in lanes_s:
            ln.shared.close()

    # launches of the tile kernel the events covered inside the timed region
    timed_launches = -(-args.steps // n_lanes)
    if BATCH:
        timed_launches = timed_launches_seen
    if G:
        timed_launches = sum(1 for i in range(args.steps) if (i % G == 0 and (i // G) % NGRP == 0))

    # ---- untimed breakdown (torch events on the same stream: stream 0 is torch's current stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    bm, bf = [], []
    fad0, diag0, reduce_ms = fad, diag, None
    if not args.timed_only:
        ma.set_timing(True)
    for _ in range(0 if args.timed_only else 7):
        ma.reset(); mb.reset()
        ev[0].record(); hip.Moments.update_multi([ma, mb], [a, b]); ev[1].record()
        fad0, diag0 = hip.frechet_from_moments(ma, mb, mean_dtype=FAD_F16); ev[2].record()       # pair 0 = the golden G7 pair
        torch.cuda.synchronize()
        bm.append(ev[0].elapsed_time(ev[1])); bf.append(ev[1].elapsed_time(ev[2]))
    single_kernel_ms = None
    if not args.timed_only:
        single_kernel_ms, reduce_ms, _ = ma.last_timing()               # the tile kernel of ONE score's launch (2 frame matrices), alone on its stream
        ma.set_timing(False)

    # ---- untimed side measurements (rank 0, single GPU)
    extra = {}
    if rank == 0 and not distributed and not args.no_extras:
        for name, fn in (("c2_vggish_e2e", lambda: extra_c2_vggish(torch, hip, device, local_rank)),
                         ("c4_encodec_embed", lambda: extra_c4_encodec(torch, hip, device, local_rank)),
                         ("c4_moments", lambda: extra_c4(torch, hip, device, local_rank)),
                         ("per_song_config5_shape", lambda: extra_c5(torch, hip, device)),
                         ("per_song_config5_encoder_frames", lambda: extra_c5_frames(torch, hip, device)),
                         ("per_song_config4_shape", lambda: extra_c4_songs(torch, hip, device)),
                         ("frechet_decaying_c3", lambda: extra_decaying(torch, hip, device))):
            try:
                cpu_quiet()
                extra[name] = fn()
            except Exception as e:      # noqa: BLE001  side measurements must never break the bench line
                extra[name] = {"error": repr(e)}

    coll_backend, coll_ranks = (dist.get_backend(), dist.get_world_size()) if distributed else (None, 1)
    if distributed:
        dist.destroy_process_group()
    if rank != 0:
        return

    n_gpus = world
    LSETS = launch_sets_mean if BATCH else SETS            # frame matrices per launch of the tile kernel in the timed region (mean)
    flops = LSETS * 2.0 * N_ROWS * DIM * DIM               # algorithmic, per launch
    achieved = flops / (kernel_ms * 1e-3) / 1e12
    nt = -(-DIM // 128)
    # issued: upper-triangular 128 x 128 tiles, 32 MFMAs per 32-row stage off the diagonal, 20 on it; the 256-column-slab kernel
    # issues the 32 x 32 blocks on and above the diagonal (136 of 256 at D = 512: the same count)
    issued = LSETS * 2.0 * N_ROWS * 128 * 128 * (nt * (nt - 1) // 2 + nt * 20.0 / 32.0)
    if variant == 2:
        nb = -(-DIM // 32)
        issued = LSETS * 2.0 * N_ROWS * 32 * 32 * (nb * (nb + 1) // 2)
    traffic, traffic_src = None, None
    tpath = ROOT / "profiles" / "moments_traffic.json"     # separate rocprofv3 --pmc passes of this same command
    if tpath.exists():
        try:
            tj = json.loads(tpath.read_text())
            by_sets = tj.get("by_sets_per_launch", {}).get(str(int(round(LSETS))))
            if by_sets:                                          # the pass that measured launches of this many frame matrices
                traffic, traffic_src = by_sets.get("hbm_bytes_per_launch"), by_sets.get("source")
            elif int(round(LSETS)) == SETS:
                traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        except Exception:       # noqa: BLE001
            traffic = None
    # Frechet chain, by route (diag["route"]): 2 = eight launches: split-float16 products (3 MFMA terms each: iteration 0 needs one
    # product, every later iteration three) + two exact products through digit planes on the int8 MFMA (30 and 26 digit pairs);
    # 1 = round 2's chain (float32 products on the f32 MFMA + two float64 products); 0 = the all-float64 iteration
    it = int(diag["iters"])
    route = int(diag.get("route", 0)) if diag["converged"] == 3 else 0
    d3 = 2.0 * DIM ** 3
    if route == 2:
        n_split = 1 + 3 * max(it - 2, 0)
        work = {"f16_mfma_flops": 3 * n_split * d3, "i8_mfma_ops": (30 + 26) * d3}
        gemms = {"split_f16_products": n_split, "exact_i8_products": 2, "launches": 2 + 1 + 2 * max(it - 2, 0) + 1}
        ideal_ms = (work["f16_mfma_flops"] / (MFMA_F16_PEAK_TFLOPS * 1e12) + work["i8_mfma_ops"] / (2 * MFMA_F16_PEAK_TFLOPS * 1e12)) * 1e3
        peak_note = "time the issued MFMA work needs at the dense peaks (f16 2.5 PFLOP/s, int8 5 POP/s: MI355X_MICROARCH.md) / measured time"
    elif route == 1:
        gemms = {"f32": 1 + 3 * max(it - 2, 0), "f64": 2}
        work = {"f32_mfma_flops": gemms["f32"] * d3, "f64_mfma_flops": 2 * d3}
        ideal_ms = (work["f32_mfma_flops"] / (MFMA_F32_PEAK_TFLOPS * 1e12) + work["f64_mfma_flops"] / 78.6e12) * 1e3
        peak_note = "f32-input MFMA 157.3 TFLOP/s, f64 MFMA 78.6 TFLOP/s (datasheet)"
    else:
        gemms = {"f64": 1 + 1 + 3 * max(it - 1, 0)}
        work = {"f64_mfma_flops": gemms["f64"] * d3}
        ideal_ms = work["f64_mfma_flops"] / 78.6e12 * 1e3
        peak_note = "f64 MFMA 78.6 TFLOP/s (datasheet)"
    # SURVEY 8-d3's rule for the dominant kernel
    t_k = kernel_ms * 1e-3
    issued_frac = issued / t_k / 1e12 / MFMA_F16_PEAK_TFLOPS
    hbm_gbs = LSETS * N_ROWS * DIM * 2 / t_k / 1e9
    hbm_frac = hbm_gbs / HBM_PEAK_GBS
    if hbm_frac >= issued_frac:
        roof_bound, roof_achieved, roof_peak, roof_unit, roof_frac = "hbm", hbm_gbs, HBM_PEAK_GBS, "GB/s", hbm_frac
    else:
        roof_bound, roof_achieved, roof_peak, roof_unit, roof_frac = "mfma", issued / t_k / 1e12, MFMA_F16_PEAK_TFLOPS, "TFLOP/s", issued_frac
    single_launch = None
    if single_kernel_ms:                                                 # ONE score's launch: 2 frame matrices, nothing else on the device
        t1 = single_kernel_ms * 1e-3
        f1 = SETS * 2.0 * N_ROWS * DIM * DIM
        i1 = issued * (SETS / LSETS)
        single_launch = {"sets_per_launch": SETS, "kernel_ms": single_kernel_ms, "frac_algorithmic": f1 / t1 / 1e12 / MFMA_F16_PEAK_TFLOPS,
                         "frac_issued": i1 / t1 / 1e12 / MFMA_F16_PEAK_TFLOPS, "hbm_frac_of_8TBps": SETS * N_ROWS * DIM * 2 / t1 / 1e9 / HBM_PEAK_GBS,
                         "frac": max(i1 / t1 / 1e12 / MFMA_F16_PEAK_TFLOPS, SETS * N_ROWS * DIM * 2 / t1 / 1e9 / HBM_PEAK_GBS),
                         "note": "the tile kernel of a launch that carries ONE score's two sets (what a caller with a single pair gets); "
                                 "`roofline` itself is the timed loop's launch of `sets_per_launch` frame matrices"}
    fr_ms = float(np.median(bf)) if bf else None
    fr_flops = float(sum(work.values()))
    kernel_name = {0: "moments_tile_h16_tr<f16> (128 x 128 tiles)", 1: "moments_tile_f64", 2: "moments_tile256<f16> (256-column slabs)"}.get(variant, str(variant))
    dtype_fr = {2: "split-f16 MFMA iterations (f32 accumulate) + exact int8-MFMA products (int32 accumulate) + f64 correction",
                1: "f32 MFMA iterations + f64 correction", 0: "f64 MFMA iterations"}[route]
    out = {
        "metric": "FAD scores/sec + cov-GEMM TFLOP/s (% MFMA peak), N=100k D=512",
        "value": n_gpus * args.steps / elapsed,
        "unit": "FAD scores/s (config-3-sized: 2 x [100000 x 512] fp16 frames per score per GPU)",
        "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": f"f16 in, f16 MFMA with f32 accumulate over <= 8192-row runs, f64 across runs (moments); {dtype_fr} (Frechet)", "data": "synthetic",
        "config": {"workload": "C3: CLAP-sized embeddings N=100000 D=512 fp16 per set per GPU, "
                               "moments of both sets + Newton-Schulz Frechet, inputs resident in HBM "
                               f"({N_PAIRS} distinct pairs rotated: every step reads its frames from HBM, not from the Infinity Cache)",
                   "rows_per_set_per_gpu": N_ROWS, "dim": DIM,
                   "sharding": ("rows sharded over ranks; ONE in-place all-reduce per moments launch over the buffer holding the packed "
                                f"(n, sum x, sum xxT) fp64 of its {2 * MG if BATCH else 2} sets [{2 * MG if BATCH else 2} x {plen} doubles]") if distributed else "single GPU, no collective",
                   "collective_backend": coll_backend, "collective_ranks": coll_ranks,
                   "per_rank_ms_per_step": ([1e3 * v / args.steps for v in per_rank_s] if per_rank_s else None),
                   "allreduce_alone": allreduce_alone},
        "fad": fad0, "fad_pair": ("pair 0 = the golden G7 pair (tests/golden/recipes.c3_pair, numpy seeds 10 / 11)" if not args.timed_only else "last timed step"),
        "parity_rel_err_vs_golden_g7": ((abs(fad0 - float(golden_g7()[1]["fad"])) / float(golden_g7()[1]["fad"]))
                                        if (not args.timed_only and not distributed and golden_g7()[1] is not None) else None),
        "fad_last_timed_step": fad, "timed_only": bool(args.timed_only), "chain_cus_per_xcd": int(args.chain_cus), "group": G, "batched_chains": BATCH,
        "newton_schulz_iters": diag["iters"], "ns_converged": diag["converged"],
        "frames_per_s": n_gpus * args.steps * 2 * N_ROWS / elapsed,
        "value_repeat_median": float(np.median([n_gpus * args.steps / t for t in repeat_s])) if repeat_s else None,
        "input_rotation": {"pairs": N_PAIRS, "bytes": N_PAIRS * SETS * N_ROWS * DIM * 2,
                           "note": f"step i feeds pair i % {N_PAIRS}: the working set of the timed loop ({N_PAIRS * SETS * N_ROWS * DIM * 2 / 1e6:.0f} MB) exceeds the 256 MiB Infinity "
                                   "Cache, every step streams its 204.8 MB from HBM"},
        "value_repeat_blocks": {"median": float(np.median([n_gpus * args.steps / t for t in repeat_s])) if repeat_s else None,
                                "min": min(n_gpus * args.steps / t for t in repeat_s) if repeat_s else None,
                                "max": max(n_gpus * args.steps / t for t in repeat_s) if repeat_s else None, "blocks": len(repeat_s),
                                "note": "the same K steps repeated outside the timed region (rank 0's clock)"},
        ("value_single_stream" if args.lane_streams else ("value_three_batch_streams" if BATCH else "value_stream_per_score")): {
            "median": float(np.median([n_gpus * args.steps / t for t in per_stream_s])) if per_stream_s else None,
            "blocks": len(per_stream_s), "tile_kernel_ms": per_stream_kernel_ms,
            "note": ("the same K steps with all scores in flight on ONE stream (--single-stream): no two kernels overlap" if args.lane_streams
                     else ("the same K steps with one HIP stream per batch in flight (--multi-stream): the idle CUs of one batch's small launches "
                           "take another batch's kernels; `tile_kernel_ms` there includes the time a dispatch waits for CUs" if BATCH
                           else "the same K steps with one HIP stream per score in flight: chains and moments kernels of consecutive scores overlap"))},
        "value_one_moments_launch_per_step": ({"median": float(np.median([n_gpus * args.steps / t for t in per_step_launch_s])), "blocks": len(per_step_launch_s),
                                               "note": "the same K steps with --moments-group 1: a tile-kernel launch, a guard launch and a reduce per step (2 frame "
                                                       "matrices each) instead of one of each per batch of 16 steps"} if per_step_launch_s else None),
        "value_same_pair": {"median": float(np.median([n_gpus * args.steps / t for t in same_pair_s])) if same_pair_s else None,
                            "blocks": len(same_pair_s),
                            "note": "K steps that re-feed ONE pair (204.8 MB: Infinity-Cache resident) -- the loop rounds 1-2 timed"},
        "host_ms_per_step": {"enqueue_moments": 1e3 * host_timed[0] / max(args.steps, 1), "enqueue_chain": 1e3 * host_timed[1] / max(args.steps, 1),
                             "wait_and_collect": 1e3 * host_timed[2] / max(args.steps, 1),
                             "note": "host wall-clock inside the timed loop: what the Python loop spends enqueueing (the device is never waited "
                                     "for there) and in FrechetJob.result (which waits for the oldest score in flight)"},
        "scores_in_flight": (3 * BATCH if BATCH else n_lanes), "lane_streams": bool(args.lane_streams and n_lanes > 1),
        "moments_group": MG,
        "schedule": (f"batched: moments of {BATCH} steps ({MG} steps = {2 * MG} frame matrices per launch of the tile kernel and of the reduce: "
                     f"fad_moments_update_multi), then ONE square-root chain for their {BATCH} scores (fad_frechet_from_moments_multi_begin), "
                     + ("3 batches in flight on ONE stream" if args.single_stream else "3 batches in flight on 3 streams") if BATCH else (f"grouped ({G})" if G else f"lanes: one chain per score, {n_lanes} in flight")),
        "step_ms_spread": {"min": float(step_ms.min()), "p10": float(np.percentile(step_ms, 10)), "median": float(np.median(step_ms)),
                           "p90": float(np.percentile(step_ms, 90)), "max": float(step_ms.max())},
        "breakdown_ms": {"moments_both_sets": float(np.median(bm)) if bm else None, "frechet": fr_ms,
                         "moments_reduce_kernels": reduce_ms},
        # what ONE blocking score costs a caller who has nothing else in flight (moments of both sets + the square-root chain)
        "latency_ms_blocking": (float(np.median(bm)) + fr_ms) if (bm and fr_ms is not None) else None,
        "value_realistic": realistic.get("value") if isinstance(realistic, dict) else None,
        "realistic": realistic,
        "roofline": {"kernel": kernel_name,
                     # achieved = max(issued MFMA flops / peak, algorithmic bytes / HBM peak), the bound named
                     "bound": roof_bound, "achieved": roof_achieved, "peak": roof_peak, "unit": roof_unit,
                     "frac": roof_frac, "traffic": traffic,
                     "frac_rule": "max(issued MFMA flops / t / 2500 TFLOP/s, algorithmic bytes / t / 8 TB/s) -- SURVEY.md 8-d3; the larger "
                                  "one names the bound.  `frac_algorithmic` (2 N D^2 per set, the symmetry credited: what rounds 1-4 "
                                  "reported as `frac`) and `frac_issued` stay beside it",
                     "achieved_tflops_algorithmic": achieved, "frac_algorithmic": achieved / MFMA_F16_PEAK_TFLOPS,
                     "single_score_launch": single_launch,
                     "traffic_source": traffic_src or "not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of "
                                                      "this command, committed under profiles/ (FETCH_SIZE doubled per the gfx950 note)",
                     "kernel_ms": kernel_ms, "kernel_ms_samples": timed_launches, "sets_per_launch": LSETS, "algorithmic_flops_per_launch": flops,
                     # only the upper-triangular 128 x 128 tiles of the symmetric result are issued
                     "issued_flops_per_launch": issued, "frac_issued": issued / (kernel_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                     # utilisation of the matrix pipe comes from ISSUED flops; `frac` above credits the symmetry
                     "mfma_util": issued / (kernel_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                     "mfma_util_note": "issued MFMA flops (the 32 x 32 blocks on and above the diagonal; on 128 x 128 tiles: upper-triangular "
                                       "tiles, 20 of 32 MFMAs on a diagonal tile) / kernel time / dense fp16 peak; `frac` = algorithmic "
                                       "2 N D^2 per set over the same time",
                     "algorithmic_bytes_per_launch": LSETS * N_ROWS * DIM * 2,
                     "hbm_GBps_algorithmic": LSETS * N_ROWS * DIM * 2 / (kernel_ms * 1e-3) / 1e9,
                     "hbm_frac_of_8TBps": LSETS * N_ROWS * DIM * 2 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "contended": bool(args.lane_streams and (BATCH or n_lanes > 1)),
                     "contended_note": "kernel_ms = the dispatch's own begin -> end stamps (hipExtLaunchKernel start / stop events on the "
                                       "launch's stream: the interval rocprofv3 --kernel-trace reports), sampled inside the timed loop, "
                                       "where other streams' kernels (the square-root chains of earlier scores) may hold CUs while it "
                                       "starts; `alone` = the same launches with all scores on one stream (no two kernels overlap), "
                                       "side block of this run",
                     "alone": ({"kernel_ms": per_stream_kernel_ms, "sets_per_launch": per_stream_sets,
                                "achieved": flops * (per_stream_sets / LSETS) / (per_stream_kernel_ms * 1e-3) / 1e12,
                                "frac": flops * (per_stream_sets / LSETS) / (per_stream_kernel_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                "mfma_util": issued * (per_stream_sets / LSETS) / (per_stream_kernel_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS}
                               if (args.lane_streams and per_stream_kernel_ms) else
                               ({"kernel_ms": kernel_ms, "sets_per_launch": LSETS, "achieved": achieved, "frac": achieved / MFMA_F16_PEAK_TFLOPS,
                                 "mfma_util": issued / (kernel_ms * 1e-3) / 1e12 / MFMA_F16_PEAK_TFLOPS,
                                 "note": "the timed loop runs on ONE stream: this IS the kernel alone"} if BATCH else None))},
        "roofline_frechet": {"route": {2: "eight launches: split-float16 Newton-Schulz + exact int8-MFMA products (csrc/ns_fast.h)",
                                       1: "float32 Newton-Schulz on the f32 MFMA + float64 correction", 0: "all-float64 iteration"}[route],
                             "bound": "launch chain: ~4.2 us per dependent launch before it does anything, then the CU's vector-memory "
                                      "path (128-192 KB of operands per 32 x 32 tile); the matrix pipes are idle most of the time",
                             "gemms": gemms, "work": work, "ms": fr_ms, "achieved": (fr_flops / (fr_ms * 1e-3) / 1e12) if fr_ms else None,
                             "unit": "T(FL)OP/s issued",
                             "ideal_ms_at_mfma_peaks": ideal_ms, "frac": (ideal_ms / fr_ms) if fr_ms else None, "peak_source": peak_note},
    }
    if extra:
        out["extra"] = extra
    if n_gpus == 1 and not args.no_cpu_baseline:
        a_host, b_host = a.cpu().numpy(), b.cpu().numpy()
        if not args.no_extras:
            try:
                import fadtk_amd
                cpu_quiet()
                out.setdefault("extra", {})["host_resident"] = extra_host(fadtk_amd, a_host, b_host, fad0)
            except Exception as e:      # noqa: BLE001
                out.setdefault("extra", {})["host_resident"] = {"error": repr(e)}
            try:
                cpu_quiet()
                out["extra"]["score_inf_c3"] = extra_score_inf(fadtk_amd, a_host, b_host)
            except Exception as e:      # noqa: BLE001
                out["extra"]["score_inf_c3"] = {"error": repr(e)}
        base, fad_cpu = cpu_baseline(a_host, b_host)
        out["cpu_baseline"] = base
        out["speedup_vs_cpu"] = out["value"] / base["value"]
        # parity on the very same inputs: the step asks for the reference's float16 mean term (mean_dtype = FAD_F16)
        out["parity_rel_err_vs_cpu"] = abs(fad0 - fad_cpu) / abs(fad_cpu)
        fad64, _ = hip.frechet_from_moments(ma, mb)
        out["parity_rel_err_vs_cpu_f64_means"] = abs(fad64 - fad_cpu) / abs(fad_cpu)
        out["fad_cpu"] = fad_cpu
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if real_pending and isinstance(out.get("realistic"), dict) and "error" not in out["realistic"]:
        from oracle import fad_oracle as O
        t0 = time.perf_counter()
        ref = float(O.fad_between(real_pending[0], real_pending[1]))
        out["realistic"]["rel_err_vs_oracle"] = abs(real_pending[2] - ref) / abs(ref)
        out["realistic"]["rel_err_vs_oracle_with_rounded_exact_means"] = abs(real_pending[3] - ref) / abs(ref)
        out["realistic"]["oracle_seconds"] = time.perf_counter() - t0
    gc.enable()
    print(json.dumps(out), flush=True)
    os.dup2(2, 1)                       # whatever libraries say while the process winds down stays off stdout as well


if __name__ == "__main__":
    main()
