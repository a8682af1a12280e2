"""Two ranks driving the HIP multi-GPU path on ONE GPU (gloo rendezvous, both ranks on cuda:0): file sharding, the
HIP accumulators in a shared device buffer, the per-file mean terms, ONE all-reduce, sharded --indiv scoring and the
rank-ordered gather -- everything `--gpus 2` does except that the collective runs over gloo instead of RCCL
(a single-GPU box cannot host a 2-rank RCCL communicator).  Results are checked against the oracle on the union."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

import recipes as R
from oracle import fad_oracle as O

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank_main(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "golden"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FAD_DIST_BACKEND="gloo", FADTK_AMD_RANDOM_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import fadtk_amd
    from fadtk_amd import dist, hip
    from fadtk_amd.fad_batch import cache_embedding_files, embed_and_accumulate
    from fadtk_amd.model_loader import EncodecEmbModel
    tmp = Path(tmp)
    ml = EncodecEmbModel("24k")
    # 1. fused embedding + statistics over file shards, one all-reduce of the shared buffer
    mu, cov = embed_and_accumulate(tmp / "base", ml, workers=2)
    assert dist.world_size() == world and dist.rank() == rank
    # 2. plain embedding cache of the eval set (rank 0 lists, everybody shards the same list)
    cache_embedding_files(tmp / "eval", ml, workers=2)
    # 3. per-song scores, songs sharded over the ranks, gathered in file order; load_stats served by rank 0's cache
    fad = fadtk_amd.FrechetAudioDistance(ml, audio_load_worker=2, load_model=False, device=0)
    fad.score_individual(tmp / "base", tmp / "eval", tmp / "indiv.csv")
    # 4. the shared-buffer reduce on bare accumulators: rank r feeds rows r::world
    x = torch.from_numpy(R.normal_rows(7, 4001, 128)).cuda()
    sh = dist.SharedStats(128, 2, 0)
    hip.Moments.update_multi(sh.moments, [x[rank::world], (2 * x)[rank::world].contiguous()])
    sh.allreduce()
    packed = [m.export() for m in sh.moments]
    sh.close()
    dist.barrier()
    if rank == 0:
        np.savez(tmp / "rank0.npz", mu=mu, cov=cov, p0=packed[0], p1=packed[1])
    import torch.distributed as td
    td.destroy_process_group()


def test_two_ranks_on_one_gpu_match_the_oracle_on_the_union(tmp_path):
    import torch.multiprocessing as mp
    from fadtk_amd import audio
    for name, n, seed, gain in (("base", 9, 800, 1.0), ("eval", 5, 900, 0.7)):
        (tmp_path / name).mkdir()
        for i in range(n):
            audio.write_pcm16(tmp_path / name / f"clip{i:03d}.wav", gain * R.audio_clip(seed + i, int((1.0 + 0.5 * (i % 3)) * 24000), 24000), 24000)
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)

    z = np.load(tmp_path / "rank0.npz")
    blocks = [np.load(p) for p in sorted((tmp_path / "base" / "embeddings" / "encodec-emb").glob("*.npy"))]
    assert len(blocks) == 9 and all(b.dtype == np.float16 and b.shape[1] == 128 for b in blocks)
    mu_o, cov_o = O.statistics_online(blocks)                   # the reference's online path on the union of both shards
    np.testing.assert_allclose(z["mu"], mu_o, rtol=0, atol=1e-9 * np.abs(mu_o).max() + 1e-12)
    np.testing.assert_allclose(z["cov"], cov_o, rtol=0, atol=2e-6 * np.abs(cov_o).max())
    assert np.array_equal(np.load(tmp_path / "base" / "stats" / "encodec-emb" / "cov.npy"), z["cov"])

    lines = (tmp_path / "indiv.csv").read_text().split("\n")
    assert len(lines) == 5
    got = {Path(ln.rsplit(",", 1)[0]).name: float(ln.rsplit(",", 1)[1]) for ln in lines}
    for p in sorted((tmp_path / "eval").glob("*.wav")):
        e = np.load(tmp_path / "eval" / "embeddings" / "encodec-emb" / (p.stem + ".npy"))
        ref = O.frechet_distance(z["mu"], z["cov"], *O.embd_statistics(e), run_sqrtm=False)
        assert abs(got[p.name] - ref) / abs(ref) < 1e-4, p.name

    x = R.normal_rows(7, 4001, 128).astype(np.float64)
    for p, rows in ((z["p0"], x), (z["p1"], 2 * x)):
        assert p[0] == 4001
        np.testing.assert_allclose(p[1:129], rows.sum(0), rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(p[129:].reshape(128, 128), rows.T @ rows, rtol=0, atol=1e-6 * np.abs(rows.T @ rows).max())


def _cli_rank_main(rank, world, port, tmp, extra):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "golden"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      FAD_DIST_BACKEND="gloo", FADTK_AMD_RANDOM_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    from fadtk_amd import cli
    tmp = Path(tmp)
    sys.argv = ["fadtk", "encodec-emb", str(tmp / "base"), str(tmp / "eval"), str(tmp / f"out{extra}.csv"), "-w", "2"] + \
        (["--inf"] if extra == "inf" else [])
    cli.score_main()                                 # ranks != 0 return after the embedding barrier; rank 0 scores alone


@pytest.mark.parametrize("extra", ["", "inf"])
def test_plain_score_cli_with_two_ranks(tmp_path, extra):
    """`fadtk <model> <base> <eval> --gpus 2` -- plain score and --inf -- used to leave rank 0 alone inside
    the collectives of load_stats.  Two ranks (gloo, one GPU) run the launcher's main; the CSV line must match the oracle."""
    import torch.multiprocessing as mp
    from fadtk_amd import audio
    n_eval = 12 if extra == "inf" else 4
    for name, n, seed, gain in (("base", 6, 1800, 1.0), ("eval", n_eval, 1900, 0.8)):
        (tmp_path / name).mkdir()
        for i in range(n):
            audio.write_pcm16(tmp_path / name / f"clip{i:03d}.wav", gain * R.audio_clip(seed + i, int((2.0 + 0.5 * (i % 3)) * 24000), 24000), 24000)
    mp.spawn(_cli_rank_main, args=(2, _free_port(), str(tmp_path), extra), nprocs=2, join=True)
    header, line = (tmp_path / f"out{extra}.csv").read_text().strip().split("\n")
    assert header == "model,baseline,eval,score,inf_r2,time"
    score = float(line.split(",")[3])
    if extra == "":
        stats = []
        for name in ("base", "eval"):
            blocks = [np.load(p) for p in sorted((tmp_path / name / "embeddings" / "encodec-emb").glob("*.npy"))]
            stats.append(O.statistics_online(blocks))
        ref = O.frechet_distance(*stats[0], *stats[1], run_sqrtm=False)
        assert abs(score - ref) / abs(ref) < 1e-4
    else:
        assert np.isfinite(score) and line.split(",")[4] != "None"


def test_bench_multi_rank_path_on_one_gpu(tmp_path):
    """bench.py's N > 1 path -- process group over RCCL, comm stream, fed / reduced events, ONE in-place
    all_reduce over the buffer that holds both sets' statistics -- runs in no other automated test.  FAD_BENCH_FORCE_DIST=1 takes
    it with a one-rank RCCL communicator; the JSON line must carry the driver's fields, say which backend and how many ranks the
    collective saw, and give the same score as the plain path on the same frames."""
    import json
    import subprocess
    import torch
    sys.path.insert(0, str(ROOT))
    import bench
    from fadtk_amd import hip
    env = dict(os.environ, FAD_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["steps"] == 3 and out["warmup"] == 1 and out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["collective_ranks"] == 1
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(out["roofline"])
    a, b = bench.make_sets(torch, torch.device("cuda", 0), 0, 0)
    with hip.Moments(bench.DIM) as ma, hip.Moments(bench.DIM) as mb:
        hip.Moments.update_multi([ma, mb], [a, b])
        want, _ = hip.frechet_from_moments(ma, mb, mean_dtype=bench.FAD_F16)
    assert abs(out["fad"] - want) <= 1e-9 * abs(want)


def test_bench_starts_its_own_ranks_when_world_size_is_unset():
    """`python bench.py --gpus N` from a plain shell re-launches itself under torch.distributed.run (here N = 1,
    forced onto the multi-rank path): one JSON line from rank 0 with the collective's backend and rank count."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR")}
    env.update(FAD_BENCH_FORCE_DIST="1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--timed-only"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 3 and out["value"] > 0 and out["timed_only"] is True
    assert out["config"]["collective_backend"] == "nccl" and out["config"]["collective_ranks"] == 1
    assert out["roofline"]["frac"] > 0


def test_fused_stats_cli_under_torchrun_with_rccl(tmp_path):
    """`fadtk <model> <base> <eval> <csv> --fused-stats` relaunched the way `--gpus N` relaunches it (torch.distributed.run),
    with a ONE-rank RCCL group (FAD_DIST_FORCE=1): embed_and_accumulate's shared buffer goes through the nccl all_reduce, rank 0
    writes the statistics cache and scores.  Checked against the oracle's online statistics of the cached embeddings."""
    import subprocess
    from fadtk_amd import audio
    for name, n, seed, gain in (("base", 7, 2800, 1.0), ("eval", 5, 2900, 0.75)):
        (tmp_path / name).mkdir()
        for i in range(n):
            audio.write_pcm16(tmp_path / name / f"clip{i:03d}.wav", gain * R.audio_clip(seed + i, int((2.0 + 0.5 * (i % 3)) * 24000), 24000), 24000)
    env = dict(os.environ, FAD_DIST_FORCE="1", FADTK_AMD_RANDOM_WEIGHTS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=str(ROOT))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "FAD_DIST_BACKEND"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "fadtk_amd", "encodec-emb", str(tmp_path / "base"), str(tmp_path / "eval"),
           str(tmp_path / "out.csv"), "--fused-stats", "-w", "2"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "accumulated on 1 GPU(s)" in r.stderr + r.stdout
    header, line = (tmp_path / "out.csv").read_text().strip().split("\n")
    score = float(line.split(",")[3])
    stats = []
    for name in ("base", "eval"):
        blocks = [np.load(p) for p in sorted((tmp_path / name / "embeddings" / "encodec-emb").glob("*.npy"))]
        stats.append(O.statistics_online(blocks))
        assert (tmp_path / name / "stats" / "encodec-emb" / "cov.npy").exists()
    ref = O.frechet_distance(*stats[0], *stats[1], run_sqrtm=False)
    assert abs(score - ref) / abs(ref) < 1e-4


def _rccl_rank_main(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "golden"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("FAD_DIST_BACKEND", None)                       # the default: nccl = RCCL over xGMI
    import torch
    from fadtk_amd import dist, hip
    torch.cuda.set_device(rank)
    assert dist.init()                                             # joins RANK / WORLD_SIZE / MASTER_* with the nccl (= RCCL) backend
    x = torch.from_numpy(R.normal_rows(17, 40001, 512)).cuda(rank)
    y = torch.from_numpy(R.normal_rows(18, 30003, 512, 1.05, 0.02)).cuda(rank)
    sh = dist.SharedStats(512, 2, rank)                            # both sets' packed statistics in ONE device buffer on this rank's GPU
    hip.Moments.update_multi(sh.moments, [x[rank::world].contiguous(), y[rank::world].contiguous()])
    sh.allreduce()                                                 # ONE in-place all-reduce over RCCL
    assert dist.world_size() == world and dist.rank() == rank
    import torch.distributed as td
    assert td.get_backend() == "nccl"
    fad, diag = hip.frechet_from_moments(sh.moments[0], sh.moments[1], mean_dtype=0)     # replicas-only Frechet: every rank scores the union
    packed = [m.export() for m in sh.moments]
    sh.close()
    np.savez(Path(tmp) / f"rank{rank}.npz", p0=packed[0], p1=packed[1], fad=fad)
    dist.barrier()
    td.destroy_process_group()


def test_two_ranks_on_two_gpus_reduce_over_rccl(tmp_path):
    """the two-rank moments path over RCCL itself -- one process per GPU, row shards, ONE in-place all-reduce of the
    packed (n, sum x, sum xxT) of both sets, the Frechet distance replicated -- wherever the box has two GPUs (the driver's 8-GPU node;
    skipped on the one-GPU boxes the round's own visits get).  Every rank must hold the union's statistics and the oracle's score."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: a one-GPU box cannot host a two-rank RCCL communicator")
    import torch.multiprocessing as mp
    mp.spawn(_rccl_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    x = R.normal_rows(17, 40001, 512); y = R.normal_rows(18, 30003, 512, 1.05, 0.02)
    ref = O.fad_between(x, y)
    z = [np.load(tmp_path / f"rank{r}.npz") for r in range(2)]
    for r in range(2):
        for p, rows in ((z[r]["p0"], x.astype(np.float64)), (z[r]["p1"], y.astype(np.float64))):
            assert p[0] == rows.shape[0]
            np.testing.assert_allclose(p[1:513], rows.sum(0), rtol=1e-6, atol=1e-5)
            np.testing.assert_allclose(p[513:].reshape(512, 512), rows.T @ rows, rtol=0, atol=2e-6 * np.abs(rows.T @ rows).max())
        assert abs(float(z[r]["fad"]) - ref) <= 1e-5 * abs(ref)
    assert np.array_equal(z[0]["p0"], z[1]["p0"]) and np.array_equal(z[0]["p1"], z[1]["p1"])      # an all-reduce leaves every rank the same sums
