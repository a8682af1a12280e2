"""World-size-2 test of the multi-rank path on CPU (gloo): contiguous sharding, one all-reduce of the
packed sufficient statistics, gather of per-song results.  The per-rank moments are produced with
numpy here (no GPU in this tier); on the GPU box the same dist.* calls move the HIP accumulators."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _packed(rows):
    x = rows.astype(np.float64)
    d = x.shape[1]
    return np.concatenate([[x.shape[0]], x.sum(0), (x.T @ x).reshape(-1)]) if x.shape[0] else np.zeros(1 + d + d * d)


def _worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests" / "golden"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import recipes as R
    from fadtk_amd import dist
    assert dist.init("gloo") and dist.world_size() == world and dist.rank() == rank
    files = R.ragged_files(70, 37, 24)                                  # the G3 "dataset": 37 per-file matrices
    mine = dist.shard(list(range(len(files))))
    local = sum((_packed(files[i]) for i in mine), np.zeros(1 + 24 + 24 * 24))
    total = dist.allreduce_packed(local)                                # the one collective of the path
    scores = dist.gather_objects([(i, float(files[i].sum())) for i in mine])
    listing = dist.broadcast_object(["a", "b", rank] if rank == 0 else None)      # rank 0's list on every rank
    assert listing == ["a", "b", 0]
    dist.barrier()
    if rank == 0:
        np.save(Path(tmp) / "total.npy", total)
        np.save(Path(tmp) / "order.npy", np.array([i for part in scores for i, _ in part]))
    import torch.distributed as td
    td.destroy_process_group()


def test_two_rank_reduce_of_sufficient_statistics(tmp_path, golden_dir):
    import recipes as R
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    total = np.load(tmp_path / "total.npy")
    files = R.ragged_files(70, 37, 24)
    allrows = np.concatenate(files).astype(np.float64)
    d = 24
    assert total[0] == allrows.shape[0]
    np.testing.assert_allclose(total[1:1 + d], allrows.sum(0), rtol=1e-13)
    np.testing.assert_allclose(total[1 + d:].reshape(d, d), allrows.T @ allrows, rtol=1e-13)
    # finalised statistics equal the single-process ones (and the reference's, up to its fp16 mean quirk)
    n = total[0]
    cov = (total[1 + d:].reshape(d, d) - np.outer(total[1:1 + d], total[1:1 + d]) / n) / (n - 1)
    z = np.load(golden_dir / "g3_online.npz")
    np.testing.assert_allclose(cov, z["cov"], rtol=0, atol=1e-4 * np.abs(z["cov"]).max())   # Q1: per-file fp16 means
    assert list(np.load(tmp_path / "order.npy")) == list(range(37))      # rank-ordered gather keeps file order


def _stats_worker(rank, world, port, tmp):
    sys.path.insert(0, str(ROOT))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as td
    from fadtk_amd import dist
    from fadtk_amd.fad import FrechetAudioDistance

    class _Loader:                                   # load_stats only needs the model's name
        name = "toy"

    assert dist.init("gloo")
    fad = FrechetAudioDistance(_Loader(), load_model=False)
    mu, cov = fad.load_stats(Path(tmp), collective=True)            # every rank: rank 0 answers "is it cached?" for all
    assert mu.shape == (4,) and cov.shape == (4, 4)
    dist.barrier()
    if rank != 0:                                    # `fadtk <model> <base> <eval> --gpus 2`: the other ranks leave here ...
        td.destroy_process_group()
        return
    mu0, cov0 = fad.load_stats(Path(tmp))            # ... and rank 0 resolves the statistics ALONE: no collective may run
    assert np.array_equal(mu0, mu) and np.array_equal(cov0, cov)
    np.save(Path(tmp) / "rank0_done.npy", np.array([1]))
    td.destroy_process_group()


def test_load_stats_runs_no_collective_unless_asked(tmp_path):
    """with several ranks `score` / `score_inf` are rank 0's alone (cli.score_main) -- load_stats must not wait for
    ranks that have left; only the `collective=True` form (score_individual: every rank calls it) may synchronise."""
    cache = tmp_path / "stats" / "toy"
    cache.mkdir(parents=True)
    np.save(cache / "mu.npy", np.arange(4.0)); np.save(cache / "cov.npy", np.eye(4))
    mp.spawn(_stats_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "rank0_done.npy").exists()
