"""Batch embedding driver -- mirror of fadtk/fad_batch.py:15-48 for one process per GPU.

fadtk splits the uncached files into ``workers`` contiguous chunks and embeds them in a spawn pool,
every worker loading its own copy of the model on cuda:0.  Here the unit of parallelism is the GPU:
when launched under torchrun (RANK / WORLD_SIZE set) each rank takes one contiguous shard of the
files and embeds it on its own device; ``workers`` becomes the number of host threads that decode /
resample audio ahead of the GPU.  Results land in the same cache layout
(<dir>/embeddings/<model>/<stem>.npy, float16).
"""
from __future__ import annotations

import logging
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Union

from . import dist
from .fad import FrechetAudioDistance
from .utils import get_cache_embedding_path

log = logging.getLogger("fadtk_amd")


def _cache_embedding_batch(fs, ml, workers: int = 8, moments=None, file_sums=None, **kwargs):
    """Embed a list of files on this process's GPU; audio decode runs ``workers`` files ahead.
    With ``moments`` (a fadtk_amd.hip.Moments) every embedding is also accumulated while it is still in HBM."""
    import numpy as np
    fad = FrechetAudioDistance(ml, audio_load_worker=workers, **kwargs)
    if not fs:
        return
    depth = max(1, workers)
    with ThreadPoolExecutor(max_workers=depth) as pool:
        pending = []
        it = iter(fs)

        def submit():
            f = next(it, None)
            if f is not None:
                pending.append((f, pool.submit(fad.load_audio, f)))

        for _ in range(depth):
            submit()
        while pending:
            f, fut = pending.pop(0)
            submit()
            cache = get_cache_embedding_path(ml.name, f)
            if cache.exists():
                if moments is not None:
                    e = np.load(cache)
                    moments.update(e)
                    if file_sums is not None:
                        file_sums.append((e.shape[0], e.astype(np.float64).sum(axis=0), e.dtype))
                continue
            log.info(f"Loading {f} using {ml.name}")
            try:
                if moments is not None:          # keep the frames on the device: fp16 exactly as stored, then moments
                    import torch
                    dev = ml._get_embedding(fut.result()).detach()
                    dev = dev.to(torch.float16) if dev.dtype == torch.float32 else dev
                    if dev.shape[0] > 0:
                        moments.update(dev.contiguous())
                    embd = dev.cpu().numpy()
                    if file_sums is not None:          # per-file column sums (D numbers) for the reference's mean quirk
                        file_sums.append((embd.shape[0], dev.to(torch.float64).sum(dim=0).cpu().numpy(), embd.dtype))
                else:
                    embd = ml.get_embedding(fut.result())
            except Exception as e:      # noqa: BLE001  a bad file must not take the shard down
                log.error(f"Embedding {f} with {ml.name} failed: {e}")
                continue
            cache.parent.mkdir(parents=True, exist_ok=True)
            np.save(cache, embd)


def cache_embedding_files(files: Union[list, str, Path], ml, workers: int = 8, **kwargs):
    """Get embeddings for all audio files in a directory (or list), skipping cached ones."""
    if isinstance(files, (str, Path)):
        files = sorted(Path(files).glob("*.*"))
    files = [Path(f) for f in files if not get_cache_embedding_path(ml.name, f).exists()]
    if len(files) == 0:
        log.info("All files already have embeddings, skipping.")
        return
    log.info(f"[Frechet Audio Distance] Loading {len(files)} audio files...")
    dist.init()
    if dist.world_size() > 1:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(dist.env_local_rank())
            ml.device = torch.device("cuda", dist.env_local_rank())
            kwargs.setdefault("device", dist.env_local_rank())
    _cache_embedding_batch(dist.shard(files), ml, workers, **kwargs)
    dist.barrier()


def embed_and_accumulate(directory: Union[str, Path], ml, workers: int = 8, compat: bool = True, **kwargs):
    """One pass over a dataset: embed every file on this rank's GPU, keep the (float16) frames in HBM long enough
    to fold them into running (n, sum x, sum x x^T), write the usual embedding cache, all-reduce the packed
    statistics once (RCCL over xGMI) and let rank 0 store ``<dir>/stats/<model>/{mu,cov}.npy`` -- the cache
    ``FrechetAudioDistance.load_stats`` picks up, so a following ``score`` never re-reads the .npy files.

    compat=True adds the sum-reducible per-file mean terms (``utils.per_file_mean_terms``) so that the result is
    what the reference's online path gives, including its per-file float16 rounding of the means (which is worth up to
    3e-4 of the FAD for files of a few frames); compat=False is the plain raw-moment estimate.
    Returns (mu, cov) on every rank.
    """
    import numpy as np
    from . import hip
    from .utils import combine_online_statistics, per_file_mean_terms
    directory = Path(directory)
    files = sorted(p for p in directory.glob("*.*") if p.is_file())
    dist.init()
    dev_index = dist.env_local_rank() if dist.world_size() > 1 else 0
    if dist.world_size() > 1:
        import torch
        if torch.cuda.is_available():
            torch.cuda.set_device(dev_index)
            ml.device = torch.device("cuda", dev_index)
            kwargs.setdefault("device", dev_index)
    d = ml.num_features
    acc = hip.Moments(d, dev_index)
    file_sums = [] if compat else None
    _cache_embedding_batch(dist.shard(files), ml, workers, moments=acc, file_sums=file_sums, **kwargs)
    dist.allreduce_moments([acc])                                      # the one collective of the data path
    packed = acc.export()
    acc.close()
    if compat:
        sizes = np.array([f[0] for f in file_sums], dtype=np.int64)
        sums = np.stack([f[1] for f in file_sums]) if file_sums else np.zeros((0, d))
        dtype = file_sums[0][2] if file_sums else np.float16
        wsum, within, between, n_short, n_empty = per_file_mean_terms(sums, sizes, dtype, dev_index)
        extra = np.concatenate([wsum, within.reshape(-1), between.reshape(-1), [n_short, n_empty]])
        extra = dist.allreduce_packed(extra)                           # D + 2 D^2 + 2 doubles, once per dataset
        wsum, within, between = extra[:d], extra[d:d + d * d].reshape(d, d), extra[d + d * d:d + 2 * d * d].reshape(d, d)
        mu, cov = combine_online_statistics(packed, wsum, within, between, int(round(extra[-2])), int(round(extra[-1])))
        n = int(round(packed[0]))
    else:
        n = int(round(packed[0]))
        assert n >= 2, f"FAD requires at least two embedding window frames, you have {n}."
        sx, sxx = packed[1:1 + d], packed[1 + d:].reshape(d, d)
        mu, cov = sx / n, (sxx - np.outer(sx, sx) / n) / (n - 1)
    if dist.rank() == 0:
        out = directory / "stats" / ml.name
        out.mkdir(parents=True, exist_ok=True)
        np.save(out / "mu.npy", mu)
        np.save(out / "cov.npy", cov)
    dist.barrier()
    log.info(f"[{ml.name}] {n} frames from {len(files)} files accumulated on {dist.world_size()} GPU(s)")
    return mu, cov
