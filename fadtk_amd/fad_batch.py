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


def _cache_embedding_batch(fs, ml, workers: int = 8, moments=None, **kwargs):
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
                    moments.update(np.load(cache))
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


def embed_and_accumulate(directory: Union[str, Path], ml, workers: int = 8, **kwargs):
    """One pass over a dataset: embed every file on this rank's GPU, keep the (float16) frames in HBM long enough
    to fold them into running (n, sum x, sum x x^T), write the usual embedding cache, all-reduce the packed
    statistics once (RCCL over xGMI) and let rank 0 store ``<dir>/stats/<model>/{mu,cov}.npy`` -- the cache
    ``FrechetAudioDistance.load_stats`` picks up, so a following ``score`` never re-reads the .npy files.

    This is the plain raw-moment estimate (what ``calc_embd_statistics`` gives on the concatenated frames); the
    reference's online path differs from it only by its per-file float16 mean rounding (<= 5e-7 relative FAD).
    Returns (mu, cov) on every rank.
    """
    import numpy as np
    from . import hip
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
    acc = hip.Moments(ml.num_features, dev_index)
    _cache_embedding_batch(dist.shard(files), ml, workers, moments=acc, **kwargs)
    dist.allreduce_moments([acc])
    mu, cov, n = acc.finalize(ddof=1)
    acc.close()
    if dist.rank() == 0:
        out = directory / "stats" / ml.name
        out.mkdir(parents=True, exist_ok=True)
        np.save(out / "mu.npy", mu)
        np.save(out / "cov.npy", cov)
    dist.barrier()
    log.info(f"[{ml.name}] {n} frames from {len(files)} files accumulated on {dist.world_size()} GPU(s)")
    return mu, cov
