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


def _cache_embedding_batch(fs, ml, workers: int = 8, **kwargs):
    """Embed a list of files on this process's GPU; audio decode runs ``workers`` files ahead."""
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
                continue
            log.info(f"Loading {f} using {ml.name}")
            try:
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
