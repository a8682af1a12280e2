"""One process per GPU: file/row sharding and the single reduce of the sufficient statistics.

fadtk's only parallelism is a spawn pool over files on cuda:0 that exchanges results through the
filesystem (fadtk/fad_batch.py:43-48).  Here every rank owns one GPU (LOCAL_RANK), embeds its own
contiguous shard of the files, accumulates local (n, sum x, sum x x^T) in HBM and ONE all-reduce of
the packed float64 statistics (RCCL over xGMI; 2.1 MB at D=512) combines them.  Raw moments are
sum-reducible, so no per-file merges cross ranks.  The same code runs under gloo on CPU tensors,
which is how the multi-rank logic is tested without GPUs.
"""
from __future__ import annotations

import os
from typing import List, Sequence

import numpy as np


def env_rank() -> int:
    return int(os.environ.get("RANK", "0"))


def env_world() -> int:
    return int(os.environ.get("WORLD_SIZE", "1"))


def env_local_rank() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def is_initialized() -> bool:
    try:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized()
    except Exception:       # noqa: BLE001
        return False


def rank() -> int:
    if is_initialized():
        import torch.distributed as dist
        return dist.get_rank()
    return 0


def world_size() -> int:
    if is_initialized():
        import torch.distributed as dist
        return dist.get_world_size()
    return 1


def _forced() -> bool:
    """FAD_DIST_FORCE=1: take the multi-rank code path (process group, collectives) even with ONE rank -- how the RCCL route
    of `--gpus N` is exercised on a single-GPU box (tests/test_gpu_dist.py)."""
    return os.environ.get("FAD_DIST_FORCE") == "1"


def _active() -> bool:
    """Do collectives have to run?  (several ranks, or a forced one-rank group)"""
    return is_initialized() and (world_size() > 1 or _forced())


def init(backend: str = None) -> bool:
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun).  No-op for a single process."""
    if (env_world() <= 1 and not (_forced() and "RANK" in os.environ)) or is_initialized():
        return is_initialized()
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:         # FAD_DIST_BACKEND=gloo: ranks that share one GPU (tests), or a CPU-only rendezvous
        backend = os.environ.get("FAD_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(env_local_rank())
        dist.init_process_group(backend, device_id=torch.device("cuda", env_local_rank()))
    else:
        dist.init_process_group(backend)
    return True


def shard(items: Sequence, r: int = None, w: int = None) -> list:
    """Contiguous shard ``r`` of ``w`` (np.array_split semantics, like fad_batch.py:43)."""
    r = rank() if r is None else r
    w = world_size() if w is None else w
    items = list(items)
    base, extra = divmod(len(items), w)
    start = r * base + min(r, extra)
    return items[start:start + base + (1 if r < extra else 0)]


def allreduce_packed(packed, device=None):
    """Sum a packed float64 statistics vector (numpy or torch) over all ranks; returns the same kind.
    RCCL reduces device tensors in place; under gloo (CPU collectives) a device tensor takes the host route."""
    if not _active():
        return packed
    import torch
    import torch.distributed as dist
    is_np = isinstance(packed, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(packed, dtype=np.float64)) if is_np else packed
    backend = dist.get_backend()
    if backend == "nccl" and not t.is_cuda:
        t = t.to(device if device is not None else torch.device("cuda", env_local_rank()))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    elif backend != "nccl" and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy() if is_np else t


class SharedStats:
    """``count`` GPU accumulators of dimension ``d`` (+ ``extra`` loose float64 words) living in ONE device buffer, so
    that the exchange of the data-parallel path -- the sum of all ranks' sufficient statistics -- is a single in-place
    all-reduce with no export/import copies (``fad_moments_bind``).  bench.py and ``embed_and_accumulate`` share it."""

    def __init__(self, d: int, count: int, device: int = 0, extra: int = 0):
        import torch
        from .hip import Moments
        self.d, self.device = int(d), int(device)
        plen = 1 + d + d * d
        self.stride = plen + (plen & 1)                     # every slice starts 16-byte aligned
        self.buffer = torch.zeros(self.stride * count + extra, dtype=torch.float64, device=torch.device("cuda", self.device))
        self.moments = []
        for i in range(count):
            m = Moments(d, self.device)
            m.bind(self.buffer[i * self.stride:i * self.stride + plen])
            self.moments.append(m)
        self.extra = self.buffer[self.stride * count:]

    def allreduce(self):
        """Sum the whole buffer over all ranks, in place (one collective)."""
        for m in self.moments:
            m.settle()                                      # a handle that saw no update since bind/reset still holds garbage
        allreduce_packed(self.buffer)

    def close(self):
        for m in self.moments:
            m.close()
        self.moments = []


def allreduce_moments(moments: Sequence) -> None:
    """In-place all-reduce of GPU accumulators (fadtk_amd.hip.Moments) that were NOT created through SharedStats:
    packs them into one buffer, one collective, unpacks.  Prefer SharedStats (no copies)."""
    if world_size() <= 1:
        return
    import torch
    lens = [m.packed_len for m in moments]
    buf = torch.empty(sum(lens), dtype=torch.float64, device=torch.device("cuda", moments[0].device))
    o = 0
    for m, n in zip(moments, lens):
        m.export_to(buf[o:o + n]); o += n
    allreduce_packed(buf)
    o = 0
    for m, n in zip(moments, lens):
        m.import_(buf[o:o + n]); o += n


def broadcast_object(obj, src: int = 0):
    """Rank ``src``'s python object on every rank (file lists: every rank must shard the SAME list)."""
    if not _active():
        return obj
    import torch.distributed as dist
    box = [obj if rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def gather_objects(obj) -> List:
    """All ranks' python objects in rank order (per-song score lists are tiny)."""
    if not _active():
        return [obj]
    import torch.distributed as dist
    out = [None] * world_size()
    dist.all_gather_object(out, obj)
    return out


def barrier():
    if _active():
        import torch.distributed as dist
        dist.barrier()
