"""Cache-path conventions, small host helpers and the online dataset statistics.

Host-side mirror of fadtk/utils.py (same names, argument meaning and results); the arithmetic of
``calculate_embd_statistics_online`` runs in libfad_hip.so.
"""
from __future__ import annotations

import logging
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path
from typing import Callable, Iterable, List, Sequence, Tuple, Union

import numpy as np

PathLike = Union[str, Path]
log = logging.getLogger("fadtk_amd")

# Host blocks handed to the GPU in one call (bytes of frames).  Large enough that the per-call
# launch + PCIe latency is amortised, small enough to pipeline file reading with compute.
STAGE_BYTES = 256 << 20


# -- replacements for the un-vendored hypy_utils helpers the reference leans on -----------------
def tmap(fn: Callable, items: Sequence, desc: str = "", max_workers: int = 8) -> list:
    """Ordered thread map (hypy_utils.tqdm_utils.tmap as used at fad.py:229, 387)."""
    items = list(items)
    if max_workers <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(max_workers=max_workers) as ex:
        return list(ex.map(fn, items))


def tq(it: Iterable, desc: str = ""):
    try:
        from tqdm import tqdm
        return tqdm(it, desc=desc)
    except Exception:       # noqa: BLE001
        return it


def write(path: PathLike, text: str):
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_text(text)


def find_sox_formats(sox_path: str) -> List[str]:
    """File formats SoX can read (fadtk/utils.py:49-57); empty list when SoX is absent."""
    try:
        out = subprocess.check_output((sox_path, "-h"), stderr=subprocess.DEVNULL).decode()
        marker = "AUDIO FILE FORMATS: "
        start = out.index(marker) + len(marker)
        return out[start:out.index("\n", start)].split()
    except Exception:       # noqa: BLE001
        return []


def write_stats_cache(cache_dir: PathLike, mu: np.ndarray, cov: np.ndarray) -> None:
    """<dir>/stats/<model>/{mu,cov}.npy (fad.py:286-289), written so that a concurrent reader never sees a
    half-written file: temporary names first, cov.npy renamed last-but-one, the directory's mu.npy last."""
    import os
    cache_dir = Path(cache_dir)
    cache_dir.mkdir(parents=True, exist_ok=True)
    tag = f".tmp{os.getpid()}"
    for name, arr in (("cov.npy", cov), ("mu.npy", mu)):
        tmp = cache_dir / (name + tag)
        with open(tmp, "wb") as fh:
            np.save(fh, arr)
        os.replace(tmp, cache_dir / name)


def get_cache_embedding_path(model: str, audio_dir: PathLike) -> Path:
    """<dir>/embeddings/<model>/<stem>.npy for an audio file <dir>/<stem>.<ext> (fadtk/utils.py:60-68)."""
    audio = Path(audio_dir)
    return audio.parent / "embeddings" / model / audio.with_suffix(".npy").name


# -- online statistics ------------------------------------------------------------------------
def _round_like(values: np.ndarray, dtype: np.dtype) -> np.ndarray:
    """Round float64 means the way ``np.mean`` returns them for ``dtype`` inputs."""
    if dtype == np.float16:
        return values.astype(np.float32).astype(np.float16).astype(np.float64)
    if dtype == np.float32:
        return values.astype(np.float32).astype(np.float64)
    return values


_DT_CODES = {np.dtype(np.float16): 0, np.dtype(np.float32): 2, np.dtype(np.float64): 3}
import os as _os
_SIDE_STREAM = _os.environ.get("FAD_ONLINE_SIDE_STREAM") == "1"


class OnlineStats:
    """Dataset statistics from per-file frame matrices, fed in groups -- the sum-reducible form of what
    ``calculate_embd_statistics_online`` (fadtk/utils.py:19-46) merges file by file.

    Four GPU accumulators: the raw moments of all frames, and -- for the reference's quirk that every per-file mean is
    rounded to the file's dtype (float16) before the merge -- the rows sqrt(n_f) m_f, sqrt(n_f) m~_f and n_f m~_f
    (``fad_moments_update_file_means``).  ``buffers`` is a ``dist.SharedStats`` when several ranks feed shards of one
    dataset: their sum is then ONE all-reduce.  Nothing but per-group column sums [files x D] is kept besides."""

    def __init__(self, d: int, device: int = 0, compat: bool = True, shared=None, ref_means: bool = True):
        from .hip import Moments
        self.d, self.device, self.compat = int(d), int(device), compat
        # per-file means as np.mean forms them (utils.py:16: a float32 running sum per file, rounded to the file's dtype) -- a second walk
        # over the group's rows (fad_moments_update_segmented_ref); False: the rounded exact means (one ulp off in ~0.3 % of the columns
        # of long files whose frames carry an offset; no second walk)
        self.ref_means = bool(ref_means)
        self.shared = shared
        if shared is not None:          # 4 accumulators when compat, else 1
            self.frames = shared.moments[0]
            self.exact, self.rounded, self.weighted = shared.moments[1:4] if compat else (None, None, None)
        else:
            self.frames = Moments(d, device)
            self.exact, self.rounded, self.weighted = (Moments(d, device) for _ in range(3)) if compat else (None, None, None)
        self._side = None         # side stream of the per-file mean terms (device inputs), see add_group
        self.n_short = 0          # files with fewer than two frames (np.cov -> NaN)
        self.n_empty = 0
        self.n_files = 0

    def add_group(self, rows, sizes: Sequence[int]) -> None:
        """``rows`` = the frames of ``len(sizes)`` files stored back to back (numpy on the host, or a torch CUDA tensor that
        stays in HBM); one dtype per group."""
        sizes = np.asarray(sizes, dtype=np.int64)

        def count():                                 # the file counters move only once the GPU calls of the group went through
            self.n_files += len(sizes)
            self.n_short += int((sizes < 2).sum())
            self.n_empty += int((sizes < 1).sum())

        if int(sizes.sum()) == 0:
            count()
            return
        offs = np.concatenate([[0], np.cumsum(sizes)])
        if not self.compat:
            self.frames.update(rows)
            count()
            return
        on_dev = type(rows).__module__.split(".")[0] == "torch" and rows.is_cuda
        runs = None
        if self.ref_means:
            got = self.frames.update_segmented(rows, offs, want_sums=True, sums_on_device=on_dev, want_runsums=True)
            sums, runs = got if isinstance(got, tuple) else (got, None)
        else:
            sums = self.frames.update_segmented(rows, offs, want_sums=True, sums_on_device=on_dev)
        if on_dev:
            import torch
            code = {torch.float16: 0, torch.bfloat16: 1, torch.float32: 2}.get(rows.dtype, 3)
        else:
            code = _DT_CODES.get(np.asarray(rows).dtype, 3)
        if on_dev and _SIDE_STREAM:
            # (opt-in, FAD_ONLINE_SIDE_STREAM=1.  Measured at config 4, profiles/r04d_c4.txt: the 41 us of these kernels disappear from
            # the stream, but the HBM-bound tile kernel they run beside loses 80 us -- their workgroups take LDS a second tile
            # workgroup of the CU needs -- so the default keeps them in line.)
            # The three small accumulators of the per-file mean terms (rows + a float64 tile kernel + its reduce: ~40 us per group)
            # depend on nothing but this group's per-file sums: on a stream of their own they run under the NEXT group's tile
            # kernel instead of between two of them.  join() orders them in front of whatever reads the accumulators.
            if self._side is None:
                self._side = torch.cuda.Stream(device=rows.device)
            main = torch.cuda.current_stream(rows.device)
            ready = torch.cuda.Event(); ready.record(main)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ready)
                type(self.frames).update_file_means(self.exact, self.rounded, self.weighted, sums, sizes, code, runs)
                sums.record_stream(self._side)
                if runs is not None:
                    runs.record_stream(self._side)
        else:
            type(self.frames).update_file_means(self.exact, self.rounded, self.weighted, sums, sizes, code, runs)
        count()

    def join(self):
        """Order the side stream's work (per-file mean terms of device groups) in front of the current stream: call before the
        accumulators are read, exported or all-reduced (pieces() / finish() do)."""
        if self._side is not None:
            import torch
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)

    def pieces(self):
        """(packed frames, sum_f n_f m~_f, sum_f n_f m_f m_f^T, sum_f n_f m~_f m~_f^T) as host arrays."""
        d = self.d
        self.join()
        packed = self.frames.export()
        if not self.compat:
            return packed, None, None, None
        return (packed, self.weighted.export()[1:1 + d], self.exact.export()[1 + d:].reshape(d, d),
                self.rounded.export()[1 + d:].reshape(d, d))

    def finish(self) -> Tuple[np.ndarray, np.ndarray]:
        """(mu, Sigma) as the reference's sequential merge leaves them (compat) or the plain raw-moment estimate."""
        d = self.d
        packed, wsum, within, between = self.pieces()
        total = int(round(packed[0]))
        if not self.compat:
            if total < 1:
                return np.full(d, np.nan), np.zeros((d, d))
            sum_x, sum_xx = packed[1:1 + d], packed[1 + d:].reshape(d, d)
            mu = sum_x / total
            if total < 2:
                return mu, np.zeros((d, d))                                   # utils.py:42-43
            return mu, (sum_xx - np.outer(sum_x, sum_x) / total) / (total - 1)
        return combine_online_statistics(packed, wsum, within, between, self.n_short, self.n_empty)

    def close(self):
        if self.shared is None:
            for m in (self.frames, self.exact, self.rounded, self.weighted):
                if m is not None:
                    m.close()


def _groups(blocks: Iterable[np.ndarray], limit_bytes: int):
    """Consecutive blocks of ONE dtype whose frames total <= limit_bytes (at least one block per group)."""
    group, nbytes, dt = [], 0, None
    for b in blocks:
        b = np.asarray(b)
        if group and (b.dtype != dt or nbytes + b.nbytes > limit_bytes):
            yield group
            group, nbytes = [], 0
        group.append(b); nbytes += b.nbytes; dt = b.dtype
    if group:
        yield group


def dataset_statistics(blocks: Iterable[np.ndarray], compat: bool = True, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Dataset (mu, Sigma) from per-file frame matrices, as calculate_embd_statistics_online
    (fadtk/utils.py:19-46) computes it -- but in ONE GPU pass over raw moments.  ``blocks`` may be a generator: the
    frames are consumed in groups of <= STAGE_BYTES, so host memory holds one group, not the dataset.

    compat=True reproduces the reference's quirk that every per-file mean is rounded to the file's
    dtype (float16) before the merge:  Sigma = (W + B~) / (N-1) with the within-file scatter W from
    exact means and the between-file scatter B~ from the rounded means (algebraically what the
    sequential merge of utils.py:36-40 yields).  compat=False is the plain (sum, sum xx^T) estimate.
    """
    it = iter(blocks)
    first = next(it, None)
    if first is None:
        raise AssertionError("No files provided")
    first = np.asarray(first)
    d = int(first.shape[-1])

    def chain():
        yield first
        yield from it

    acc = OnlineStats(d, device, compat)
    try:
        for group in _groups(chain(), STAGE_BYTES):
            sizes = [g.shape[0] for g in group]
            rows = np.concatenate(group, axis=0) if len(group) > 1 else group[0]
            acc.add_group(rows, sizes)
        return acc.finish()
    finally:
        acc.close()


def per_file_mean_terms(seg_sums: np.ndarray, sizes: np.ndarray, dtype, device: int = 0):
    """What the reference's per-file float16 means (utils.py:16) change, as additive (sum-reducible) pieces:
    -> (sum_f n_f m~_f  [D],  sum_f n_f m_f m_f^T  [D x D] with exact means,  sum_f n_f m~_f m~_f^T with rounded means,
        number of files with < 2 rows, number of empty files).  Computed on the GPU (fad_moments_update_file_means)."""
    from .hip import Moments
    seg_sums = np.asarray(seg_sums, dtype=np.float64)
    d = seg_sums.shape[1]
    sizes = np.asarray(sizes, dtype=np.int64)
    if len(sizes) == 0:
        z = np.zeros((d, d))
        return np.zeros(d), z, z.copy(), 0, 0
    with Moments(d, device) as e, Moments(d, device) as r, Moments(d, device) as w:
        Moments.update_file_means(e, r, w, seg_sums, sizes, _DT_CODES.get(np.dtype(dtype), 3))
        within_corr = e.export()[1 + d:].reshape(d, d)
        between = r.export()[1 + d:].reshape(d, d)
        wsum_ref = w.export()[1:1 + d]
    return wsum_ref, within_corr, between, int((sizes < 2).sum()), int((sizes < 1).sum())


def combine_online_statistics(packed: np.ndarray, wsum_ref, within_corr, between, n_short: int, n_empty: int):
    """(mu, Sigma) exactly as the sequential merge of utils.py:36-45 leaves them, from sum-reducible pieces:
    Sigma = (W + B~) / (N - 1),  W = sum xx^T - sum_f n_f m_f m_f^T (within-file scatter, exact means),
    B~ = sum_f n_f m~_f m~_f^T - N mu~ mu~^T (between-file scatter of the float16-rounded means), mu~ = sum n_f m~_f / N."""
    d = wsum_ref.shape[0]
    total = int(round(packed[0]))
    sum_xx = packed[1 + d:].reshape(d, d)
    if total < 1:
        return np.full(d, np.nan), np.zeros((d, d))
    mu = wsum_ref / total
    if total < 2:
        return mu, np.zeros((d, d))                                     # utils.py:42-43
    if n_short > 0:
        # np.cov of a one-row (or empty) file is NaN and poisons the merged scatter
        return (np.full(d, np.nan) if n_empty > 0 else mu), np.full((d, d), np.nan)
    scatter = (sum_xx - within_corr) + (between - total * np.outer(mu, mu))
    return mu, scatter / (total - 1)


def calculate_embd_statistics_online(files: List[PathLike], compat: bool = True, device: int = 0,
                                     workers: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """(mu, Sigma) of all frames stored in ``files`` (.npy, [n_frames x n_features]) -- fadtk/utils.py:19-46."""
    assert len(files) > 0, "No files provided"

    def stream():                                    # files are read `workers` ahead of the GPU, never all at once
        files_l = list(files)
        with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
            window = 4 * max(1, workers)
            pending = [ex.submit(np.load, f) for f in files_l[:window]]
            nxt = len(pending)
            while pending:
                blk = pending.pop(0).result()
                if nxt < len(files_l):
                    pending.append(ex.submit(np.load, files_l[nxt])); nxt += 1
                yield blk
    return dataset_statistics(stream(), compat=compat, device=device)
