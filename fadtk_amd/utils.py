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


def dataset_statistics(blocks: Sequence[np.ndarray], compat: bool = True, device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """Dataset (mu, Sigma) from per-file frame matrices, as calculate_embd_statistics_online
    (fadtk/utils.py:19-46) computes it -- but in ONE GPU pass over raw moments.

    compat=True reproduces the reference's quirk that every per-file mean is rounded to the file's
    dtype (float16) before the merge:  Sigma = (W + B~) / (N-1) with the within-file scatter W from
    exact means and the between-file scatter B~ from the rounded means (algebraically what the
    sequential merge of utils.py:36-40 yields).  compat=False is the plain (sum, sum xx^T) estimate.
    """
    from .hip import Moments
    if len(blocks) == 0:
        raise AssertionError("No files provided")
    d = int(blocks[0].shape[-1])
    dtype = np.asarray(blocks[0]).dtype
    sizes = np.array([b.shape[0] for b in blocks], dtype=np.int64)
    total = int(sizes.sum())

    with Moments(d, device) as acc:
        seg_sums = []
        i = 0
        while i < len(blocks):                       # host blocks of <= STAGE_BYTES, one GPU call each
            j, nbytes = i, 0
            while j < len(blocks) and (j == i or nbytes + blocks[j].nbytes <= STAGE_BYTES):
                nbytes += blocks[j].nbytes
                j += 1
            group = [np.asarray(b) for b in blocks[i:j]]
            same = all(g.dtype == group[0].dtype for g in group)
            rows = np.concatenate(group if same else [g.astype(np.float64) for g in group], axis=0)
            offs = np.concatenate([[0], np.cumsum(sizes[i:j])])
            if rows.shape[0] > 0:
                seg_sums.append(acc.update_segmented(rows, offs, want_sums=True))
            else:
                seg_sums.append(np.zeros((j - i, d)))
            i = j
        packed = acc.export()
    seg_sums = np.concatenate(seg_sums, axis=0)
    n = packed[0]
    sum_x, sum_xx = packed[1:1 + d], packed[1 + d:].reshape(d, d)

    if total < 1:
        return np.full(d, np.nan), np.zeros((d, d))
    if not compat:
        mu = sum_x / n
        if total < 2:
            return mu, np.zeros((d, d))                                   # utils.py:42-43
        return mu, (sum_xx - np.outer(sum_x, sum_x) / n) / (n - 1)

    return finish_online_statistics(packed, seg_sums, sizes, dtype, device)


def per_file_mean_terms(seg_sums: np.ndarray, sizes: np.ndarray, dtype, device: int = 0):
    """What the reference's per-file float16 means (utils.py:16) change, as additive (sum-reducible) pieces:
    -> (sum_f n_f m~_f  [D],  sum_f n_f m_f m_f^T  [D x D] with exact means,  sum_f n_f m~_f m~_f^T with rounded means,
        number of files with < 2 rows, number of empty files).  The two matrices are raw moments of the rows
    sqrt(n_f) m_f, computed by the same GPU kernel."""
    from .hip import Moments
    d = seg_sums.shape[1]
    sizes = np.asarray(sizes, dtype=np.int64)
    ok = sizes > 0
    means = np.zeros_like(seg_sums)
    means[ok] = seg_sums[ok] / sizes[ok, None]
    means_ref = _round_like(means, np.dtype(dtype))
    w = sizes.astype(np.float64)
    wsum_ref = (means_ref * w[:, None]).sum(axis=0)
    root_w = np.sqrt(w)[:, None]
    if len(sizes) == 0:
        z = np.zeros((d, d))
        return wsum_ref, z, z.copy(), 0, 0
    with Moments(d, device) as a_exact, Moments(d, device) as a_ref:
        a_exact.update(np.ascontiguousarray(means * root_w))
        a_ref.update(np.ascontiguousarray(means_ref * root_w))
        within_corr = a_exact.export()[1 + d:].reshape(d, d)
        between = a_ref.export()[1 + d:].reshape(d, d)
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


def finish_online_statistics(packed, seg_sums, sizes, dtype, device: int = 0):
    return combine_online_statistics(packed, *per_file_mean_terms(seg_sums, sizes, dtype, device))


def calculate_embd_statistics_online(files: List[PathLike], compat: bool = True, device: int = 0,
                                     workers: int = 8) -> Tuple[np.ndarray, np.ndarray]:
    """(mu, Sigma) of all frames stored in ``files`` (.npy, [n_frames x n_features]) -- fadtk/utils.py:19-46."""
    assert len(files) > 0, "No files provided"
    blocks = tmap(np.load, files, desc="Loading embeddings", max_workers=workers)
    return dataset_statistics(blocks, compat=compat, device=device)
