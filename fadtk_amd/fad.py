"""FAD math and orchestration -- host-side mirror of fadtk/fad.py on top of libfad_hip.so.

Same public names, argument meaning, return dtypes and error behaviour as the reference
(`calc_embd_statistics`, `calc_frechet_distance`, `FrechetAudioDistance`, `FADInfResults`), so
callers and tests written against fadtk read the same.  All reductions over frames and the matrix
square root run on the GPU; numpy only carries host buffers and O(D) glue (dtype casts, traces).
"""
from __future__ import annotations

import logging
import os
import threading
import traceback
from pathlib import Path
from typing import NamedTuple, Union

import numpy as np

from . import _capi as K
from . import hip
from .utils import PathLike, calculate_embd_statistics_online, find_sox_formats, get_cache_embedding_path, tmap, tq, write

log = logging.getLogger("fadtk_amd")
sox_path = os.environ.get("SOX_PATH", "sox")
ffmpeg_path = os.environ.get("FFMPEG_PATH", "ffmpeg")


class FADInfResults(NamedTuple):
    score: float
    slope: float
    r2: float
    points: list


def _mean_dtype(dtype: np.dtype) -> np.dtype:
    """dtype of ``np.mean(x, axis=0)``: float16/32 stay, everything else is float64."""
    return dtype if dtype in (np.float16, np.float32) else np.dtype(np.float64)


_tls = threading.local()


def _thread_accumulator(d: int, device: int):
    """One GPU accumulator per (host thread, device, D), kept between calls: creating a handle allocates its packed statistics,
    the partial tiles and -- for host rows -- a staging area the size of the frame matrix, and destroying it frees them again
    (hipFree waits for the device): ~1 ms of the 3 ms a [100000 x 512] call took in round 2.  A handle serves one thread at a
    time (include/fad_hip.h), callers come from thread pools (fad.py:229, 387): hence thread-local; at most four are kept."""
    pool = getattr(_tls, "acc", None)
    if pool is None:
        pool = _tls.acc = {}
    acc = pool.get((device, d))
    if acc is None:
        if len(pool) >= 4:
            pool.pop(next(iter(pool))).close()
        acc = pool[(device, d)] = hip.Moments(d, device)
        acc.set_reference_mean(True)                   # mu as np.mean forms it (float32 running sum over the rows): calc_embd_statistics
    return acc


def _thread_inf_accumulators(d: int, device: int, count: int, ref_mean: bool):
    """The accumulators of FAD-inf's device route -- one for the baseline, `count` for the resamples of a launch -- kept per (host thread,
    device, D) like ``_thread_accumulator``: creating and freeing a handle costs 0.27 ms (its packed statistics, partial tiles, workspaces;
    hipFree waits for the device) and a call needs seventeen of them."""
    pool = getattr(_tls, "inf", None)
    if pool is None:
        pool = _tls.inf = {}
    ent = pool.get((device, d))
    if ent is None:
        if len(pool) >= 2:
            for h in pool.pop(next(iter(pool))):
                h.close()
        ent = pool[(device, d)] = [hip.Moments(d, device) for _ in range(1 + count)]
    for a in ent[1:]:
        a.set_reference_mean(bool(ref_mean))            # the resamples' means as np.mean forms them (fad.py:48: a float32 running sum)
    return ent[0], ent[1:]


def calc_embd_statistics(embd_lst, device: int = 0):
    """Mean and covariance of a frame matrix [N x D] (fadtk/fad.py:42-48), computed on the GPU.

    Returns (mu, cov) with numpy's dtypes: mu in the input's float dtype, formed the way np.mean forms it -- the rows added one
    after the other in float32, the quotient in float32, then the cast (``Moments.set_reference_mean``: bit for bit numpy's float16 /
    float32 mean, which for frames with a sizeable offset is NOT the rounded exact mean) --, cov float64 with ddof = 1 from exact
    float64 sums.
    """
    n = embd_lst.shape[0]
    assert n >= 2, (f"FAD requires at least two embedding window frames, you have {tuple(embd_lst.shape)}."
                    " (This probably means that your audio is too short)")
    acc = _thread_accumulator(int(embd_lst.shape[1]), device)
    acc.reset()
    acc.update(embd_lst)
    mu, cov, _ = acc.finalize(ddof=1)
    acc.release_inputs()                               # (finalize synchronised: the caller's tensor / a large staging area are not kept)
    try:
        in_dtype = embd_lst.dtype if isinstance(embd_lst, np.ndarray) else \
            np.dtype(str(embd_lst.dtype).replace("torch.", ""))
    except TypeError:                                  # torch.bfloat16 has no numpy twin
        in_dtype = np.dtype(np.float32)
    out_dtype = _mean_dtype(in_dtype) if in_dtype.kind == "f" else np.dtype(np.float64)
    if out_dtype == np.float16:
        mu = mu.astype(np.float32).astype(np.float16)      # numpy sums fp16 in fp32, then casts
    elif out_dtype == np.float32:
        mu = mu.astype(np.float32)
    return mu, cov


def calc_frechet_distance(mu1, cov1, mu2, cov2, eps: float = 1e-6, device: int = 0):
    """Frechet distance between N(mu1, cov1) and N(mu2, cov2) (fadtk/fad.py:51-120).

        d^2 = ||mu1 - mu2||^2 + Tr(cov1) + Tr(cov2) - 2 Tr sqrt(cov1 cov2)

    Tr sqrt(cov1 cov2) comes from the GPU (Newton-Schulz, see csrc/frechet.hip and csrc/frechet_f64.hip); the
    O(D) terms are formed here with numpy so that the reference's dtype behaviour carries over
    (float16 means give a float16 ``diff.dot(diff)``).  Returns ``np.float64``.
    Raises AssertionError on shape mismatch (fad.py:78-81) and ValueError when the product has no
    real root / is not finite (fad.py:102-106).
    """
    mu1 = np.atleast_1d(mu1)
    mu2 = np.atleast_1d(mu2)
    cov1 = np.atleast_2d(cov1)
    cov2 = np.atleast_2d(cov2)
    assert mu1.shape == mu2.shape, \
        f"Training and test mean vectors have different lengths ({mu1.shape} vs {mu2.shape})"
    assert cov1.shape == cov2.shape, \
        f"Training and test covariances have different dimensions ({cov1.shape} vs {cov2.shape})"
    assert cov1.shape == (mu1.shape[0], mu1.shape[0]), \
        f"Covariance shape {cov1.shape} does not match the mean length {mu1.shape[0]}"

    gap = mu1 - mu2
    _, diag = hip.frechet(mu1, cov1, mu2, cov2, eps=eps, device=device)
    if diag["used_eps"]:
        log.info("fid calculation produces singular product; adding %s to diagonal of cov estimates", eps)
    return np.float64(gap.dot(gap) + np.trace(cov1) + np.trace(cov2) - 2 * diag["tr_sqrt"])


class FrechetAudioDistance:
    """Drop-in for fadtk.FrechetAudioDistance (fadtk/fad.py:123-395)."""
    loaded = False

    def __init__(self, ml, audio_load_worker: int = 8, load_model: bool = True, device: int = 0):
        self.ml = ml
        self.audio_load_worker = audio_load_worker
        self.sox_formats = find_sox_formats(sox_path)
        self.device_index = device
        if load_model:
            self.ml.load_model()
            self.loaded = True
        try:
            import torch
            torch.autograd.set_grad_enabled(False)
        except Exception:       # noqa: BLE001
            pass

    # ------------------------------------------------------------------ audio -> embeddings
    def load_audio(self, f: PathLike):
        """Normalise one audio file to mono PCM16 WAV at the model's rate, cached under
        <dir>/convert/<sr>/<stem>.wav (fad.py:139-186), and hand it to the loader."""
        from .audio import convert_to_model_rate
        f = Path(f)
        cache_dir = f.parent / "convert" / str(self.ml.sr)
        new = (cache_dir / f.name).with_suffix(".wav")
        if not new.exists():
            cache_dir.mkdir(parents=True, exist_ok=True)
            convert_to_model_rate(f, new, self.ml.sr, device=self.device_index)
        return self.ml.load_wav(new)

    def cache_embedding_file(self, audio_dir: PathLike):
        """Embed one audio file and store <dir>/embeddings/<model>/<stem>.npy (fad.py:188-201)."""
        cache = get_cache_embedding_path(self.ml.name, audio_dir)
        if cache.exists():
            return
        wav = self.load_audio(audio_dir)
        embd = self.ml.get_embedding(wav)
        cache.parent.mkdir(parents=True, exist_ok=True)
        np.save(cache, embd)

    def read_embedding_file(self, audio_dir: PathLike):
        cache = get_cache_embedding_path(self.ml.name, audio_dir)
        assert cache.exists(), f"Embedding file {cache} does not exist, please run cache_embedding_file first."
        return np.load(cache)

    def load_embeddings(self, dir: PathLike, max_count: int = -1, concat: bool = True):
        files = list(Path(dir).glob("*.*"))
        log.info(f"Loading {len(files)} audio files from {dir}...")
        return self._load_embeddings(files, max_count=max_count, concat=concat)

    def _load_embeddings(self, files, max_count: int = -1, concat: bool = True):
        if len(files) == 0:
            raise ValueError("No files provided")
        if max_count == -1:
            embd_lst = tmap(self.read_embedding_file, files, desc="Loading audio files...",
                            max_workers=self.audio_load_worker)
        else:                                        # stop once more than max_count frames are in (fad.py:231-237)
            embd_lst, seen = [], 0
            for f in tq(files, "Loading files"):
                embd_lst.append(self.read_embedding_file(f))
                seen += embd_lst[-1].shape[0]
                if seen > max_count:
                    break
        if concat:
            return np.concatenate(embd_lst, axis=0)
        return embd_lst, files

    # ------------------------------------------------------------------ statistics
    def load_stats(self, path: PathLike, collective: bool = False):
        """Resolve a dataset spec to (mu, cov) in the reference's order (fad.py:245-290):
        bundled ``stats/<name>.npz`` -> an .npz file with ``<model>.mu/.cov`` -> the cached
        ``<dir>/stats/<model>/{mu,cov}.npy`` -> compute from ``<dir>/embeddings/<model>/*.npy``
        on the GPU and write that cache (float64).

        ``collective=False`` (the default, and what ``score`` / ``score_inf`` use) runs no collective at all: under
        ``--gpus N`` only rank 0 scores, the other ranks have left by then.  ``collective=True`` is for callers that
        EVERY rank of the job goes through (``score_individual``): rank 0 decides whether the cache is there, computes and
        writes it if not (atomic renames), everybody else waits at the barrier and then reads it."""
        if isinstance(path, str):
            bundled = Path(__file__).parent / "stats" / (path.lower() + ".npz")
            if bundled.exists():
                path = bundled
        path = Path(path)

        if path.is_file():
            log.info(f"Loading embedding statistics from {path}...")
            with np.load(path) as data:
                k_mu, k_cov = f"{self.ml.name}.mu", f"{self.ml.name}.cov"
                if k_mu not in data or k_cov not in data:
                    raise ValueError(f"FAD statistics file {path} doesn't contain data for model {self.ml.name}")
                return data[k_mu], data[k_cov]

        from . import dist
        from .utils import write_stats_cache
        cache_dir = path / "stats" / self.ml.name
        emb_dir = path / "embeddings" / self.ml.name

        def cached():
            return (cache_dir / "mu.npy").exists() and (cache_dir / "cov.npy").exists()

        together = collective and dist.world_size() > 1
        leader = dist.rank() == 0 or not together
        have = dist.broadcast_object(cached() if dist.rank() == 0 else None) if together else cached()
        if have:
            log.info(f"Embedding statistics is already cached for {path}, loading...")
            return np.load(cache_dir / "mu.npy"), np.load(cache_dir / "cov.npy")

        if not path.is_dir():
            log.error(f"The dataset you want to use ({path}) is not a directory nor a file.")
            exit(1)

        if leader:
            log.info(f"Loading embedding files from {path}...")
            mu, cov = calculate_embd_statistics_online(list(emb_dir.glob("*.npy")), device=self.device_index,
                                                       workers=self.audio_load_worker)
            log.info("> Embeddings statistics calculated.")
            write_stats_cache(cache_dir, mu, cov)
        if together:
            dist.barrier()
            if not leader:
                mu, cov = np.load(cache_dir / "mu.npy"), np.load(cache_dir / "cov.npy")
        return mu, cov

    # ------------------------------------------------------------------ scores
    def score(self, baseline: PathLike, eval: PathLike):
        """One FAD value between a baseline and an eval set (fad.py:292-302)."""
        mu_bg, cov_bg = self.load_stats(baseline)
        mu_eval, cov_eval = self.load_stats(eval)
        return calc_frechet_distance(mu_bg, cov_bg, mu_eval, cov_eval, device=self.device_index)

    def score_inf(self, baseline: PathLike, eval_files, steps: int = 25, min_n: int = 500, raw: bool = False):
        """FAD-infinity: FAD at ``steps`` resampled sizes, extrapolated to 1/n -> 0 (fad.py:304-351).
        Resampling uses numpy's global RNG exactly like the reference (seed it to reproduce)."""
        log.info(f"Calculating FAD-inf for {self.ml.name}...")
        mu_base, cov_base = self.load_stats(baseline)
        if all(Path(f).suffix == ".npy" for f in eval_files):
            embeds = np.concatenate([np.load(f) for f in eval_files], axis=0)
        else:
            embeds = self._load_embeddings(eval_files, concat=True)
        max_n = len(embeds)
        ns = [int(n) for n in np.linspace(min_n, max_n, steps)]

        # numpy's global RNG is drawn in the reference's order (one choice() per point, fad.py:333) before anything runs
        picks = [np.random.choice(embeds.shape[0], size=n, replace=True) for n in ns]
        values = None
        try:                                          # (both routes run on the HIP library; only WHERE the frames live differs)
            values = self._score_inf_points_on_device(mu_base, cov_base, embeds, picks)
        except ImportError as e:                      # no torch to hold the frames in HBM: host arrays, one point at a time
            log.info(f"FAD-inf: device route not available ({type(e).__name__}), scoring point by point")
            values = None
        except RuntimeError as e:                     # not enough HBM for frames + resamples: torch's OutOfMemoryError or the library's
            if not K.is_out_of_memory(e):             # FadOutOfMemory, by TYPE -- nothing else: a failure of the library itself must not be scored twice
                raise
            log.warning(f"FAD-inf: device route ran out of memory ({e}), scoring point by point")
            values = None
        if values is None:
            values = self._score_inf_points_sequential(mu_base, cov_base, embeds, picks)
        results = [[n, v] for n, v in zip(ns, values)]

        ys = np.array(results)
        xs = 1 / np.array(ns)
        slope, intercept = np.polyfit(xs, ys[:, 1], 1)
        fit = slope * xs + intercept
        r2 = 1 - np.sum((ys[:, 1] - fit) ** 2) / np.sum((ys[:, 1] - np.mean(ys[:, 1])) ** 2)
        return FADInfResults(score=intercept, slope=slope, r2=r2, points=results)

    def _score_inf_points_sequential(self, mu_base, cov_base, embeds, picks):
        """One point after the other through the public functions, as the reference does it (fad.py:333-341): every point gathers
        on the host, crosses PCIe, and brings (mu, Sigma) back before its distance is taken.  Kept as the fallback of the batched
        route and as what bench.py compares it with."""
        out = []
        for idx in tq(picks, desc="Calculating FAD-inf"):
            mu_eval, cov_eval = calc_embd_statistics(embeds[idx], device=self.device_index)
            out.append(calc_frechet_distance(mu_base, cov_base, mu_eval, cov_eval, device=self.device_index))
        return out

    def _score_inf_points_on_device(self, mu_base, cov_base, embeds, picks):
        """All points of FAD-inf with the frames resident in HBM: gather-with-replacement on the device, the moments of up to eight
        resamples per launch (``fad_moments_update_multi``), every distance straight from the accumulators with its square-root
        chain in flight (``fad_frechet_from_moments_begin``) -- no (mu, Sigma) crosses PCIe.  The baseline's (mu, Sigma) enter as
        the sufficient statistics of a two-frame pseudo-dataset with exactly that mean and covariance; ``mean_dtype`` asks for
        the reference's arithmetic of the mean term here: a float64 baseline mean against np.mean of the resampled frames,
        which keeps their dtype (float16)."""
        import torch
        if not torch.cuda.is_available() or embeds.dtype not in (np.float16, np.float32, np.float64):
            return None
        dev = torch.device("cuda", self.device_index)
        d = int(embeds.shape[1])
        rows = torch.from_numpy(np.ascontiguousarray(embeds)).to(dev)
        mu_b = np.asarray(mu_base, dtype=np.float64).reshape(-1)
        cov_b = np.asarray(cov_base, dtype=np.float64)
        packed = np.concatenate([[2.0], 2.0 * mu_b, (cov_b + 2.0 * np.outer(mu_b, mu_b)).reshape(-1)])
        code = {np.dtype(np.float16): hip.K.FAD_F16, np.dtype(np.float32): hip.K.FAD_F32}.get(embeds.dtype)
        mean_dtype = -1 if code is None else (hip.K.FAD_MEAN_SECOND_ONLY | code)
        # Resamples of fewer than 16 rows per column have (nearly) rank-deficient covariances: the distance then moves with the
        # SQUARE ROOT of a perturbation of the moments, and the library sums such inputs exactly in float64 -- but it picks the
        # kernel per launch, by the launch's largest set.  So short and long resamples never share a launch.
        order = [k for k in range(len(picks)) if picks[k].size < 16 * d] + [k for k in range(len(picks)) if picks[k].size >= 16 * d]
        n_short = sum(1 for idx in picks if idx.size < 16 * d)
        values = [None] * len(picks)
        with torch.cuda.device(dev):
            # ONE upload of all points' indices (int32: 4 bytes per resampled frame); the gathers themselves are never materialised --
            # fad_moments_update_multi_indexed reads rows[idx] straight into the moments kernels (round 5 built 1.3 GB of copies with
            # index_select and read them again: 11.9 ms per call at config 3)
            flat = torch.from_numpy(np.concatenate([np.asarray(p_, dtype=np.int32) for p_ in picks])).to(dev)
            starts = np.concatenate([[0], np.cumsum([p_.size for p_ in picks])])
            idx_dev = [flat[int(starts[k]):int(starts[k + 1])] for k in range(len(picks))]
            base, accs = _thread_inf_accumulators(d, self.device_index, 16, ref_mean=code is not None)
            base.reset(); base.import_(packed)
            pos = 0
            while pos < len(order):
                stop = n_short if pos < n_short else len(order)
                group = order[pos:min(pos + 16, stop)]                     # (a launch of the indexed update takes sixteen sets)
                pos += len(group)
                for a in accs[:len(group)]:
                    a.reset()
                hip.Moments.update_multi_indexed(accs[:len(group)], rows, [idx_dev[k] for k in group])
                # the group's distances as ONE batch: the launches of the square-root chain carry all of them
                vals, _ = hip.FrechetMultiJob([(base, a) for a in accs[:len(group)]], mean_dtype=mean_dtype).result_arrays()
                for k, fad in zip(group, vals):
                    values[k] = np.float64(fad)
            for a in accs:                                 # (the scores are in: the frames need not outlive the call)
                a.release_inputs(staging=False)
        return values

    def score_individual(self, baseline: PathLike, eval_dir: PathLike, csv_name: Union[Path, str]) -> Path:
        """Per-file FAD against the baseline, written as ``path,score`` lines sorted by |score|
        (fad.py:353-395).  All songs are scored in one batched GPU call; files whose embedding is
        missing, unreadable or shorter than two frames are logged and dropped like the reference."""
        csv = Path(csv_name)
        if isinstance(csv_name, str):
            csv = Path("data") / "fad-individual" / self.ml.name / csv_name
        if csv.exists():
            log.info(f"CSV file {csv} already exists, exiting...")
            return csv

        from . import dist
        mu, cov = self.load_stats(baseline, collective=True)          # every rank of the job comes through here
        all_files = sorted(Path(eval_dir).glob("*.*")) if dist.world_size() > 1 else list(Path(eval_dir).glob("*.*"))
        _files = dist.shard(all_files) if dist.world_size() > 1 else all_files      # songs are independent: shard them

        def _read(f):
            try:
                return self.read_embedding_file(f)
            except Exception as e:      # noqa: BLE001
                traceback.print_exc()
                log.error(f"An error occurred calculating individual FAD using model {self.ml.name} on file {f}")
                log.error(e)
                return None

        embds = tmap(_read, _files, desc="Loading embeddings", max_workers=self.audio_load_worker)
        scores = [None] * len(_files)
        ok = [i for i, e in enumerate(embds) if e is not None and e.ndim == 2 and e.shape[1] == np.shape(mu)[-1]]
        for i, e in enumerate(embds):
            if e is not None and i not in set(ok):
                log.error(f"Embedding of {_files[i]} has shape {e.shape}; expected [*, {np.shape(mu)[-1]}]")
        # one batched call per embedding dtype: the reference takes np.mean of every file in ITS dtype (fad.py:373, Q1), so files
        # cached as float16 and float32 in one directory must not be cast to a common type before their means are rounded
        by_dtype = {}
        for i in ok:
            by_dtype.setdefault(embds[i].dtype, []).append(i)
        for group in by_dtype.values():
            rows = np.concatenate([embds[i] for i in group], axis=0)
            offs = np.concatenate([[0], np.cumsum([embds[i].shape[0] for i in group])])
            vals, status = hip.frechet_batched(mu, cov, rows, offs, mean_mode=1, device=self.device_index)
            for j, i in enumerate(group):
                if status[j] == 0 or status[j] == -8:
                    scores[i] = np.float64(vals[j])
                else:
                    log.error(f"An error occurred calculating individual FAD using model {self.ml.name} on file "
                              f"{_files[i]} (status {status[j]}: "
                              f"{'fewer than two frames' if status[j] == -6 else 'non-finite result'})")

        pairs = [p for p in zip(_files, scores) if p[1] is not None]
        if dist.world_size() > 1:                      # rank-ordered gather keeps the file order; rank 0 writes
            pairs = [p for part in dist.gather_objects(pairs) for p in part]
        if dist.rank() == 0:
            pairs = sorted(pairs, key=lambda x: np.abs(x[1]))
            write(csv, "\n".join(",".join(str(x).replace(",", "_") for x in row) for row in pairs))
        dist.barrier()                                 # no rank returns before the CSV is there
        return csv
