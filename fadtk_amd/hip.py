"""Thin object layer over the C ABI: running moments, Frechet distance, batched per-song FAD.

Everything here forwards to libfad_hip.so; numpy is only used to own host buffers.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _capi as K


class Moments:
    """Running (n, sum x, sum x x^T) of frames in float64 HBM (``fad_moments_*``).

    Replaces ``np.mean``/``np.cov`` of fadtk/fad.py:48 and the per-file merge of
    fadtk/utils.py:13-45.  ``update`` accepts numpy arrays (host, staged over PCIe by the library)
    or torch CUDA tensors (used in place, on torch's current stream).
    """

    def __init__(self, d: int, device: int = 0):
        self._lib = K.load_library()
        K.require_gpu(device)
        self.d = int(d)
        self.device = int(device)
        h = C.c_void_p()
        K.check(self._lib.fad_moments_create(self.d, self.device, C.byref(h)), "fad_moments_create")
        self._h = h

    # -- lifetime
    def close(self):
        if getattr(self, "_h", None):
            self._lib.fad_moments_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _stream(self) -> int:
        return K.current_stream_ptr(self.device)

    # -- accumulation
    def bind(self, tensor) -> "Moments":
        """Keep the packed statistics in ``tensor`` (float64 CUDA, ``packed_len`` elements, contiguous) and reset.
        Collectives can then run over the tensor in place; the handle holds a reference to keep it alive."""
        assert tensor.is_cuda and tensor.dtype.is_floating_point and tensor.element_size() == 8
        assert tensor.numel() == self.packed_len and tensor.is_contiguous() and tensor.device.index == self.device
        K.check(self._lib.fad_moments_bind(self._h, C.c_void_p(tensor.data_ptr())), "fad_moments_bind")
        self._bound = tensor
        return self

    def reset(self):
        K.check(self._lib.fad_moments_reset(self._h, self._stream()), "fad_moments_reset")

    def release_inputs(self, staging: bool = True):
        """Forget the last device tensor fed (kept alive for the enqueued kernels -- call this only after something synchronised,
        e.g. finalize() or a collected score) and, with ``staging``, give back a host-input staging area above 256 MiB: a cached
        handle must not pin the caller's frames in HBM between calls, nor a large share of it -- but freeing and re-allocating the
        102 MB of a config-3 set on every call (hipFree synchronises, hipMalloc maps pages) cost ~0.5 ms of a 2.7 ms
        calc_embd_statistics (round 5); 256 MiB is 0.1 % of this GPU's memory."""
        self._keep = None
        if staging:
            K.check(self._lib.fad_moments_trim(self._h, 256 << 20), "fad_moments_trim")

    def settle(self):
        """Make a pending reset visible in the packed buffer (the zeroing is deferred until something reads it)."""
        K.check(self._lib.fad_moments_settle(self._h, self._stream()), "fad_moments_settle")

    def update(self, rows) -> "Moments":
        ptr, n, d, ld, code, on_dev, keep = K.rows_view(rows)
        if d != self.d:
            raise AssertionError(f"frame matrix has {d} features, accumulator has {self.d}")
        K.check(self._lib.fad_moments_update(self._h, ptr, n, ld, code, on_dev, self._stream()), "fad_moments_update")
        if on_dev:
            self._keep = keep          # torch tensor must outlive the enqueued kernels
        return self

    def update_segmented(self, rows, offsets: Sequence[int], want_sums: bool = True, sums_on_device: bool = False, want_runsums: bool = False):
        """Feed files/songs stored back to back; returns per-segment column sums [S x D] float64 -- a numpy array, or
        (``sums_on_device`` with device rows) a torch CUDA tensor that never leaves HBM.  ``want_runsums``: returns ``(sums, runsums)``
        with numpy's float32 running column sums per segment [S x D] float32 (``fad_moments_update_segmented_ref``; None for float64
        rows, whose numpy sum is the exact one)."""
        ptr, n, d, ld, code, on_dev, keep = K.rows_view(rows)
        if d != self.d:
            raise AssertionError(f"frame matrix has {d} features, accumulator has {self.d}")
        off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
        n_seg = off.shape[0] - 1
        sums = None
        sums_ptr = None
        sums_dev = None
        runs, runs_ptr, runs_dev = None, None, None
        want_runsums = want_runsums and code != K.FAD_F64 and n_seg > 0
        if want_sums and n_seg > 0:
            if on_dev:
                import torch
                sums_dev = torch.empty((n_seg, self.d), dtype=torch.float64, device=keep.device)   # every entry is written (segment_gather_sums)
                sums_ptr = sums_dev.data_ptr()
            else:
                sums = np.zeros((n_seg, self.d), dtype=np.float64)
                sums_ptr = sums.ctypes.data
        if want_runsums:
            if on_dev:
                import torch
                runs_dev = torch.empty((n_seg, self.d), dtype=torch.float32, device=keep.device)
                runs_ptr = runs_dev.data_ptr()
            else:
                runs = np.zeros((n_seg, self.d), dtype=np.float32)
                runs_ptr = runs.ctypes.data
            K.check(self._lib.fad_moments_update_segmented_ref(
                self._h, ptr, n, ld, code, off.ctypes.data_as(C.POINTER(C.c_int64)), n_seg, sums_ptr, runs_ptr, on_dev,
                self._stream()), "fad_moments_update_segmented_ref")
        else:
            K.check(self._lib.fad_moments_update_segmented(
                self._h, ptr, n, ld, code, off.ctypes.data_as(C.POINTER(C.c_int64)), n_seg, sums_ptr, on_dev,
                self._stream()), "fad_moments_update_segmented")
        if on_dev:
            self._keep = keep
        if sums_dev is not None:
            sums = sums_dev if sums_on_device else sums_dev.cpu().numpy()
        if runs_dev is not None:
            runs = runs_dev if sums_on_device else runs_dev.cpu().numpy()
        return (sums, runs) if want_runsums else sums

    @staticmethod
    def update_multi(accs: Sequence["Moments"], blocks: Sequence) -> None:
        """``accs[i].update(blocks[i])`` for up to 32 accumulators of one dimension with ONE launch of each kernel
        (``fad_moments_update_multi``); every block must be a device tensor of one common dtype."""
        assert 1 <= len(accs) == len(blocks) <= 32
        views = [K.rows_view(b) for b in blocks]
        code = views[0][4]
        for a, v in zip(accs, views):
            if v[1] > 0 and v[2] != a.d:
                raise AssertionError(f"frame matrix has {v[2]} features, accumulator has {a.d}")
            if not v[5] or v[4] != code:
                raise AssertionError("update_multi needs device tensors of one common dtype")
        m = len(accs)
        hs = (C.c_void_p * m)(*[a._h for a in accs])
        ptrs = (C.c_void_p * m)(*[v[0] for v in views])
        ns = (C.c_int64 * m)(*[v[1] for v in views])
        lds = (C.c_int64 * m)(*[max(v[3], a.d) for a, v in zip(accs, views)])
        K.check(accs[0]._lib.fad_moments_update_multi(m, hs, ptrs, ns, lds, code, accs[0]._stream()), "fad_moments_update_multi")
        for a, v in zip(accs, views):
            a._keep = v[6]

    @staticmethod
    def update_multi_indexed(accs: Sequence["Moments"], rows, indices: Sequence) -> None:
        """``accs[i].update(rows[indices[i]])`` for up to 16 accumulators WITHOUT materialising the gathered matrices
        (``fad_moments_update_multi_indexed``): ``rows`` one resident device tensor [n_src x D], ``indices[i]`` int32 device tensors
        (1-D, values in [0, n_src)).  The resamples with replacement of FAD-inf (fad.py:333-337)."""
        import torch
        assert 1 <= len(accs) == len(indices) <= 16
        v = K.rows_view(rows)
        if not v[5]:
            raise AssertionError("update_multi_indexed needs a device tensor")
        m = len(accs)
        for a in accs:
            if v[2] != a.d:
                raise AssertionError(f"frame matrix has {v[2]} features, accumulator has {a.d}")
        idx = [i if (i.dtype == torch.int32 and i.is_contiguous()) else i.to(torch.int32).contiguous() for i in indices]
        for i in idx:
            assert i.is_cuda and i.dim() == 1
        hs = (C.c_void_p * m)(*[a._h for a in accs])
        ips = (C.c_void_p * m)(*[i.data_ptr() for i in idx])
        ns = (C.c_int64 * m)(*[int(i.numel()) for i in idx])
        K.check(accs[0]._lib.fad_moments_update_multi_indexed(m, hs, v[0], int(v[1]), int(max(v[3], accs[0].d)), v[4], ips, ns, accs[0]._stream()),
                "fad_moments_update_multi_indexed")
        for a, i in zip(accs, idx):
            a._keep = (v[6], i)

    @staticmethod
    def prepared_update_multi(accs: Sequence["Moments"], blocks: Sequence) -> "PreparedMultiUpdate":
        """``update_multi`` for a loop that feeds the SAME accumulators from the SAME device tensors again and again: see PreparedMultiUpdate."""
        return PreparedMultiUpdate(accs, blocks)

    @staticmethod
    def update_file_means(exact: "Moments", rounded: "Moments", weighted: "Moments", seg_sums, sizes, dtype_code: int, seg_runsums=None) -> None:
        """Accumulate the per-file mean rows of the online statistics (``fad_moments_update_file_means[_ref]``).
        ``seg_sums`` [F x D] float64 and ``sizes`` [F] int64: both numpy, or both torch CUDA tensors (sizes may stay on the host);
        ``seg_runsums`` [F x D] float32 (where ``seg_sums`` lives) = numpy's per-file running sums: the rounded means are then the
        reference's own (utils.py:16), else the rounded exact means."""
        lib = exact._lib
        if K._is_torch(seg_sums) and seg_sums.is_cuda:
            import torch
            sums_t = seg_sums.to(torch.float64).contiguous()
            n_files = int(sums_t.shape[0])
            if K._is_torch(sizes) and sizes.is_cuda:
                sizes_t = sizes.to(torch.int64).contiguous()
                flag, sizes_ptr, keep = 3, sizes_t.data_ptr(), sizes_t
            else:                                       # the usual case: sums in HBM, sizes known on the host
                sz = np.ascontiguousarray(np.asarray(sizes.cpu() if K._is_torch(sizes) else sizes, dtype=np.int64))
                flag, sizes_ptr, keep = 1, sz.ctypes.data, sz
            if seg_runsums is not None:
                runs_t = seg_runsums.to(torch.float32).contiguous()
                assert runs_t.is_cuda and tuple(runs_t.shape) == tuple(sums_t.shape)
                K.check(lib.fad_moments_update_file_means_ref(exact._h, rounded._h, weighted._h, sums_t.data_ptr(), runs_t.data_ptr(),
                                                              sizes_ptr, n_files, int(dtype_code), flag | 4, exact._stream()),
                        "fad_moments_update_file_means_ref")
                exact._keep = (sums_t, runs_t, keep)
            else:
                K.check(lib.fad_moments_update_file_means(exact._h, rounded._h, weighted._h, sums_t.data_ptr(),
                                                          sizes_ptr, n_files, int(dtype_code), flag, exact._stream()),
                        "fad_moments_update_file_means")
                exact._keep = (sums_t, keep)
        else:
            sums = K.f64_host(seg_sums)
            sz = np.ascontiguousarray(np.asarray(sizes, dtype=np.int64))
            if seg_runsums is not None:
                runs = np.ascontiguousarray(np.asarray(seg_runsums, dtype=np.float32))
                assert runs.shape == sums.shape
                K.check(lib.fad_moments_update_file_means_ref(exact._h, rounded._h, weighted._h, sums.ctypes.data, runs.ctypes.data,
                                                              sz.ctypes.data, int(sums.shape[0]), int(dtype_code), 0, exact._stream()),
                        "fad_moments_update_file_means_ref")
            else:
                K.check(lib.fad_moments_update_file_means(exact._h, rounded._h, weighted._h, sums.ctypes.data,
                                                          sz.ctypes.data, int(sums.shape[0]), int(dtype_code), 0, exact._stream()),
                        "fad_moments_update_file_means")

    def merge(self, other: "Moments") -> "Moments":
        K.check(self._lib.fad_moments_merge(self._h, other._h, self._stream()), "fad_moments_merge")
        return self

    def allreduce_rccl(self, comm_ptr: int) -> "Moments":
        """Sum the statistics of all ranks in place through the caller's ``ncclComm_t`` (its address as an int).
        Python hosts normally go through ``fadtk_amd.dist.allreduce_moments`` (torch.distributed) instead."""
        K.check(self._lib.fad_moments_allreduce(self._h, C.c_void_p(comm_ptr), self._stream()), "fad_moments_allreduce")
        return self

    # -- packed statistics (what an RCCL all-reduce runs over)
    @property
    def packed_len(self) -> int:
        return 1 + self.d + self.d * self.d

    def export(self) -> np.ndarray:
        out = np.empty(self.packed_len, dtype=np.float64)
        K.check(self._lib.fad_moments_export(self._h, out.ctypes.data, 0, self._stream()), "fad_moments_export")
        return out

    def export_to(self, tensor):
        """Copy the packed statistics into a float64 torch CUDA tensor of ``packed_len`` elements."""
        assert tensor.is_cuda and tensor.numel() == self.packed_len and tensor.is_contiguous()
        K.check(self._lib.fad_moments_export(self._h, tensor.data_ptr(), 1, self._stream()), "fad_moments_export")
        return tensor

    def import_(self, packed) -> "Moments":
        if K._is_torch(packed) and packed.is_cuda:
            assert packed.numel() == self.packed_len and packed.is_contiguous()
            K.check(self._lib.fad_moments_import(self._h, packed.data_ptr(), 1, self._stream()), "fad_moments_import")
        else:
            a = K.f64_host(packed, (self.packed_len,))
            K.check(self._lib.fad_moments_import(self._h, a.ctypes.data, 0, self._stream()), "fad_moments_import")
        return self

    @property
    def count(self) -> int:
        n = C.c_int64()
        K.check(self._lib.fad_moments_count(self._h, C.byref(n), self._stream()), "fad_moments_count")
        return int(n.value)

    def finalize(self, ddof: int = 1) -> Tuple[np.ndarray, np.ndarray, int]:
        """-> (mu [D] float64, cov [D x D] float64, n).  n < 2 raises AssertionError (fad.py:46-47)."""
        mu = np.empty(self.d, dtype=np.float64)
        cov = np.empty((self.d, self.d), dtype=np.float64)
        n = C.c_int64()
        K.check(self._lib.fad_moments_finalize(self._h, int(ddof), mu.ctypes.data, cov.ctypes.data, C.byref(n), 0,
                                               self._stream()), "fad_moments_finalize")
        return mu, cov, int(n.value)

    # -- timing of the dominant kernel (bench.py)
    def set_reference_mean(self, on: bool = True, detached: bool = False) -> None:
        """Carry numpy's float32 running column sums beside the exact ones (``fad_moments_set_reference_mean``): ``finalize`` then
        returns the mean ``np.mean(frames, axis=0)`` has (fadtk/fad.py:48) before its final cast -- bit for bit after the caller's
        ``astype`` -- instead of the exact mean.  The walk runs on a stream of its own beside the update's other kernels;
        ``detached=True``: the caller vouches that the frames it feeds are complete at the call and stay unchanged until the statistics
        are next read -- the walk then waits for nothing and holds nothing up (see include/fad_hip.h)."""
        K.check(self._lib.fad_moments_set_reference_mean(self._h, (2 if detached else 1) if on else 0))

    def set_timing(self, on=True):
        """True / 1: events around the tile kernel and behind the reduce; 2: around the tile kernel only; False / 0: off."""
        K.check(self._lib.fad_moments_set_timing(self._h, 2 if on == 2 else (1 if on else 0)))

    def last_timing(self):
        a, b, v = C.c_float(), C.c_float(), C.c_int()
        K.check(self._lib.fad_moments_last_timing(self._h, C.byref(a), C.byref(b), C.byref(v)))
        return float(a.value), float(b.value), int(v.value)


def cu_masked_stream(cus_per_xcd, device: int = 0, xcds: int = 8, cus_in_xcd: int = 32):
    """A torch stream confined to CUs `cus_per_xcd` = range(lo, hi) of every XCD (diagnostics: bench.py --chain-cus).  Mask bit
    i = CU i of the runtime's numbering, XCD-interleaved: CU c of XCD x is bit c * xcds + x."""
    import torch
    lib = K.load_library()
    K.require_gpu(device)
    words = (xcds * cus_in_xcd + 31) // 32
    mask = (C.c_uint32 * words)()
    for c in cus_per_xcd:
        for x in range(xcds):
            bit = c * xcds + x
            mask[bit // 32] |= (1 << (bit % 32))
    ptr = C.c_void_p()
    K.check(lib.fad_stream_create_cu_mask(int(device), mask, words, C.byref(ptr)), "fad_stream_create_cu_mask")
    return torch.cuda.ExternalStream(ptr.value, device=torch.device("cuda", device))


def frechet(mu1, cov1, mu2, cov2, eps: float = 1e-6, max_iter: int = 0, tol: float = 0.0, device: int = 0):
    """``fad_frechet`` on host float64 arrays -> (fad, diag dict).  Shapes are checked by the caller."""
    lib = K.load_library()
    K.require_gpu(device)
    mu1 = K.f64_host(mu1)
    mu2 = K.f64_host(mu2)
    d = mu1.shape[0]
    cov1 = K.f64_host(cov1, (d, d))
    cov2 = K.f64_host(cov2, (d, d))
    out = C.c_double()
    diag = K.FadDiag()
    st = lib.fad_frechet(d, mu1.ctypes.data, cov1.ctypes.data, mu2.ctypes.data, cov2.ctypes.data, float(eps),
                         int(max_iter), float(tol), 0, int(device), K.current_stream_ptr(device), C.byref(out),
                         C.byref(diag))
    K.check(st, "fad_frechet")
    return float(out.value), diag.as_dict()


def frechet_from_moments(m1: Moments, m2: Moments, ddof: int = 1, eps: float = 1e-6, max_iter: int = 0,
                         tol: float = 0.0, mean_dtype: int = -1):
    """FAD straight from two accumulators, all in HBM (``fad_frechet_from_moments``).  ``mean_dtype`` = ``K.FAD_F16``
    gives the reference's value for float16 embeddings (means and mean term rounded the way numpy does)."""
    lib = K.load_library()
    out = C.c_double()
    diag = K.FadDiag()
    st = lib.fad_frechet_from_moments(m1._h, m2._h, int(ddof), float(eps), int(max_iter), float(tol), int(mean_dtype),
                                      K.current_stream_ptr(m1.device), C.byref(out), C.byref(diag))
    K.check(st, "fad_frechet_from_moments")
    return float(out.value), diag.as_dict()


class FrechetJob:
    """A score in flight (``fad_frechet_from_moments_begin``): the whole square-root chain is enqueued on the stream that
    was current when it was created; ``result()`` waits for it -> (fad, diag dict).

    The slot behind a job belongs to the host thread that created it (the library keeps its workspaces per thread and frees
    them when the thread ends), so ``result()`` / ``cancel()`` from another thread raise, and a job that is garbage-collected
    on another thread -- or after its thread has gone -- is left alone instead of touching a workspace that may no longer
    exist."""

    def __init__(self, m1: Moments, m2: Moments, ddof: int = 1, eps: float = 1e-6, mean_dtype: int = -1):
        import threading
        self._lib = K.load_library()
        self._owner = threading.get_ident()
        job = C.c_void_p()
        K.check(self._lib.fad_frechet_from_moments_begin(m1._h, m2._h, int(ddof), float(eps), int(mean_dtype),
                                                         K.current_stream_ptr(m1.device), C.byref(job)),
                "fad_frechet_from_moments_begin")
        self._job = job

    def _on_owner_thread(self) -> bool:
        import threading
        return threading.get_ident() == self._owner

    def result(self):
        if self._job is None:
            raise RuntimeError("this job was collected already")
        if not self._on_owner_thread():
            raise RuntimeError("a FrechetJob must be collected by the thread that created it (its slot is thread-local)")
        out, diag = C.c_double(), K.FadDiag()
        job, self._job = self._job, None
        K.check(self._lib.fad_frechet_end(job, C.byref(out), C.byref(diag)), "fad_frechet_end")
        return float(out.value), diag.as_dict()

    def cancel(self):
        """Give the slot back without collecting the score (waits for the enqueued kernels)."""
        if self._job is not None:
            if not self._on_owner_thread():
                raise RuntimeError("a FrechetJob must be cancelled by the thread that created it (its slot is thread-local)")
            job, self._job = self._job, None
            K.check(self._lib.fad_frechet_cancel(job), "fad_frechet_cancel")

    def __del__(self):            # a job dropped without result(): its slot must not stay taken (8 per thread)
        try:
            if getattr(self, "_job", None) is not None:
                if self._on_owner_thread():
                    self.cancel()
                else:             # the slot is thread-local: it stays taken until its thread ends -- say so instead of failing later
                    import warnings
                    warnings.warn("a FrechetJob was dropped on a thread other than the one that created it: its slot (8 per thread "
                                  "and device) stays busy until that thread ends; call result() or cancel() on the owner thread",
                                  ResourceWarning, stacklevel=2)
        except Exception:         # noqa: BLE001  interpreter shutdown, library gone
            pass


class PreparedMultiUpdate:
    """``Moments.update_multi(accs, blocks)`` with the views and ctypes tables built ONCE: a loop that scores resident sets over and over
    (bench.py's; a service that re-scores a fixed evaluation set against changing baselines) spends 5-6 us of Python per frame matrix in
    ``update_multi`` -- 0.2 ms for the 32 matrices of a batch, during which the GPU waits for its first launch.  ``run()`` is one call into
    the library (two with ``reset=True``: ``fad_moments_reset_multi`` first).  The tensors are kept alive by the object."""

    def __init__(self, accs, blocks):
        assert 1 <= len(accs) == len(blocks) <= 32
        views = [K.rows_view(b) for b in blocks]
        code = views[0][4]
        for a, v in zip(accs, views):
            if v[1] > 0 and v[2] != a.d:
                raise AssertionError(f"frame matrix has {v[2]} features, accumulator has {a.d}")
            if not v[5] or v[4] != code:
                raise AssertionError("update_multi needs device tensors of one common dtype")
        m = len(accs)
        self._accs, self._keep, self._m, self._code = list(accs), [v[6] for v in views], m, code
        self._hs = (C.c_void_p * m)(*[a._h for a in accs])
        self._ptrs = (C.c_void_p * m)(*[v[0] for v in views])
        self._ns = (C.c_int64 * m)(*[v[1] for v in views])
        self._lds = (C.c_int64 * m)(*[max(v[3], a.d) for a, v in zip(accs, views)])
        self._lib = accs[0]._lib

    def run(self, reset: bool = False) -> None:
        st = self._accs[0]._stream()
        if reset:
            K.check(self._lib.fad_moments_reset_multi(self._m, self._hs, st), "fad_moments_reset_multi")
        K.check(self._lib.fad_moments_update_multi(self._m, self._hs, self._ptrs, self._ns, self._lds, self._code, st), "fad_moments_update_multi")
        for a, k in zip(self._accs, self._keep):
            a._keep = k


class FrechetMultiJob:
    """Up to MAX_PAIRS scores in flight as ONE batch (``fad_frechet_from_moments_multi_begin``): pair b = (pairs[b][0], pairs[b][1]);
    the eight launches of the square-root chain carry all of them.  ``result()`` -> [(fad, diag dict), ...] in order.  Thread
    rules as FrechetJob."""

    MAX_PAIRS = 32                 # FAD_MULTI_MAX_PAIRS (include/fad_hip.h)

    def __init__(self, pairs, ddof: int = 1, eps: float = 1e-6, mean_dtype: int = -1):
        import threading
        self._lib = K.load_library()
        self._owner = threading.get_ident()
        self._n = len(pairs)
        if not 1 <= self._n <= self.MAX_PAIRS:
            raise ValueError(f"a FrechetMultiJob holds 1..{self.MAX_PAIRS} pairs")
        a = (C.c_void_p * self._n)(*[p[0]._h for p in pairs])
        b = (C.c_void_p * self._n)(*[p[1]._h for p in pairs])
        job = C.c_void_p()
        K.check(self._lib.fad_frechet_from_moments_multi_begin(self._n, a, b, int(ddof), float(eps), int(mean_dtype),
                                                               K.current_stream_ptr(pairs[0][0].device), C.byref(job)),
                "fad_frechet_from_moments_multi_begin")
        self._job = job

    def result(self):
        import threading
        if self._job is None:
            raise RuntimeError("this job was collected already")
        if threading.get_ident() != self._owner:
            raise RuntimeError("a FrechetMultiJob must be collected by the thread that created it (its slot is thread-local)")
        out = (C.c_double * self._n)()
        diag = (K.FadDiag * self._n)()
        job, self._job = self._job, None
        K.check(self._lib.fad_frechet_multi_end(job, self._n, out, diag), "fad_frechet_multi_end")
        return [(float(out[i]), diag[i].as_dict()) for i in range(self._n)]

    def result_arrays(self):
        """``result()`` without the per-pair Python objects: (float64 array of the distances, the ctypes array of fad_diag_t records --
        ``diags[i].as_dict()`` on demand).  Thirty-two dicts of diagnostics cost more host time than the library's own collection."""
        import threading
        if self._job is None:
            raise RuntimeError("this job was collected already")
        if threading.get_ident() != self._owner:
            raise RuntimeError("a FrechetMultiJob must be collected by the thread that created it (its slot is thread-local)")
        out = np.empty(self._n, dtype=np.float64)
        diag = (K.FadDiag * self._n)()
        job, self._job = self._job, None
        K.check(self._lib.fad_frechet_multi_end(job, self._n, out.ctypes.data_as(C.POINTER(C.c_double)), diag), "fad_frechet_multi_end")
        return out, diag

    def cancel(self):
        import threading
        if self._job is not None and threading.get_ident() == self._owner:
            job, self._job = self._job, None
            K.check(self._lib.fad_frechet_cancel(job), "fad_frechet_cancel")

    def __del__(self):
        try:
            self.cancel()
        except Exception:         # noqa: BLE001
            pass


def frechet_batched(mu_b, cov_b, rows, offsets: Sequence[int], mean_mode: int = 1, device: int = 0):
    """Per-song FAD against one baseline (``fad_frechet_batched_vs_baseline``).

    -> (scores float64 [S], status int32 [S]); songs with < 2 frames get NaN and FAD_ERR_TOO_FEW_ROWS.
    """
    lib = K.load_library()
    K.require_gpu(device)
    base_on_dev = K._is_torch(mu_b) and K._is_torch(cov_b) and mu_b.is_cuda and cov_b.is_cuda
    if base_on_dev:                 # a baseline that already lives in HBM (float64) is used in place: no 8 D^2-byte upload per call
        import torch
        mu_t = mu_b.to(torch.float64).contiguous(); cov_t = cov_b.to(torch.float64).contiguous()
        d = int(mu_t.shape[0])
        assert tuple(cov_t.shape) == (d, d)
    else:
        mu_b = K.f64_host(mu_b)
        d = mu_b.shape[0]
        cov_b = K.f64_host(cov_b, (d, d))
    ptr, n, dd, ld, code, on_dev, keep = K.rows_view(rows)
    if n > 0 and dd != d:
        raise AssertionError(f"songs have {dd} features, baseline has {d}")
    if base_on_dev and not on_dev:
        mu_b, cov_b, base_on_dev = mu_t.cpu().numpy(), cov_t.cpu().numpy(), False      # host rows: the library stages everything itself
    off = np.ascontiguousarray(np.asarray(offsets, dtype=np.int64))
    n_songs = off.shape[0] - 1
    scores = np.full(max(n_songs, 0), np.nan, dtype=np.float64)
    status = np.zeros(max(n_songs, 0), dtype=np.int32)
    if on_dev:
        # baseline must live where the rows live
        import torch
        if not base_on_dev:
            mu_t = torch.from_numpy(mu_b).to(keep.device)
            cov_t = torch.from_numpy(cov_b).to(keep.device)
        elif mu_t.device != keep.device:
            mu_t, cov_t = mu_t.to(keep.device), cov_t.to(keep.device)
        mu_p, cov_p = mu_t.data_ptr(), cov_t.data_ptr()
    else:
        mu_p, cov_p = mu_b.ctypes.data, cov_b.ctypes.data
    st = lib.fad_frechet_batched_vs_baseline(d, mu_p, cov_p, ptr, n, max(ld, d), code,
                                             off.ctypes.data_as(C.POINTER(C.c_int64)), n_songs, int(mean_mode),
                                             on_dev, int(device), K.current_stream_ptr(device),
                                             scores.ctypes.data, status.ctypes.data)
    K.check(st, "fad_frechet_batched_vs_baseline")
    return scores, status


# ---------------------------------------------------------------------------------------------
# log-mel front ends (csrc/logmel.hip)
# ---------------------------------------------------------------------------------------------
def _clips_view(clips, device: int):
    """list of 1-D arrays (numpy / torch) or one 1-D array -> (ptr, offsets int64[n+1], on_device, keepalive)."""
    if not isinstance(clips, (list, tuple)):
        clips = [clips]
    if len(clips) and K._is_torch(clips[0]) and clips[0].is_cuda:
        import torch
        flat = torch.cat([c.reshape(-1).to(torch.float32) for c in clips]) if len(clips) > 1 else \
            clips[0].reshape(-1).to(torch.float32).contiguous()
        lens = [int(c.numel()) for c in clips]
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        return flat.data_ptr(), off, 1, flat
    arrs = [np.asarray(c.cpu().numpy() if K._is_torch(c) else c, dtype=np.float32).reshape(-1) for c in clips]
    flat = np.ascontiguousarray(np.concatenate(arrs)) if arrs else np.zeros(0, np.float32)
    off = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.int64)
    return flat.ctypes.data, off, 0, flat


def _out_buffer(shape, on_dev: int, like):
    if on_dev:
        import torch
        t = torch.empty(shape, dtype=torch.float32, device=like.device)
        return t, t.data_ptr()
    a = np.empty(shape, dtype=np.float32)
    return a, a.ctypes.data


def vggish_num_examples(n_samples: int) -> int:
    return int(K.load_library().fad_logmel_vggish_num_examples(int(n_samples)))


def logmel_vggish(clips, device: int = 0):
    """16 kHz mono clips -> (examples [E, 96, 64] float32, example_offsets int64 [n_clips + 1])."""
    lib = K.load_library()
    K.require_gpu(device)
    ptr, off, on_dev, keep = _clips_view(clips, device)
    n = len(off) - 1
    total = sum(vggish_num_examples(int(off[i + 1] - off[i])) for i in range(n))
    out, optr = _out_buffer((total, 96, 64), on_dev, keep)
    ex_off = np.zeros(n + 1, dtype=np.int64)
    K.check(lib.fad_logmel_vggish(ptr, off.ctypes.data_as(C.POINTER(C.c_int64)), n, optr, total,
                                  ex_off.ctypes.data_as(C.POINTER(C.c_int64)), on_dev, device,
                                  K.current_stream_ptr(device)), "fad_logmel_vggish")
    return out, ex_off


def logmel_whisper(clips, n_mels: int = 80, device: int = 0):
    """16 kHz mono clips -> [n_clips, n_mels, 3000] float32 (each clip padded / cut to 30 s)."""
    lib = K.load_library()
    K.require_gpu(device)
    ptr, off, on_dev, keep = _clips_view(clips, device)
    n = len(off) - 1
    out, optr = _out_buffer((n, n_mels, 3000), on_dev, keep)
    K.check(lib.fad_logmel_whisper(ptr, off.ctypes.data_as(C.POINTER(C.c_int64)), n, int(n_mels), optr, on_dev, device,
                                   K.current_stream_ptr(device)), "fad_logmel_whisper")
    return out


def logmel_htsat(clips, device: int = 0):
    """48 kHz mono clips of ONE common length -> [n_clips, 1 + n/480, 64] float32 (dB log-mel)."""
    lib = K.load_library()
    K.require_gpu(device)
    ptr, off, on_dev, keep = _clips_view(clips, device)
    n = len(off) - 1
    frames = 1 + int(off[1] - off[0]) // 480 if n else 1
    out, optr = _out_buffer((n, frames, 64), on_dev, keep)
    K.check(lib.fad_logmel_htsat(ptr, off.ctypes.data_as(C.POINTER(C.c_int64)), n, frames, optr, on_dev, device,
                                 K.current_stream_ptr(device)), "fad_logmel_htsat")
    return out


def resample_kaiser(wav, orig_sr: int, new_sr: int, quantize_pcm16: bool = False, device: int = 0):
    """Mono float32 audio (numpy, or a torch CUDA tensor that stays on the device) resampled from ``orig_sr`` to
    ``new_sr`` with fadtk's Kaiser-windowed sinc filter (fad.py:151-159); ``quantize_pcm16`` adds the reference's
    16-bit cache-file round trip."""
    lib = K.load_library()
    K.require_gpu(device)
    ptr, off, on_dev, keep = _clips_view([wav], device)
    n = int(off[1])
    n_out = int(lib.fad_resample_num_samples(n, int(orig_sr), int(new_sr)))
    if n_out < 0:
        K.check(n_out, "fad_resample_num_samples")
    out, optr = _out_buffer((n_out,), on_dev, keep)
    K.check(lib.fad_resample_kaiser(ptr, n, int(orig_sr), int(new_sr), int(bool(quantize_pcm16)), optr, n_out, on_dev, device,
                                    K.current_stream_ptr(device)), "fad_resample_kaiser")
    return out
