"""ctypes binding of libfad_hip.so (include/fad_hip.h) -- the ONLY compute path of this package.

There is deliberately no CPU fallback: if the library or a gfx950 GPU is missing, calls raise
``FadHipUnavailable``.  (The numpy oracle under ``oracle/`` is test infrastructure and is never
imported from here.)
"""
from __future__ import annotations

import ctypes as C
import logging
import os
import threading
from pathlib import Path
from typing import Optional, Tuple

import numpy as np

log = logging.getLogger("fadtk_amd")

FAD_OK = 0
FAD_ERR_INVALID = -1
FAD_ERR_NO_DEVICE = -2
FAD_ERR_HIP = -3
FAD_ERR_ALLOC = -4
FAD_ERR_SHAPE = -5
FAD_ERR_TOO_FEW_ROWS = -6
FAD_ERR_NOT_FINITE = -7
FAD_ERR_NOT_CONVERGED = -8

FAD_F16, FAD_BF16, FAD_F32, FAD_F64 = 0, 1, 2, 3
FAD_MEAN_SECOND_ONLY = 16     # | dtype: fad_frechet_from_moments' mean term with only the second mean rounded (include/fad_hip.h)

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libfad_hip.so"


class FadHipUnavailable(RuntimeError):
    """The HIP library / GPU needed by the FAD hot path is not usable (no CPU fallback exists)."""


class FadDiag(C.Structure):
    _fields_ = [("iters", C.c_int32), ("converged", C.c_int32), ("used_eps", C.c_int32), ("route", C.c_int32),
                ("residual", C.c_double), ("scale", C.c_double), ("mean_term", C.c_double),
                ("tr1", C.c_double), ("tr2", C.c_double), ("tr_sqrt", C.c_double), ("verified", C.c_int32), ("reserved", C.c_int32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


_P = C.c_void_p
_I64 = C.c_int64
# name -> (restype, argtypes)     -- one entry per declaration in include/fad_hip.h
SIGNATURES = {
    "fad_version": (C.c_int, []),
    "fad_device_count": (C.c_int, []),
    "fad_last_error": (C.c_char_p, []),
    "fad_device_arch": (C.c_char_p, [C.c_int]),
    "fad_moments_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(_P)]),
    "fad_moments_destroy": (C.c_int, [_P]),
    "fad_moments_reset": (C.c_int, [_P, _P]),
    "fad_moments_reset_multi": (C.c_int, [C.c_int, C.POINTER(_P), _P]),
    "fad_moments_settle": (C.c_int, [_P, _P]),
    "fad_moments_dim": (C.c_int, [_P]),
    "fad_moments_packed_len": (_I64, [_P]),
    "fad_moments_update": (C.c_int, [_P, _P, _I64, _I64, C.c_int, C.c_int, _P]),
    "fad_moments_update_multi": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_I64), C.POINTER(_I64), C.c_int, _P]),
    "fad_moments_update_multi_indexed": (C.c_int, [C.c_int, C.POINTER(_P), _P, _I64, _I64, C.c_int, C.POINTER(_P), C.POINTER(_I64), _P]),
    "fad_moments_update_file_means": (C.c_int, [_P, _P, _P, _P, _P, _I64, C.c_int, C.c_int, _P]),
    "fad_moments_update_segmented": (C.c_int, [_P, _P, _I64, _I64, C.c_int, C.POINTER(_I64), _I64, _P, C.c_int, _P]),
    "fad_moments_update_segmented_ref": (C.c_int, [_P, _P, _I64, _I64, C.c_int, C.POINTER(_I64), _I64, _P, _P, C.c_int, _P]),
    "fad_moments_update_file_means_ref": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, C.c_int, C.c_int, _P]),
    "fad_moments_merge": (C.c_int, [_P, _P, _P]),
    "fad_moments_export": (C.c_int, [_P, _P, C.c_int, _P]),
    "fad_moments_import": (C.c_int, [_P, _P, C.c_int, _P]),
    "fad_moments_count": (C.c_int, [_P, C.POINTER(_I64), _P]),
    "fad_moments_allreduce": (C.c_int, [_P, _P, _P]),
    "fad_moments_bind": (C.c_int, [_P, _P]),
    "fad_moments_finalize": (C.c_int, [_P, C.c_int, _P, _P, C.POINTER(_I64), C.c_int, _P]),
    "fad_moments_trim": (C.c_int, [_P, _I64]),
    "fad_moments_set_timing": (C.c_int, [_P, C.c_int]),
    "fad_moments_set_reference_mean": (C.c_int, [_P, C.c_int]),
    "fad_moments_last_timing": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "fad_stream_create_cu_mask": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(_P)]),
    "fad_stream_destroy": (C.c_int, [C.c_int, _P]),
    "fad_frechet": (C.c_int, [C.c_int, _P, _P, _P, _P, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, _P,
                              C.POINTER(C.c_double), C.POINTER(FadDiag)]),
    "fad_frechet_from_moments": (C.c_int, [_P, _P, C.c_int, C.c_double, C.c_int, C.c_double, C.c_int, _P,
                                           C.POINTER(C.c_double), C.POINTER(FadDiag)]),
    "fad_frechet_from_moments_begin": (C.c_int, [_P, _P, C.c_int, C.c_double, C.c_int, _P, C.POINTER(_P)]),
    "fad_frechet_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(FadDiag)]),
    "fad_frechet_cancel": (C.c_int, [_P]),
    "fad_frechet_from_moments_multi_begin": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_P), C.c_int, C.c_double, C.c_int, _P, C.POINTER(_P)]),
    "fad_frechet_multi_end": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(FadDiag)]),
    "fad_frechet_batched_vs_baseline": (C.c_int, [C.c_int, _P, _P, _P, _I64, _I64, C.c_int, C.POINTER(_I64), _I64,
                                                  C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "fad_resample_num_samples": (_I64, [_I64, C.c_int, C.c_int]),
    "fad_resample_kaiser": (C.c_int, [_P, _I64, C.c_int, C.c_int, C.c_int, _P, _I64, C.c_int, C.c_int, _P]),
    "fad_logmel_vggish_num_examples": (_I64, [_I64]),
    "fad_logmel_vggish": (C.c_int, [_P, C.POINTER(_I64), _I64, _P, _I64, C.POINTER(_I64), C.c_int, C.c_int, _P]),
    "fad_logmel_whisper": (C.c_int, [_P, C.POINTER(_I64), _I64, C.c_int, _P, C.c_int, C.c_int, _P]),
    "fad_logmel_htsat": (C.c_int, [_P, C.POINTER(_I64), _I64, _I64, _P, C.c_int, C.c_int, _P]),
}

_lib = None
_lib_lock = threading.Lock()


def load_library(path: Optional[os.PathLike] = None):
    """dlopen libfad_hip.so and attach prototypes.  Works without a GPU (symbol checks only)."""
    global _lib
    with _lib_lock:
        if _lib is not None and path is None:
            return _lib
        p = Path(path) if path else LIB_PATH
        if not p.exists():
            raise FadHipUnavailable(
                f"{p} is missing: build it with `python -m fadtk_amd.build` (needs hipcc). "
                "fadtk_amd has no CPU fallback for the FAD hot path.")
        try:                       # bring PyTorch's bundled HIP runtime in first: one runtime per process,
            import torch           # noqa: F401  so torch streams / events / device pointers are ours too
        except Exception:          # noqa: BLE001  (pure C users run on the system ROCm runtime)
            pass
        lib = C.CDLL(str(p))
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                continue            # declared in a later revision of the header than this .so
            fn.restype = res
            fn.argtypes = args
        if path is None:
            _lib = lib
        return lib


def last_error() -> str:
    return load_library().fad_last_error().decode("utf-8", "replace")


class FadOutOfMemory(RuntimeError):
    """The library could not get memory: FAD_ERR_ALLOC, or a HIP call that failed with hipErrorOutOfMemory.  A RuntimeError like
    torch's own out-of-memory error; callers that can fall back to a leaner route test for the TYPE, not for message wording."""


def is_out_of_memory(e: BaseException) -> bool:
    """This library's allocation failure, or torch's (torch.cuda.OutOfMemoryError / torch.OutOfMemoryError where torch has them)."""
    if isinstance(e, FadOutOfMemory):
        return True
    try:
        import torch
        for name in ("OutOfMemoryError",):
            for mod in (getattr(torch, "cuda", None), torch):
                t = getattr(mod, name, None) if mod is not None else None
                if isinstance(t, type) and isinstance(e, t):
                    return True
    except ImportError:
        pass
    return False


def check(status: int, what: str = "libfad_hip"):
    """Map a fad_status to the exception type the reference raises for the same condition."""
    if status == FAD_OK:
        return
    msg = f"{what}: {last_error()} (status {status})"
    if status in (FAD_ERR_SHAPE, FAD_ERR_TOO_FEW_ROWS):
        raise AssertionError(msg)            # fad.py:46-47, 78-81 are asserts
    if status == FAD_ERR_NOT_FINITE:
        raise ValueError(msg)                # fad.py:105 / scipy "array must not contain infs or NaNs"
    if status == FAD_ERR_NO_DEVICE:
        raise FadHipUnavailable(msg)
    if status == FAD_ERR_NOT_CONVERGED:
        log.warning(msg)
        return
    if status == FAD_ERR_ALLOC or (status == FAD_ERR_HIP and ("hipErrorOutOfMemory" in msg or "out of memory" in msg.lower())):
        raise FadOutOfMemory(msg)
    raise RuntimeError(msg)


def device_count() -> int:
    return int(load_library().fad_device_count())


def require_gpu(device: int = 0):
    n = device_count()
    if n <= device:
        raise FadHipUnavailable(
            f"no gfx950 GPU visible as device {device} ({n} found): the FAD hot path only runs on the HIP library")


# ---------------------------------------------------------------------------------------------
# array plumbing: numpy (host) or torch (device) -> (pointer, n, ld, dtype code, on_device, keepalive)
# ---------------------------------------------------------------------------------------------
_NP_CODES = {np.dtype(np.float16): FAD_F16, np.dtype(np.float32): FAD_F32, np.dtype(np.float64): FAD_F64}


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def current_stream_ptr(device: Optional[int] = None) -> int:
    """hipStream_t of torch's current stream (0 = default stream; also when torch is absent)."""
    try:
        import torch
        if torch.cuda.is_available():
            return int(torch.cuda.current_stream(device).cuda_stream)
    except Exception:       # noqa: BLE001
        pass
    return 0


def rows_view(x) -> Tuple[int, int, int, int, int, int, object]:
    """-> (ptr, n, d, ld, dtype_code, on_device, keepalive) for a 2-D frame matrix."""
    if _is_torch(x):
        import torch
        if x.dim() != 2:
            raise AssertionError(f"expected a 2-D frame matrix, got shape {tuple(x.shape)}")
        codes = {torch.float16: FAD_F16, torch.bfloat16: FAD_BF16, torch.float32: FAD_F32, torch.float64: FAD_F64}
        if x.dtype not in codes:
            x = x.to(torch.float64)
        if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
            x = x.contiguous()
        if x.is_cuda:
            return x.data_ptr(), x.shape[0], x.shape[1], max(x.stride(0), x.shape[1]), codes[x.dtype], 1, x
        if x.dtype == torch.bfloat16:
            x = x.to(torch.float32)
        x = x.numpy()
    a = np.asarray(x)
    if a.ndim != 2:
        raise AssertionError(f"expected a 2-D frame matrix, got shape {a.shape}")
    if a.dtype not in _NP_CODES:
        a = a.astype(np.float64)           # np.cov promotes everything else to float64
    if a.shape[0] > 0 and (a.strides[1] != a.itemsize or a.strides[0] % a.itemsize or a.strides[0] < a.shape[1] * a.itemsize):
        a = np.ascontiguousarray(a)
    ld = a.strides[0] // a.itemsize if a.shape[0] > 1 else a.shape[1]
    return a.ctypes.data, a.shape[0], a.shape[1], max(ld, a.shape[1]), _NP_CODES[a.dtype], 0, a


def f64_host(a, shape=None) -> np.ndarray:
    out = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    if shape is not None and out.shape != tuple(shape):
        raise AssertionError(f"expected shape {tuple(shape)}, got {out.shape}")
    return out
