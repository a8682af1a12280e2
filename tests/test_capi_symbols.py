"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol that
include/fad_hip.h declares; without a GPU every compute entry fails loudly (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (ROOT / "include" / "fad_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fad_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared_functions()
    for must in ("fad_moments_create", "fad_moments_update", "fad_moments_update_segmented", "fad_moments_finalize",
                 "fad_moments_export", "fad_moments_import", "fad_frechet", "fad_frechet_from_moments",
                 "fad_frechet_batched_vs_baseline", "fad_last_error", "fad_version"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from fadtk_amd import _capi
    if not _capi.LIB_PATH.exists():
        from fadtk_amd.build import build_library
        build_library(verbose=False)
    lib = _capi.load_library()
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, f"declared in fad_hip.h but not exported by libfad_hip.so: {missing}"
    assert lib.fad_version() == 2
    unbound = [n for n in _declared_functions() if n not in _capi.SIGNATURES]
    assert not unbound, f"no ctypes prototype for {unbound}"


def test_no_cpu_fallback_without_gpu():
    """On a GPU-less box the product path must raise, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from fadtk_amd import _capi, calc_embd_statistics, calc_frechet_distance
    assert _capi.device_count() == 0
    x = np.random.default_rng(0).standard_normal((32, 8)).astype(np.float16)
    with pytest.raises(_capi.FadHipUnavailable):
        calc_embd_statistics(x)
    with pytest.raises(_capi.FadHipUnavailable):
        calc_frechet_distance(np.zeros(8), np.eye(8), np.zeros(8), np.eye(8))
    lib = _capi.load_library()
    h = C.c_void_p()
    assert lib.fad_moments_create(8, 0, C.byref(h)) == _capi.FAD_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.fad_last_error() or b"no HIP device" in lib.fad_last_error()


def test_product_package_never_imports_the_oracle():
    for py in (ROOT / "fadtk_amd").rglob("*.py"):
        src = py.read_text()
        assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", ""), py
