"""pytest configuration: the ``gpu`` marker and import paths.

``-m "not gpu"``: oracle vs golden vectors, host logic, C-ABI symbol checks (no GPU compute).
``-m gpu``      : parity tests proper -- every compute call goes through the C-ABI HIP library.
"""
import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests" / "golden"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _cpu_quota_cores():
    """CPUs' worth of time per period the process's CPU-bandwidth cgroup grants (cgroup v2 cpu.max / v1 cfs_quota), or None."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


_BLAS_LIMIT = None


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is numpy / scipy on the host.  On the GPU boxes the process may use 16 CPUs' worth of time per 100 ms (cpu.max) while the
    # BLAS pool starts 64 threads: most of every period the whole process -- the thread that launches kernels included -- is frozen.  The
    # pool is held to what the cgroup grants (kept alive for the session in _BLAS_LIMIT).
    global _BLAS_LIMIT
    quota = _cpu_quota_cores()
    if quota and quota >= 1:
        try:
            import numpy, scipy.linalg      # noqa: F401,E401  (the limit reaches the BLAS libraries that are loaded when it is set: numpy's and scipy's)
            from threadpoolctl import threadpool_limits
            _BLAS_LIMIT = threadpool_limits(limits=max(1, int(quota)), user_api="blas")
        except Exception:
            _BLAS_LIMIT = None


def _gpu_present() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU visible in this process")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden():
    import json
    return json.loads((ROOT / "tests" / "golden" / "golden.json").read_text())


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"
