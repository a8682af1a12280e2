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


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
