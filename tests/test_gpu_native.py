"""Kernel-by-kernel checks of the nine-launch Frechet chain (fadtk_amd/csrc/ns_fast.h) against host float64 arithmetic: a
gfx950 executable built from tests/native/nsfast_check.hip by `python -m fadtk_amd.build` (see the header of that file for
what is compared).  The chain as a whole is checked against the oracle in test_gpu_parity.py."""
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
EXE = Path(__file__).resolve().parent / "native" / "nsfast_check"


@pytest.mark.parametrize("dims", [["512"], ["256", "768"], ["1024", "384"]])
def test_ns_fast_kernels_against_host_arithmetic(dims):
    if not EXE.exists():
        from fadtk_amd.build import build_native_tests
        build_native_tests()
    r = subprocess.run([str(EXE), *dims], capture_output=True, text=True, timeout=900)
    print(r.stdout[-6000:])
    assert r.returncode == 0, r.stdout[-6000:] + r.stderr[-2000:]
    assert "all checks passed" in r.stdout
