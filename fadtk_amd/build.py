"""Build libfad_hip.so (the C-ABI HIP library) in-tree for gfx950.

    python -m fadtk_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so lives next to the package
(fadtk_amd/lib/libfad_hip.so) so that it travels with the source tree and is never
pip-installed.  Also builds nothing else: there is no CPU fallback library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libfad_hip.so"
ARCH = "gfx950"
SOURCES = ["common.cpp", "host_stage.cpp", "moments.hip", "gemm_f64.hip", "gemm_f32.hip", "frechet_f64.hip", "frechet.hip", "frechet_songs.hip", "logmel.hip", "resample.hip"]
HEADERS = [*sorted(CSRC.glob("*.h")), PKG.parent / "include" / "fad_hip.h"]
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", *os.environ.get("FAD_EXTRA_HIPCC_FLAGS", "").split(), "-x", "hip"]      # (ablation builds: -DFAD_BIG_ABL_...)


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found: cannot build libfad_hip.so")


def _torch_lib_dir():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            d = Path(spec.origin).parent / "lib"
            return str(d) if d.exists() else None
    except Exception:       # noqa: BLE001
        pass
    return None


def _stale(out: Path, deps) -> bool:
    if not out.exists():
        return True
    t = out.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> Path:
    srcs = [CSRC / s for s in SOURCES if (CSRC / s).exists()]
    objdir = PKG / "build"
    objdir.mkdir(exist_ok=True)
    LIBDIR.mkdir(exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    for s in srcs:
        o = objdir / (s.stem + ".o")
        if force or _stale(o, [s, *HEADERS]):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc, *FLAGS, "-c", str(s), "-o", str(o)]
        if verbose:
            print("[fadtk_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {s.name}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [objdir / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        # ONE HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64.so (no SONAME,
        # found by file name).  Linking against that copy makes our NEEDED entry "libamdhip64.so", which
        # the loader resolves to torch's already-loaded runtime -- so torch's streams, events and device
        # pointers are valid inside the library.  Without torch the system ROCm runtime is used.
        link_dirs = [d for d in (_torch_lib_dir(), "/opt/rocm/lib") if d and (Path(d) / "libamdhip64.so").exists()]
        cmd = [os.environ.get("CXX", "g++"), "-shared", "-fPIC", "-o", str(LIB), *map(str, objs),
               f"-L{link_dirs[0]}", "-lamdhip64", "-ldl", *[f"-Wl,-rpath,{d}" for d in link_dirs], "-Wl,--enable-new-dtags"]
        if verbose:
            print("[fadtk_amd.build]", " ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return LIB


NATIVE_TESTS = PKG.parent / "tests" / "native"


def build_native_tests(force: bool = False, verbose: bool = True) -> list:
    """tests/native/*.hip -> tests/native/<name> (gfx950 executables that check single kernels of csrc/ against host
    arithmetic; run by `pytest -m gpu`).  Test infrastructure, not part of the library."""
    out = []
    for src in sorted(NATIVE_TESTS.glob("*.hip")):
        exe = src.with_suffix("")
        if force or _stale(exe, [src, *HEADERS]):
            cmd = [_hipcc(), f"--offload-arch={ARCH}", "-O2", "-std=c++17", "-o", str(exe), str(src)]
            if verbose:
                print("[fadtk_amd.build]", " ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {src.name}:\n{r.stdout}\n{r.stderr}")
        out.append(exe)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
    print(build_native_tests(force="--force" in sys.argv))
