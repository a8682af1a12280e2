#!/usr/bin/env python3
"""HISTORICAL (round 1): the FAD_MOM_* compile-time switches this script drives were moved out of the library with the kernel
generations they belonged to (scripts/probes/moments_generations.hip); it no longer builds anything useful and is kept for the
record of how the table in DESIGN.md 4.1 was obtained.

Ablation of the moments tile kernel (v4): which of {MFMA, LDS transpose reads, global->LDS loads, barrier}
bounds it?  `python scripts/probe_ablate.py build` (CPU box, hipcc) makes one library per mask under
scripts/probes/ablate/; `python scripts/probe_ablate.py` (GPU box) times the tile kernel of each at config 3.
Mask bits 8..10 select the LDS ring depth of v4 (256 * NST; 0 = default 4), bit 11 (2048) spreads v8's loads
between its MFMAs, bits 12..16 (4096 * aux) set the cache-policy bits of v8's LDS-DMA loads, bit 17 (131072) plans three
workgroups per CU for v4 (combine with NST = 3: 131072 + 768).
Masks: 1 = no MFMA, 2 = no LDS reads, 4 = no global loads, 8 = no barrier (results are garbage by design)."""
import ctypes as C, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
OUT = ROOT / "scripts" / "probes" / "ablate"
MASKS = [int(m) for m in os.environ.get('ABLATE_MASKS', '0,1,2,4,3,5,6,7,8,9,15').split(',')]


def build():
    from fadtk_amd import build as B
    B.build_library()
    OUT.mkdir(parents=True, exist_ok=True)
    objs = [str(B.PKG / "build" / f"{Path(s).stem}.o") for s in B.SOURCES if s != "moments.hip"]
    tl = B._torch_lib_dir()
    for m in MASKS:
        o = OUT / f"moments_{m}.o"
        subprocess.run([B._hipcc(), *B.FLAGS, f"-DFAD_MOM_ABLATE={m & 63}", f"-DFAD_MOM_NST={((m >> 8) & 7) or 4}", f"-DFAD_MOM_SPREAD={(m >> 11) & 1}", f"-DFAD_MOM_WGPCU={3 if (m >> 17) & 1 else 2}", f"-DFAD_MOM_AUX={(m >> 12) & 31}", "-c", str(B.CSRC / "moments.hip"), "-o", str(o)], check=True)
        subprocess.run(["g++", "-shared", "-fPIC", "-o", str(OUT / f"libfad_ablate_{m}.so"), str(o), *objs, f"-L{tl}", "-lamdhip64",
                        f"-Wl,-rpath,{tl}", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--enable-new-dtags"], check=True)
        o.unlink()
        print("built mask", m, flush=True)


def run():
    import torch
    from fadtk_amd import _capi
    shapes = [(100_000, 512), (1_000_000, 512), (100_000, 1024)] if not os.environ.get('ABLATE_MASKS') else [(100_000, 512)]
    xs = {s: torch.randn(s, device="cuda", dtype=torch.float16) for s in shapes}
    for m in MASKS:
        lib = _capi.load_library(OUT / f"libfad_ablate_{m}.so")
        line = f"mask={m:2d} [{'noMFMA ' if m & 1 else ''}{'noLDSrd ' if m & 2 else ''}{'noGLD ' if m & 4 else ''}{'noBAR' if m & 8 else ''}]".ljust(40)
        for (n, d) in shapes:
            h = C.c_void_p()
            assert lib.fad_moments_create(d, 0, C.byref(h)) == 0
            x = xs[(n, d)]
            for _ in range(3):
                assert lib.fad_moments_update(h, x.data_ptr(), n, d, 0, 1, None) == 0
            lib.fad_moments_set_timing(h, 1)
            if m & 16:
                torch.cuda.synchronize(); print(f"--- mask {m}: per-workgroup clocks of one launch", flush=True)
            for _ in range(1 if m & 16 else 10):
                lib.fad_moments_update(h, x.data_ptr(), n, d, 0, 1, None)
            torch.cuda.synchronize()
            k, r, v = C.c_float(), C.c_float(), C.c_int()
            lib.fad_moments_last_timing(h, C.byref(k), C.byref(r), C.byref(v))
            line += f"  N={n} D={d}: {k.value*1e3:7.1f} us"
            lib.fad_moments_destroy(h)
        print(line, flush=True)


if __name__ == "__main__":
    build() if sys.argv[1:] == ["build"] else run()
