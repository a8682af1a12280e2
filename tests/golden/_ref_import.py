"""Import the read-only reference (``/root/reference/fadtk``) inside THIS container only.

Used solely by ``make_golden.py`` to generate fixtures.  The reference's modules import
packages that are not installed here (torchaudio, hypy_utils, soundfile, librosa); they
are replaced by inert stand-in modules *in sys.modules of the generator process only* so
that the reference's pure numpy/scipy functions (fad.py:42-120, utils.py:13-46) can run
unmodified.  Nothing from the reference is copied; the fixtures hold inputs/outputs only.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("FADTK_REFERENCE_ROOT", "/root/reference")


def _serial_map(fn, items, *a, **k):
    return [fn(x) for x in items]


def _tq(it, *a, **k):
    return it


def _write(path, text):
    from pathlib import Path
    p = Path(path)
    p.parent.mkdir(parents=True, exist_ok=True)
    p.write_text(text)


def import_reference():
    """Return (fad_module, utils_module) of the reference, with stand-ins for absent deps."""
    sys.dont_write_bytecode = True
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    import logging

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    hy = mod("hypy_utils", write=_write)
    hy.tqdm_utils = mod("hypy_utils.tqdm_utils", tq=_tq, tmap=_serial_map, pmap=_serial_map)
    hy.logging_utils = mod("hypy_utils.logging_utils",
                           setup_logger=lambda *a, **k: logging.getLogger("fadtk-ref"))
    hy.nlp_utils = mod("hypy_utils.nlp_utils", substr_between=lambda s, a, b: "")
    hy.downloader = mod("hypy_utils.downloader", download_file=lambda *a, **k: None)
    for name in ("torchaudio", "soundfile", "librosa"):
        if name not in sys.modules:
            mod(name)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import fadtk.fad as ref_fad          # noqa: E402
    import fadtk.utils as ref_utils      # noqa: E402
    return ref_fad, ref_utils
