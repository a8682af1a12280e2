"""Command lines of the engine: `fadtk` (score), `fadtk.embeds` (cache embeddings), `fadtk.package` (stats npz).

The flags, positional arguments, printed lines and CSV schema are those of the reference launchers
(fadtk/__main__.py:15-32, fadtk/embeds.py:11-21, fadtk/package.py:13-21) so that existing scripts keep working;
``--gpus N`` is new: it re-launches the same command under torch.distributed.run with one process per GPU.
"""
from __future__ import annotations

import logging
import os
import subprocess
import sys
import time
from argparse import ArgumentParser
from pathlib import Path

from . import dist

log = logging.getLogger("fadtk_amd")
CSV_HEADER = "model,baseline,eval,score,inf_r2,time\n"


def _registry():
    from .model_loader import get_all_models
    return {m.name: m for m in get_all_models()}


def _parser(prog: str, positionals=(), with_models: bool = False) -> ArgumentParser:
    p = ArgumentParser(prog=prog)
    for name, kw in positionals:
        p.add_argument(name, **kw)
    p.add_argument("-w", "--workers", type=int, default=8, help="host threads decoding audio ahead of the GPU")
    p.add_argument("-s", "--sox-path", type=str, default="/usr/bin/sox", help="accepted for compatibility, unused")
    p.add_argument("--gpus", type=int, default=1, help="shard the work over this many GPUs (one process each)")
    return p


def _relaunch(gpus: int, module: str):
    """--gpus N given outside torchrun: start N ranks of the same command and exit with their status."""
    if gpus <= 1 or dist.env_world() > 1:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29533"), "-m", module,
           *sys.argv[1:]]
    raise SystemExit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")))


def _setup_logging():
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")


# ----------------------------------------------------------------------------------------------- fadtk
def score_main():
    from .fad import FrechetAudioDistance
    from .fad_batch import cache_embedding_files
    _setup_logging()
    models = _registry()
    p = _parser("fadtk", positionals=(
        ("model", dict(type=str, choices=list(models), help="embedding model")),
        ("baseline", dict(type=str, help="baseline dataset: directory, statistics .npz, or a bundled name")),
        ("eval", dict(type=str, help="directory (or statistics) to evaluate")),
        ("csv", dict(type=str, nargs="?", help="append the result to this CSV; with --indiv: where per-song scores "
                                                "go (default fad-individual-results.csv)"))))
    p.add_argument("--inf", action="store_true", help="FAD-infinity extrapolation")
    p.add_argument("--indiv", action="store_true", help="one score per song of the eval directory")
    p.add_argument("--fused-stats", action="store_true",
                   help="accumulate dataset statistics on the GPU while embedding (one all-reduce across GPUs) "
                        "instead of re-reading the cached .npy files")
    a = p.parse_args()
    _relaunch(a.gpus, "fadtk_amd")
    model = models[a.model]

    for dataset in (a.baseline, a.eval):                      # 1. embeddings (file-sharded over ranks)
        if Path(dataset).is_dir():
            if a.fused_stats and not (Path(dataset) / "stats" / model.name).exists():
                from .fad_batch import embed_and_accumulate
                embed_and_accumulate(dataset, model, workers=a.workers)
            else:
                cache_embedding_files(dataset, model, workers=a.workers)
    if dist.rank() != 0 and not a.indiv:                      # one score is one small problem: rank 0 finishes
        return

    fad = FrechetAudioDistance(model, audio_load_worker=a.workers, load_model=False, device=dist.env_local_rank())
    inf_r2 = None
    if a.inf:                                                 # 2. the score
        assert Path(a.eval).is_dir(), "FAD-inf requires a directory as the evaluation dataset"
        result = fad.score_inf(a.baseline, list(Path(a.eval).glob("*.*")))
        print("FAD-inf Information:", result)
        score, inf_r2 = result.score, result.r2
    elif a.indiv:
        assert Path(a.eval).is_dir(), "Individual FAD requires a directory as the evaluation dataset"
        out = Path(a.csv or "fad-individual-results.csv")
        fad.score_individual(a.baseline, a.eval, out)         # songs are sharded over ranks inside
        log.info(f"Individual FAD scores saved to {out}")
        return
    else:
        score = fad.score(a.baseline, a.eval)

    log.info("FAD computed.")                                 # 3. report
    if a.csv:
        target = Path(a.csv)
        target.parent.mkdir(parents=True, exist_ok=True)
        if not target.is_file():
            target.write_text(CSV_HEADER)
        with open(target, "a") as fh:
            fh.write(f"{model.name},{a.baseline},{a.eval},{score},{inf_r2},{time.time()}\n")
        log.info(f"FAD score appended to {a.csv}")
    log.info(f"The FAD {model.name} score between {a.baseline} and {a.eval} is: {score}")


# ----------------------------------------------------------------------------------------------- fadtk.embeds
def embeds_main():
    from .fad_batch import cache_embedding_files
    _setup_logging()
    models = _registry()
    p = _parser("fadtk.embeds")
    p.add_argument("-m", "--models", type=str, choices=list(models), nargs="+", required=True)
    p.add_argument("-d", "--dirs", type=str, nargs="+", required=True)
    a = p.parse_args()
    _relaunch(a.gpus, "fadtk_amd.embeds")
    for name in a.models:
        for folder in a.dirs:
            log.info(f"Caching embeddings for {folder} using {name}")
            cache_embedding_files(folder, models[name], workers=a.workers)


# ----------------------------------------------------------------------------------------------- fadtk.package
def package_main():
    import numpy as np
    from .fad import FrechetAudioDistance
    from .fad_batch import cache_embedding_files
    _setup_logging()
    models = _registry()
    p = _parser("fadtk.package", positionals=(("directory", dict(type=str)), ("out", dict(type=str))))
    p.add_argument("-m", "--models", type=str, nargs="*", default=None, help="subset of models (default: all registered)")
    a = p.parse_args()
    out = Path(a.out)
    if out.suffix != ".npz":
        print("The output file you specified is not a npz file, are you sure? (y/N)")
        if input().lower() != "y":
            raise SystemExit(1)
    stats = {}
    for name, model in models.items():
        if a.models and name not in a.models:
            continue
        cache_embedding_files(a.directory, model, workers=a.workers)
        mu, cov = FrechetAudioDistance(model, load_model=False).load_stats(a.directory)
        stats[f"{name}.mu"], stats[f"{name}.cov"] = mu, cov       # the key format load_stats reads (fad.py:264-266)
    np.savez(out, **stats)
