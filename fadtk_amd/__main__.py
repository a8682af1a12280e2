"""`python -m fadtk_amd <model> <baseline> <eval> [csv] [-w N] [-s sox] [--inf] [--indiv]`

Same arguments, printed lines and CSV schema as fadtk's launcher (fadtk/__main__.py:9-70); adds
``--gpus N`` which re-launches the command under torch.distributed.run, one process per GPU.
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


def relaunch_if_needed(gpus: int, module: str) -> bool:
    """--gpus N outside torchrun: exec `python -m torch.distributed.run ... -m <module> <same args>`."""
    if gpus <= 1 or dist.env_world() > 1:
        return False
    port = os.environ.get("MASTER_PORT", "29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", port, "-m", module, *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    from .fad import FrechetAudioDistance
    from .fad_batch import cache_embedding_files
    from .model_loader import get_all_models
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    models = {m.name: m for m in get_all_models()}

    agupa = ArgumentParser(prog="fadtk")
    agupa.add_argument("model", type=str, choices=list(models.keys()), help="The embedding model to use")
    agupa.add_argument("baseline", type=str, help="The baseline dataset")
    agupa.add_argument("eval", type=str, help="The directory to evaluate against")
    agupa.add_argument("csv", type=str, nargs="?",
                       help="The CSV file to append results to. If this argument is not supplied, single-value "
                            "results will be printed to stdout, and for --indiv, the results will be saved to "
                            "'fad-individual-results.csv'")
    agupa.add_argument("-w", "--workers", type=int, default=8)
    agupa.add_argument("-s", "--sox-path", type=str, default="/usr/bin/sox")       # accepted, unused (as in fadtk)
    agupa.add_argument("--inf", action="store_true", help="Use FAD-inf extrapolation")
    agupa.add_argument("--indiv", action="store_true",
                       help="Calculate FAD for individual songs and store the results in the given file")
    agupa.add_argument("--gpus", type=int, default=1, help="GPUs (processes) to shard embedding extraction over")
    args = agupa.parse_args()
    relaunch_if_needed(args.gpus, "fadtk_amd")

    model = models[args.model]
    baseline, eval = args.baseline, args.eval

    # 1. embeddings for both datasets (sharded over ranks when launched with several GPUs)
    for d in [baseline, eval]:
        if Path(d).is_dir():
            cache_embedding_files(d, model, workers=args.workers)
    if dist.rank() != 0 and not args.indiv:   # one score is one small problem: rank 0 finishes the job
        return                                # (--indiv shards the songs over all ranks instead)

    # 2. FAD
    fad = FrechetAudioDistance(model, audio_load_worker=args.workers, load_model=False, device=dist.env_local_rank())
    if args.inf:
        assert Path(eval).is_dir(), "FAD-inf requires a directory as the evaluation dataset"
        score = fad.score_inf(baseline, list(Path(eval).glob("*.*")))
        print("FAD-inf Information:", score)
        score, inf_r2 = score.score, score.r2
    elif args.indiv:
        assert Path(eval).is_dir(), "Individual FAD requires a directory as the evaluation dataset"
        csv_path = Path(args.csv or "fad-individual-results.csv")
        fad.score_individual(baseline, eval, csv_path)
        log.info(f"Individual FAD scores saved to {csv_path}")
        return
    else:
        score = fad.score(baseline, eval)
        inf_r2 = None

    # 3. results
    log.info("FAD computed.")
    if args.csv:
        Path(args.csv).parent.mkdir(parents=True, exist_ok=True)
        if not Path(args.csv).is_file():
            Path(args.csv).write_text("model,baseline,eval,score,inf_r2,time\n")
        with open(args.csv, "a") as f:
            f.write(f"{model.name},{baseline},{eval},{score},{inf_r2},{time.time()}\n")
        log.info(f"FAD score appended to {args.csv}")
    log.info(f"The FAD {model.name} score between {baseline} and {eval} is: {score}")


if __name__ == "__main__":
    main()
