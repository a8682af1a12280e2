"""`python -m fadtk_amd.embeds -m MODEL [MODEL ...] -d DIR [DIR ...] [-w N] [--gpus N]`

Cache embeddings of directories with several models (fadtk/embeds.py:5-27)."""
from __future__ import annotations

import logging
from argparse import ArgumentParser


def main():
    from .__main__ import relaunch_if_needed
    from .fad_batch import cache_embedding_files
    from .model_loader import get_all_models
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s")
    log = logging.getLogger("fadtk_amd")
    models = {m.name: m for m in get_all_models()}

    agupa = ArgumentParser(prog="fadtk.embeds")
    agupa.add_argument("-m", "--models", type=str, choices=list(models.keys()), nargs="+", required=True)
    agupa.add_argument("-d", "--dirs", type=str, nargs="+", required=True)
    agupa.add_argument("-w", "--workers", type=int, default=8)
    agupa.add_argument("-s", "--sox-path", type=str, default="/usr/bin/sox")
    agupa.add_argument("--gpus", type=int, default=1)
    args = agupa.parse_args()
    relaunch_if_needed(args.gpus, "fadtk_amd.embeds")

    for model_name in args.models:
        for d in args.dirs:
            log.info(f"Caching embeddings for {d} using {model_name}")
            cache_embedding_files(d, models[model_name], workers=args.workers)


if __name__ == "__main__":
    main()
