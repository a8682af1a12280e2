"""`python -m fadtk_amd.package <directory> <out.npz> [-w N]` -- statistics packager (fadtk/package.py:7-42).

Embeds ``directory`` with every registered model and stores ``{model}.mu`` / ``{model}.cov`` in one npz --
the format ``FrechetAudioDistance.load_stats`` reads for baselines such as fma_pop."""
from __future__ import annotations

from argparse import ArgumentParser
from pathlib import Path

import numpy as np


def main():
    from .fad import FrechetAudioDistance
    from .fad_batch import cache_embedding_files
    from .model_loader import get_all_models
    agupa = ArgumentParser(prog="fadtk.package")
    agupa.add_argument("directory", type=str)
    agupa.add_argument("out", type=str)
    agupa.add_argument("-w", "--workers", type=int, default=8)
    agupa.add_argument("-s", "--sox-path", type=str, default="/usr/bin/sox")
    agupa.add_argument("-m", "--models", type=str, nargs="*", default=None, help="subset of models (default: all)")
    args = agupa.parse_args()

    out = Path(args.out)
    if out.suffix != ".npz":
        print("The output file you specified is not a npz file, are you sure? (y/N)")
        if input().lower() != "y":
            raise SystemExit(1)
    models = [m for m in get_all_models() if not args.models or m.name in args.models]
    data = {}
    for model in models:
        cache_embedding_files(args.directory, model, workers=args.workers)
        mu, cov = FrechetAudioDistance(model, load_model=False).load_stats(args.directory)
        data[f"{model.name}.mu"] = mu
        data[f"{model.name}.cov"] = cov
    np.savez(out, **data)


if __name__ == "__main__":
    main()
