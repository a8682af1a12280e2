"""Drop-in alias: ``import fadtk`` / ``python -m fadtk`` / ``python -m fadtk.embeds`` resolve to the
MI355X engine (fadtk_amd) so existing scripts and ModelLoader plugins run unchanged."""
from fadtk_amd import *                                           # noqa: F401,F403
from fadtk_amd import (FADInfResults, FrechetAudioDistance, calc_embd_statistics,        # noqa: F401
                       calc_frechet_distance, calculate_embd_statistics_online, get_cache_embedding_path)
from fadtk_amd.fad_batch import cache_embedding_files             # noqa: F401
from fadtk_amd.model_loader import *                              # noqa: F401,F403
from fadtk_amd.model_loader import ModelLoader, get_all_models    # noqa: F401
