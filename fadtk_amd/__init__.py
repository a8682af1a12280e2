"""fadtk_amd -- MI355X-native Frechet Audio Distance engine behind fadtk's plugin surface.

The FAD hot path (running mean/covariance of embedding frames, matrix square root of
Sigma1 Sigma2, batched per-song scores, log-mel front ends) runs in hand-written HIP for gfx950
(`fadtk_amd/csrc`, C ABI in `include/fad_hip.h`).  This package is the host-side mirror of
fadtk's public interface over that library.
"""
from .fad import FADInfResults, FrechetAudioDistance, calc_embd_statistics, calc_frechet_distance  # noqa: F401
from .utils import (PathLike, calculate_embd_statistics_online, dataset_statistics, find_sox_formats,  # noqa: F401
                    get_cache_embedding_path)

__version__ = "0.1.0"


def __getattr__(name):
    # loaders and the batch driver import torch; keep `import fadtk_amd` light
    if name in ("ModelLoader", "get_all_models", "VGGishModel", "EncodecEmbModel", "CLAPLaionModel", "WhisperModel",
                "W2V2Model", "HuBERTModel", "WavLMModel", "MERTModel"):
        from . import model_loader
        return getattr(model_loader, name)
    if name == "cache_embedding_files":
        from .fad_batch import cache_embedding_files
        return cache_embedding_files
    raise AttributeError(name)
