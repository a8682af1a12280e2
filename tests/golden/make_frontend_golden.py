#!/usr/bin/env python3
"""Golden vectors for the log-mel front ends, produced by the INSTALLED third-party extractors
(transformers WhisperFeatureExtractor / ClapFeatureExtractor) -- the same arithmetic the packages
pinned by fadtk's uv.lock run.  Run in the authoring container:

    python tests/golden/make_frontend_golden.py

Stores only a decimated subset of each feature matrix (every 7th frame / all mels) plus checksums:
fixtures stay small and the test still pins every filter and the normalisation.
The VGGish front end has no installed third-party implementation to run (torch.hub, no network):
it stays "parity unpinned" and is only cross-checked against oracle/logmel_oracle.py.
"""
import json
from pathlib import Path

import numpy as np
import transformers
from transformers import ClapFeatureExtractor, WhisperFeatureExtractor

HERE = Path(__file__).resolve().parent


import sys
sys.path.insert(0, str(HERE))
from recipes import audio_clip as clip          # noqa: E402

arrays, meta = {}, {"transformers": transformers.__version__, "numpy": np.__version__, "cases": []}
wfe = WhisperFeatureExtractor()
for k, (seed, secs) in enumerate(((300, 3.0), (301, 30.0), (302, 31.5), (303, 0.37))):
    x = clip(seed, int(16000 * secs), 16000)
    feats = wfe(x, sampling_rate=16000, return_tensors="np").input_features[0]       # [80, 3000]
    arrays[f"whisper{k}"] = feats[:, ::7].astype(np.float32)
    meta["cases"].append({"name": f"whisper{k}", "seed": seed, "n": len(x), "sr": 16000, "stride": 7,
                          "sum": float(feats.astype(np.float64).sum()), "shape": list(feats.shape)})

cfe = ClapFeatureExtractor(frequency_min=50, frequency_max=14000)   # laion-clap audio_cfg: fmin=50, fmax=14000
for k, seed in enumerate((310, 311)):
    x = clip(seed, 480000, 48000)
    feats = cfe._np_extract_fbank_features(x, cfe.mel_filters_slaney)                 # [1001, 64]
    arrays[f"htsat{k}"] = feats[::7].astype(np.float32)
    meta["cases"].append({"name": f"htsat{k}", "seed": seed, "n": len(x), "sr": 48000, "stride": 7,
                          "sum": float(np.asarray(feats, dtype=np.float64).sum()), "shape": list(feats.shape)})

np.savez_compressed(HERE / "g9_frontend.npz", **arrays)
(HERE / "g9_frontend.json").write_text(json.dumps(meta, indent=1))
print("wrote g9_frontend.npz/json", {k: v.shape for k, v in arrays.items()})
