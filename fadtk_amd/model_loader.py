"""ModelLoader plugin surface (fadtk/model_loader.py:21-86) and the loaders on the measured path.

The contract is the reference's: subclasses set (name, num_features, sr, min_len), implement
``load_model`` and ``_get_embedding(audio) -> [n_frames, n_features]`` tensor; ``get_embedding``
returns a CPU numpy array, float32 cast to float16 (model_loader.py:40-50) -- that cast defines
the dtype of every matrix the FAD kernels see.  Loaders stay picklable before ``load_model``.

What is different underneath:
  * the network forward passes run on PyTorch-ROCm (plumbing), but the STFT / log-mel front end of
    VGGish, Whisper and CLAP-HTSAT is the hand-written HIP kernel of csrc/logmel.hip, batched over
    all windows of a file, instead of per-file numpy / torch-CPU code inside third-party packages;
  * nothing here touches the network: constructors never download, weights come from local files
    (``FADTK_AMD_CHECKPOINTS`` or the Hugging Face cache, ``local_files_only``) and, when none are
    found, from a seeded random initialisation if ``FADTK_AMD_RANDOM_WEIGHTS=1`` / ``random_init=True``
    (synthetic benchmarks and tests; such embeddings are not comparable with fadtk's).
"""
from __future__ import annotations

import logging
import os
from abc import ABC, abstractmethod
from pathlib import Path
from typing import List, Literal, Optional

import numpy as np
import torch
from torch import nn

from . import hip
from .audio import read_audio, read_pcm16, resample_kaiser

log = logging.getLogger("fadtk_amd")


def _allow_random(flag: Optional[bool]) -> bool:
    return bool(flag) if flag is not None else os.environ.get("FADTK_AMD_RANDOM_WEIGHTS", "0") == "1"


def _checkpoint_dir() -> Path:
    return Path(os.environ.get("FADTK_AMD_CHECKPOINTS", Path.home() / ".cache" / "fadtk_amd"))


class ModelLoader(ABC):
    """Load a model and get embeddings from it (fadtk/model_loader.py:21-86)."""

    def __init__(self, name: str, num_features: int, sr: int, min_len: int = -1):
        self.model = None
        self.sr = sr
        self.num_features = num_features
        self.name = name
        self.min_len = min_len
        self.device = torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")

    def get_embedding(self, audio: np.ndarray):
        embd = self._get_embedding(audio)
        if embd.is_cuda:
            embd = embd.cpu()
        embd = embd.detach().numpy()
        if embd.dtype == np.float32:                  # space-efficient storage (model_loader.py:47-48)
            embd = embd.astype(np.float16)
        return embd

    @abstractmethod
    def load_model(self):
        pass

    @abstractmethod
    def _get_embedding(self, audio: np.ndarray):
        """Returns the embedding of the audio, shape (n_frames, n_features)."""
        pass

    # How many files the batch driver (fad_batch.py) hands to ``_get_embedding_batch`` at once.  The reference's loop is one file per
    # forward (fad_batch.py:18-22, fad.py:188-201); a loader that can do better says so here.  1 keeps that loop for third-party loaders.
    batch_files = 1

    def _get_embedding_batch(self, audios: list) -> list:
        """Embeddings of several files: a list of (n_frames_i, n_features) tensors, one per input, each what ``_get_embedding`` returns
        for that file alone.  Not part of the reference's plugin contract -- the default is the reference's own loop, so a loader written
        for fadtk works unchanged; the loaders below override it with ONE front-end launch and ONE forward over all files."""
        return [self._get_embedding(a) for a in audios]

    def load_wav(self, wav_file: Path):
        wav_data, _ = read_pcm16(wav_file)
        wav_data = wav_data / 32768.0                 # int16 -> [-1, 1) float64 (model_loader.py:64-65)
        return self.enforce_min_len(wav_data)

    def enforce_min_len(self, audio: np.ndarray) -> np.ndarray:
        if self.min_len < 0:
            return audio
        if audio.shape[0] < self.min_len * self.sr:
            log.warning(f"Audio is too short for {self.name}.\n"
                        f"The model requires a minimum length of {self.min_len}s, audio is "
                        f"{audio.shape[0] / self.sr:.2f}s.\nPadding with zeros.")
            audio = np.pad(audio, (0, int(np.ceil(self.min_len * self.sr - audio.shape[0]))))
        return audio

    # -- helpers shared by the concrete loaders
    def _device_index(self) -> int:
        return self.device.index or 0 if self.device.type == "cuda" else 0

    def _seed(self) -> int:
        import zlib
        return zlib.crc32(self.name.encode()) % (2 ** 31)


# ---------------------------------------------------------------------------------------------
# VGGish (model_loader.py:89-108)
# ---------------------------------------------------------------------------------------------
class _VGGishNet(nn.Module):
    """The published VGGish architecture: 6 conv + 3 FC, input [N, 1, 96, 64] log-mel patches."""

    def __init__(self, strip_last_relu: bool = True):
        super().__init__()
        layers, cin = [], 1
        for v in (64, "M", 128, "M", 256, 256, "M", 512, 512, "M"):
            if v == "M":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        self.features = nn.Sequential(*layers)
        emb = [nn.Linear(512 * 4 * 6, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True), nn.Linear(4096, 128)]
        if not strip_last_relu:
            emb.append(nn.ReLU(True))
        self.embeddings = nn.Sequential(*emb)

    def forward(self, x):
        x = self.features(x)
        x = x.permute(0, 2, 3, 1).contiguous().view(x.size(0), -1)     # channels last before flattening
        return self.embeddings(x)


class VGGishModel(ModelLoader):
    """S. Hershey et al., "CNN Architectures for Large-Scale Audio Classification", ICASSP 2017.
    128-d embedding per 0.96 s example, last ReLU stripped and PCA off like the reference."""

    def __init__(self, use_pca=False, use_activation=False, random_init: Optional[bool] = None):
        super().__init__("vggish", 128, 16000, min_len=1)
        self.use_pca = use_pca
        self.use_activation = use_activation
        self.random_init = random_init

    def load_model(self):
        if self.use_pca:
            raise NotImplementedError("the PCA post-processor is off in fadtk's registry and not provided here")
        self.model = _VGGishNet(strip_last_relu=not self.use_activation)
        ckpt = next((p for p in (_checkpoint_dir() / "vggish-10086976.pth",
                                 Path.home() / ".cache/torch/hub/checkpoints/vggish-10086976.pth") if p.exists()), None)
        if ckpt is not None:
            # the published checkpoint also carries the PCA post-processor (pproc.*), which fadtk switches off; every
            # other key must match: a renamed / missing layer would otherwise keep its random initialisation silently
            state = {k: v for k, v in torch.load(ckpt, map_location="cpu").items() if not k.startswith("pproc.")}
            self.model.load_state_dict(state, strict=True)
        elif _allow_random(self.random_init):
            log.warning("vggish: no local checkpoint, using seeded random weights (synthetic runs only)")
            g = torch.Generator().manual_seed(self._seed())
            with torch.no_grad():                      # He initialisation: activations keep their scale through the stack
                for p in self.model.parameters():
                    if p.dim() > 1:
                        p.copy_(torch.randn(p.shape, generator=g) * (2.0 / p[0].numel()) ** 0.5)
                    else:
                        p.copy_(0.05 * torch.randn(p.shape, generator=g))
        else:
            raise FileNotFoundError("vggish checkpoint not found (set FADTK_AMD_CHECKPOINTS or FADTK_AMD_RANDOM_WEIGHTS=1)")
        self.model.eval().to(self.device)

    def _get_embedding(self, audio: np.ndarray):
        return self._get_embedding_batch([audio])[0]

    batch_files = 64
    _forward_examples = 1024                  # examples per forward: the first convolution keeps 1.5 MB of activations per example

    def _get_embedding_batch(self, audios: list) -> list:
        # all files' samples in ONE upload, all files' examples from ONE launch of the HIP front end, the network over them in slices
        arrs = [np.asarray(a, dtype=np.float32).reshape(-1) for a in audios]
        flat = torch.from_numpy(np.concatenate(arrs)) if len(arrs) > 1 else torch.from_numpy(np.ascontiguousarray(arrs[0]))
        if self.device.type == "cuda":
            flat = flat.to(self.device)
        lens = [len(a) for a in arrs]
        wavs = list(torch.split(flat, lens))
        examples, ex_off = hip.logmel_vggish(wavs, device=self._device_index())   # HIP front end -> [E, 96, 64], file i = examples ex_off[i] : ex_off[i + 1]
        if not torch.is_tensor(examples):
            examples = torch.from_numpy(examples).to(self.device)
        with torch.no_grad():
            x = examples.unsqueeze(1)
            out = torch.cat([self.model(x[o:o + self._forward_examples]) for o in range(0, x.shape[0], self._forward_examples)]) \
                if x.shape[0] > self._forward_examples else self.model(x)
        return [out[int(ex_off[i]):int(ex_off[i + 1])] for i in range(len(arrs))]


# ---------------------------------------------------------------------------------------------
# Encodec embeddings (model_loader.py:111-186)
# ---------------------------------------------------------------------------------------------
class EncodecEmbModel(ModelLoader):
    """Continuous SEANet-encoder output of Encodec: 128 features, 75 frames/s at 24 kHz."""

    def __init__(self, variant: Literal["48k", "24k"] = "24k", random_init: Optional[bool] = None):
        super().__init__("encodec-emb" if variant == "24k" else f"encodec-emb-{variant}", 128,
                         sr=24000 if variant == "24k" else 48000)
        self.variant = variant
        self.random_init = random_init

    def load_model(self):
        from transformers import EncodecConfig, EncodecModel
        hub_id = "facebook/encodec_24khz" if self.variant == "24k" else "facebook/encodec_48khz"
        try:
            self.model = EncodecModel.from_pretrained(hub_id, local_files_only=True)
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            log.warning(f"{self.name}: no local weights, using seeded random weights (synthetic runs only)")
            torch.manual_seed(self._seed())
            cfg = EncodecConfig() if self.variant == "24k" else EncodecConfig(
                sampling_rate=48000, audio_channels=2, normalize=True, chunk_length_s=1.0, overlap=0.01,
                norm_type="time_group_norm", use_causal_conv=False)
            self.model = EncodecModel(cfg)
        self.channels = self.model.config.audio_channels
        self.segment_length = None if self.variant == "24k" else self.sr            # 48k: 1 s segments
        self.model.eval().to(self.device)

    def _get_frame(self, audio: torch.Tensor):
        with torch.no_grad():
            emb = self.model.encoder(audio.to(self.device))           # [1, 128, frames]
            return emb[0].transpose(0, 1)                             # [frames, 128]

    batch_files = 8

    def _get_embedding_batch(self, audios: list) -> list:
        # 24 kHz: clips of ONE length go through the encoder as one [B, 1, T] batch (the convolutions are per clip: the frames of a clip
        # do not depend on its neighbours); clips of different lengths -- and the segmented 48 kHz variant -- keep the per-file loop
        if self.segment_length is None and len(audios) > 1 and all(torch.is_tensor(a) and a.shape == audios[0].shape for a in audios):
            with torch.no_grad():
                emb = self.model.encoder(torch.cat(list(audios), dim=0).to(self.device))      # [B, 128, frames]
            return list(emb.transpose(1, 2).unbind(0))
        return [self._get_embedding(a) for a in audios]

    def _get_embedding(self, audio):
        if self.segment_length is None:
            return self._get_frame(audio)
        assert audio.dim() == 3
        frames = [self._get_frame(audio[:, :, o:o + self.segment_length])
                  for o in range(0, audio.shape[-1], self.segment_length)]
        return torch.cat(frames, dim=0)

    def load_wav(self, wav_file: Path):
        x, sr = read_audio(wav_file)                                  # [channels, samples]
        if sr != self.sr:
            x = np.stack([resample_kaiser(c, sr, self.sr) for c in x])
        want = getattr(self, "channels", 1 if self.variant == "24k" else 2)
        if want == 1:
            x = x.mean(axis=0, keepdims=True)
        elif x.shape[0] == 1:
            x = np.repeat(x, want, axis=0)
        wav = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
        limit = 3 * 60 * self.sr                                      # cut at 3 minutes (model_loader.py:172-174)
        return wav[:, :limit].unsqueeze(0)


# ---------------------------------------------------------------------------------------------
# LAION-CLAP (model_loader.py:291-418)
# ---------------------------------------------------------------------------------------------
class CLAPLaionModel(ModelLoader):
    """CLAP audio branch (HTSAT): 512-d embedding per 10 s window, 1 s hop."""

    def __init__(self, type: Literal["audio", "music"], random_init: Optional[bool] = None):
        super().__init__(f"clap-laion-{type}", 512, 48000)
        self.type = type
        self.random_init = random_init

    def load_model(self):
        from transformers import ClapAudioConfig, ClapAudioModelWithProjection
        hub_id = "laion/clap-htsat-unfused" if self.type == "audio" else "laion/larger_clap_music"
        try:
            self.model = ClapAudioModelWithProjection.from_pretrained(hub_id, local_files_only=True)
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            log.warning(f"{self.name}: no local weights, using seeded random weights (synthetic runs only)")
            torch.manual_seed(self._seed())
            cfg = ClapAudioConfig() if self.type == "audio" else ClapAudioConfig(patch_embeds_hidden_size=128,
                                                                                 depths=[2, 2, 12, 2], hidden_size=1024,
                                                                                 num_attention_heads=[4, 8, 16, 32])
            self.model = ClapAudioModelWithProjection(cfg)
        self.model.eval().to(self.device)

    @staticmethod
    def int16_to_float32(x):
        return (x / 32767.0).astype(np.float32)

    @staticmethod
    def float32_to_int16(x):
        return (np.clip(x, a_min=-1.0, a_max=1.0) * 32767.0).astype(np.int16)

    def _get_embedding(self, audio: np.ndarray):
        audio = np.asarray(audio).reshape(-1)
        audio = self.int16_to_float32(self.float32_to_int16(audio))          # quantisation round trip (:391-392)
        size, hop = 10 * self.sr, self.sr                                    # 10 s windows, 1 s hop (:395-398)
        chunks = []
        for i in range(0, audio.shape[0], hop):
            c = audio[i:i + size]
            chunks.append(c if c.shape[0] == size else np.pad(c, (0, size - c.shape[0])))
        if not chunks:
            return torch.zeros((0, 512))
        out = []
        with torch.no_grad():
            for s in range(0, len(chunks), 32):                              # all windows of a file, 32 per forward
                batch = chunks[s:s + 32]
                mel = hip.logmel_htsat([torch.from_numpy(c).to(self.device) if self.device.type == "cuda" else c
                                        for c in batch], device=self._device_index())        # [B, 1001, 64]
                mel = mel if torch.is_tensor(mel) else torch.from_numpy(mel).to(self.device)
                emb = self.model(input_features=mel.unsqueeze(1),
                                 is_longer=torch.zeros((len(batch), 1), dtype=torch.bool, device=self.device)).audio_embeds
                out.append(torch.nn.functional.normalize(emb, dim=-1))
        return torch.cat(out, dim=0)


# ---------------------------------------------------------------------------------------------
# MS-CLAP 2023 (model_loader.py:463-522)
# ---------------------------------------------------------------------------------------------
class _StandInAudioEncoder(nn.Module):
    """Seeded stand-in for msclap's audio encoder (synthetic runs only -- the real one lives in the `msclap` package):
    strided 1-D convolutions over the 7 s window, mean over time, projection to 1024.  Same call shape as
    `CLAP.clap.audio_encoder`: waveform [B, samples] -> (embedding [B, 1024], None)."""

    def __init__(self, out_dim: int = 1024):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(1, 32, 1024, stride=441), nn.GELU(), nn.Conv1d(32, 128, 8, stride=4), nn.GELU(),
                                 nn.Conv1d(128, 256, 4, stride=2), nn.GELU())
        self.proj = nn.Linear(256, out_dim)

    def forward(self, wav):
        return self.proj(self.net(wav.unsqueeze(1)).mean(dim=-1)), None


class CLAPModel(ModelLoader):
    """Microsoft CLAP (https://github.com/microsoft/CLAP), version 2023: 1024-d embedding per 7 s window, 1 s hop, 44.1 kHz.
    The audio encoder comes from the `msclap` package and its `CLAP_weights_2023.pth` (looked up locally under
    FADTK_AMD_CHECKPOINTS / fadtk's `.model-checkpoints`; nothing is downloaded); when either is missing the loader fails
    loudly unless random weights are allowed, in which case a seeded stand-in encoder keeps the plumbing testable."""

    def __init__(self, type: Literal["2023"] = "2023", random_init: Optional[bool] = None):
        super().__init__(f"clap-{type}", 1024, 44100)
        self.type = type
        self.random_init = random_init

    def load_model(self):
        ckpt = next((p for p in (_checkpoint_dir() / "CLAP_weights_2023.pth",
                                 Path(__file__).parent / ".model-checkpoints" / "CLAP_weights_2023.pth") if p.exists()), None)
        try:
            if ckpt is None:
                raise FileNotFoundError("CLAP_weights_2023.pth not found (set FADTK_AMD_CHECKPOINTS)")
            from msclap import CLAP
            self.model = CLAP(str(ckpt), version=self.type, use_cuda=self.device.type == "cuda")
            self.encoder = self.model.clap.audio_encoder
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            log.warning(f"{self.name}: msclap / its checkpoint not available, using a seeded stand-in encoder (synthetic runs only)")
            torch.manual_seed(self._seed())
            self.model = _StandInAudioEncoder(self.num_features)
            self.encoder = self.model
        self.encoder.eval().to(self.device)

    def _get_embedding(self, audio: np.ndarray):
        audio = np.asarray(audio, dtype=np.float32).reshape(-1)
        size, hop = 7 * self.sr, self.sr                                     # 7 s windows, 1 s hop (:498-500)
        out = []
        with torch.no_grad():
            for i in range(0, audio.shape[0], hop):                          # every window zero-padded to 7 s (:503-512)
                c = audio[i:i + size]
                c = c if c.shape[0] == size else np.pad(c, (0, size - c.shape[0]))
                out.append(self.encoder(torch.from_numpy(c)[None, :].float().to(self.device))[0])
        if not out:
            return torch.zeros((0, self.num_features))
        return torch.cat(out, dim=0)                                         # [windows, 1024]


# ---------------------------------------------------------------------------------------------
# Whisper (model_loader.py:636-672)
# ---------------------------------------------------------------------------------------------
_WHISPER_SHAPES = {  # size: (d_model, layers, heads)
    "tiny": (384, 4, 6), "base": (512, 6, 8), "small": (768, 12, 12), "medium": (1024, 24, 16), "large": (1280, 32, 20)}


class WhisperModel(ModelLoader):
    """Whisper: the DECODER hidden state for two start tokens -> [2, D] per clip, exactly what the
    reference stores under ``whisper-<size>`` (model_loader.py:662, 669-670)."""

    def __init__(self, size: Literal["tiny", "base", "small", "medium", "large"], random_init: Optional[bool] = None):
        super().__init__(f"whisper-{size}", _WHISPER_SHAPES[size][0], 16000)
        self.size = size
        self.huggingface_id = f"openai/whisper-{size}"
        self.random_init = random_init

    def load_model(self):
        from transformers import WhisperConfig
        from transformers import WhisperModel as HFWhisper
        try:
            self.model = HFWhisper.from_pretrained(self.huggingface_id, local_files_only=True)
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            log.warning(f"{self.name}: no local weights, using seeded random weights (synthetic runs only)")
            torch.manual_seed(self._seed())
            d, layers, heads = _WHISPER_SHAPES[self.size]
            self.model = HFWhisper(WhisperConfig(d_model=d, encoder_layers=layers, decoder_layers=layers,
                                                 encoder_attention_heads=heads, decoder_attention_heads=heads,
                                                 encoder_ffn_dim=4 * d, decoder_ffn_dim=4 * d))
        self.n_mels = int(self.model.config.num_mel_bins)
        self.decoder_input_ids = (torch.tensor([[1, 1]]) * self.model.config.decoder_start_token_id).to(self.device)
        self.model.eval().to(self.device)

    def _get_embedding(self, audio: np.ndarray):
        return self._get_embedding_batch([audio])[0]                                             # [2, D]

    batch_files = 16

    def _get_embedding_batch(self, audios: list) -> list:
        wavs = [torch.as_tensor(np.asarray(a, dtype=np.float32).reshape(-1)) for a in audios]
        if self.device.type == "cuda":
            wavs = [w.to(self.device) for w in wavs]
        feats = hip.logmel_whisper(wavs, n_mels=self.n_mels, device=self._device_index())      # [B, n_mels, 3000]: one launch for all clips
        feats = feats if torch.is_tensor(feats) else torch.from_numpy(feats).to(self.device)
        with torch.no_grad():
            out = self.model(feats, decoder_input_ids=self.decoder_input_ids.expand(len(wavs), -1)).last_hidden_state
        return list(out.unbind(0))                                                               # B x [2, D]


class _HFLayerModel(ModelLoader):
    """Shared body of the wav2vec2-family loaders (reference model_loader.py:254-288, 526-633): the waveform is
    truncated to ``limit_minutes``, normalised to zero mean / unit variance when the checkpoint's feature extractor
    says so (``do_normalize`` of its preprocessor_config.json; Wav2Vec2's default is True), pushed through the encoder with ``output_hidden_states`` and the
    hidden state of ``layer`` (0 = the projected CNN features) is the embedding, ``[frames, D]``.
    Weights: a local Hugging Face cache entry, else (opt-in) seeded random weights of the right shape."""

    _family = ""            # prefix of the model name
    _sizes: dict = {}       # size -> (hidden, layers, heads, hub id)
    _default_sr = 16000

    def __init__(self, size: str, layer: int, limit_minutes: float = 6, random_init: Optional[bool] = None):
        hidden, layers, _, hub = self._sizes[size]
        last = layer == layers
        super().__init__(f"{self._family}-{size}" + ("" if last else f"-{layer}"), hidden, self._default_sr)
        self.size, self.layer = size, int(layer)
        self.huggingface_id = hub
        self.limit = int(limit_minutes * 60 * self.sr)
        self.random_init = random_init

    def _hf_classes(self):
        raise NotImplementedError

    def load_model(self):
        cfg_cls, model_cls = self._hf_classes()
        try:
            self.model = model_cls.from_pretrained(self.huggingface_id, local_files_only=True)
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            log.warning(f"{self.name}: no local weights, using seeded random weights (synthetic runs only)")
            hidden, layers, heads, _ = self._sizes[self.size]
            torch.manual_seed(self._seed())
            self.model = model_cls(cfg_cls(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                                           intermediate_size=4 * hidden))
        self.do_normalize = self._checkpoint_do_normalize()
        self.model.eval().to(self.device)

    def _checkpoint_do_normalize(self) -> bool:
        """``do_normalize`` of the checkpoint's own feature extractor (the reference builds it with
        ``AutoFeatureExtractor.from_pretrained``): read from the locally cached preprocessor_config.json, True (the
        Wav2Vec2FeatureExtractor default) when there is none."""
        try:
            import json
            from transformers.utils import cached_file
            path = cached_file(self.huggingface_id, "preprocessor_config.json", local_files_only=True,
                               _raise_exceptions_for_missing_entries=False)
            if path:
                return bool(json.loads(Path(path).read_text()).get("do_normalize", True))
        except Exception:       # noqa: BLE001
            pass
        return True

    def _get_embedding(self, audio: np.ndarray):
        audio = np.asarray(audio, dtype=np.float32).reshape(-1)
        if audio.shape[0] > self.limit:
            log.warning(f"Audio is too long ({audio.shape[0] / self.sr / 60:.2f} minutes > "
                        f"{self.limit / self.sr / 60:.2f} minutes). Truncating.")
            audio = audio[:self.limit]
        x = torch.from_numpy(audio).to(self.device)
        if getattr(self, "do_normalize", True):
            x = (x - x.mean()) / torch.sqrt(x.var(unbiased=False) + 1e-7)
        with torch.no_grad():
            out = self.model(x[None, :], output_hidden_states=True)
        return out.hidden_states[self.layer].squeeze(0)          # [frames, D]


class W2V2Model(_HFLayerModel):
    """wav2vec 2.0 (facebook/wav2vec2-{base,large}-960h), reference model_loader.py:526-560."""
    _family = "w2v2"
    _sizes = {"base": (768, 12, 12, "facebook/wav2vec2-base-960h"), "large": (1024, 24, 16, "facebook/wav2vec2-large-960h")}

    def _hf_classes(self):
        from transformers import Wav2Vec2Config, Wav2Vec2Model
        return Wav2Vec2Config, Wav2Vec2Model


class HuBERTModel(_HFLayerModel):
    """HuBERT (facebook/hubert-{base,large}-ls960), reference model_loader.py:563-597."""
    _family = "hubert"
    _sizes = {"base": (768, 12, 12, "facebook/hubert-base-ls960"), "large": (1024, 24, 16, "facebook/hubert-large-ls960")}

    def _hf_classes(self):
        from transformers import HubertConfig, HubertModel
        return HubertConfig, HubertModel


class WavLMModel(_HFLayerModel):
    """WavLM (patrickvonplaten/wavlm-libri-clean-100h-{base,base-plus,large}), reference model_loader.py:600-633."""
    _family = "wavlm"
    _sizes = {"base": (768, 12, 12, "patrickvonplaten/wavlm-libri-clean-100h-base"),
              "base-plus": (768, 12, 12, "patrickvonplaten/wavlm-libri-clean-100h-base-plus"),
              "large": (1024, 24, 16, "patrickvonplaten/wavlm-libri-clean-100h-large")}

    def _hf_classes(self):
        from transformers import WavLMConfig
        from transformers import WavLMModel as HFWavLM
        return WavLMConfig, HFWavLM


class MERTModel(_HFLayerModel):
    """MERT-v1-95M (m-a-p/MERT-v1-95M; a HuBERT-shaped encoder on 24 kHz audio), reference model_loader.py:254-288.
    The checkpoint ships its own model class (``trust_remote_code``); only a locally cached copy can be loaded
    here.  The random-weight stand-in is a HuBERT encoder of the same width, depth and frame rate (75 Hz)."""
    _family = "MERT"
    _sizes = {"v1-95M": (768, 12, 12, "m-a-p/MERT-v1-95M")}
    _default_sr = 24000

    def __init__(self, size: str = "v1-95M", layer: int = 12, limit_minutes: float = 6, random_init: Optional[bool] = None):
        super().__init__(size, layer, limit_minutes, random_init)

    def load_model(self):
        try:
            from transformers import AutoConfig, AutoModel
            cfg = AutoConfig.from_pretrained(self.huggingface_id, trust_remote_code=True, local_files_only=True)
            cfg.conv_pos_batch_norm = False
            self.model = AutoModel.from_pretrained(self.huggingface_id, trust_remote_code=True, config=cfg, local_files_only=True)
            self.model.eval().to(self.device)
        except Exception:       # noqa: BLE001
            if not _allow_random(self.random_init):
                raise
            super().load_model()

    def _hf_classes(self):
        from transformers import HubertConfig, HubertModel
        return HubertConfig, HubertModel


# ---------------------------------------------------------------------------------------------
def get_all_models() -> List[ModelLoader]:
    """The loaders under fadtk's names, in the reference's order (model_loader.py:676-701).  Construction is
    cheap and offline (the reference's CLAP constructors download checkpoints; these never do).  Not offered: DAC and CDPAM
    (optional in the reference too: its registry skips them when their packages are missing, as they are here)."""
    return [
        CLAPModel("2023"), CLAPLaionModel("audio"), CLAPLaionModel("music"),
        VGGishModel(),
        *(MERTModel(layer=v) for v in range(1, 13)),
        EncodecEmbModel("24k"), EncodecEmbModel("48k"),
        *(W2V2Model("base", layer=v) for v in range(1, 13)),
        *(W2V2Model("large", layer=v) for v in range(1, 25)),
        *(HuBERTModel("base", layer=v) for v in range(1, 13)),
        *(HuBERTModel("large", layer=v) for v in range(1, 25)),
        *(WavLMModel("base", layer=v) for v in range(1, 13)),
        *(WavLMModel("base-plus", layer=v) for v in range(1, 13)),
        *(WavLMModel("large", layer=v) for v in range(1, 25)),
        WhisperModel("tiny"), WhisperModel("small"), WhisperModel("base"), WhisperModel("medium"), WhisperModel("large"),
    ]
