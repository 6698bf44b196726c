"""Perceptual encoder on MI355X -- host-side mirror of ``hippomm.models.foundation_models.ImageBind``
(reference hippomm/models/foundation_models.py:21-151), bound to the HIP towers behind
``hmm_encoder_*`` (include/hippomm_hip.h).

Same surface as the reference class: ``ImageBind(model_path)`` with ``.device`` and ``.model``,
``load_data(inputs, modalities)``, ``forward(inputs)`` under ``no_grad`` and
``extract_features(inputs, modalities)``; modality keys are the plain strings the callers pass
('vision', 'audio'; hippocampal_memory.py:475, :1181, :1223).  Returned tensors are (B,1024) fp32
on the GPU and support ``.detach().cpu().numpy()`` (:480, :1186, :1335).

Differences, all at construction time: the reference ignores ``model_path`` and downloads
``imagebind_huge(pretrained=True)`` (:31-35); here weights come from ``state_dict=`` (a mapping with
the UPSTREAM key names) or from an ``imagebind_huge.pth`` found under ``model_path`` /
``.checkpoints``; nothing is downloaded and a missing checkpoint raises.  The 'text' tower
takes CLIP-BPE token ids (B,77) int64: the BPE vocabulary file is not shipped, so strings need a
``tokenizer=`` callable (str list -> (B,77) int64 tensor).
"""
from __future__ import annotations

import ctypes as C
import logging
import os
from pathlib import Path
from typing import Dict, Iterable, List, Mapping, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import _lib

logger = logging.getLogger(__name__)

TOWER_ID = {"vision": 0, "audio": 1, "text": 2}
FULL_DEPTH = {"vision": 32, "audio": 12, "text": 24}
INPUT_SHAPE = {"vision": (3, 224, 224), "audio": (3, 1, 128, 204), "text": (77,)}
INPUT_DTYPE = {"vision": torch.float32, "audio": torch.float32, "text": torch.int64}
_PREFIXES = {
    "vision": ("modality_preprocessors.vision.", "modality_trunks.vision.", "modality_heads.vision."),
    "audio": ("modality_preprocessors.audio.", "modality_trunks.audio.", "modality_heads.audio.",
              "modality_postprocessors.audio."),
    "text": ("modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text.",
             "modality_postprocessors.text."),
}
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


class ModalityType:
    """String constants, as ``imagebind.models.imagebind_model.ModalityType``."""
    VISION = "vision"
    AUDIO = "audio"
    TEXT = "text"


class HipTower:
    """One tower (vision or audio) resident on one GPU: packed bf16 weights + workspace."""

    def __init__(self, name: str, state_dict: Mapping[str, torch.Tensor], depth: Optional[int] = None,
                 device: Optional[torch.device] = None):
        self.name = name
        self.device = device or _lib.require_gpu()
        self._lib = _lib.load()
        self.depth = int(depth or FULL_DEPTH[name])
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.hmm_encoder_create(C.byref(handle), TOWER_ID[name], self.depth), "hmm_encoder_create")
            self._h = handle
            self._load(state_dict)
        self._ws = None

    def _load(self, state_dict):
        used = 0
        for key, value in state_dict.items():
            if not key.startswith(_PREFIXES[self.name]):
                continue
            if key.endswith(".mask"):                   # text preprocessor's causal-mask buffer: built into the kernel
                continue
            if ".blocks." in key:
                blk = int(key.split(".blocks.")[1].split(".")[0])
                if blk >= self.depth:
                    continue
            t = value.detach().to(device=self.device, dtype=torch.float32).contiguous()
            _lib.check(self._lib.hmm_encoder_load_param(self._h, key.encode(), t.data_ptr(), t.numel(),
                                                        _lib.stream_ptr()), f"hmm_encoder_load_param({key})")
            used += 1
        torch.cuda.current_stream().synchronize()     # sources may be freed after this returns
        missing = self._lib.hmm_encoder_missing_params(self._h)
        if missing != 0:
            raise _lib.HippoMMHipError(f"{self.name} tower: {self._lib.hmm_last_error().decode()}")
        self.n_params_loaded = used

    def close(self):
        if getattr(self, "_h", None):
            self._lib.hmm_encoder_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def flops(self, batch: int) -> float:
        """FLOPs of one forward as the reference computes it (the roofline figure)."""
        return float(self._lib.hmm_encoder_flops(self._h, batch))

    def flops_executed(self, batch: int) -> float:
        """FLOPs this build executes for the same forward (folded patch conv, cls-only last block)."""
        return float(self._lib.hmm_encoder_flops_executed(self._h, batch))

    def weight_bytes(self) -> int:
        """Bytes of the bf16 matrices one forward reads (blocks + head; embeddings, biases and norms left out): the
        weight-stream floor of a one-sample forward."""
        D, mlp = {"vision": (1280, 5120), "audio": (768, 3072), "text": (1024, 4096)}[self.name]
        return 2 * (self.depth * (4 * D * D + 2 * D * mlp) + 1024 * D)

    def set_fused_attention(self, on: bool):
        """Vision tower: in_proj + attention as one kernel (default) or as GEMM + attention kernel; same bits."""
        _lib.check(self._lib.hmm_encoder_set_fused_attention(self._h, int(bool(on))), "hmm_encoder_set_fused_attention")

    def set_streams(self, n: int):
        """2 (default): half-batches on two streams from 13 frames / 4 audio segments / 54 questions on; 1: a single chain."""
        _lib.check(self._lib.hmm_encoder_set_streams(self._h, int(n)), "hmm_encoder_set_streams")

    def _workspace(self, batch: int) -> torch.Tensor:
        need = self._lib.hmm_encoder_workspace_bytes(self._h, batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward_into(self, x: torch.Tensor, out: torch.Tensor):
        """x: (b, *INPUT_SHAPE) fp32 contiguous CUDA; out: (b,1024) fp32 CUDA.  Asynchronous."""
        b = x.shape[0]
        with torch.cuda.device(self.device):          # the handle's streams / events / weights live on self.device
            ws = self._workspace(b)
            _lib.check(self._lib.hmm_encoder_forward(self._h, x.data_ptr(), b, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     _lib.stream_ptr()), "hmm_encoder_forward")

    def __call__(self, x: torch.Tensor, max_batch: int = 256) -> torch.Tensor:
        if tuple(x.shape[1:]) != INPUT_SHAPE[self.name]:
            raise ValueError(f"{self.name} input must be (B,{','.join(map(str, INPUT_SHAPE[self.name]))}), "
                             f"got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=INPUT_DTYPE[self.name]).contiguous()
        out = torch.empty(x.shape[0], 1024, dtype=torch.float32, device=self.device)
        for s in range(0, x.shape[0], max_batch):
            self.forward_into(x[s:s + max_batch], out[s:s + max_batch])
        return out


class HipImageBindModel(nn.Module):
    """Stands where the reference keeps ``imagebind_huge`` in ``ImageBind.model``: a callable taking
    ``{modality: tensor}`` and returning ``{modality: (B,1024)}``."""

    def __init__(self, towers: Dict[str, HipTower], max_batch: Dict[str, int]):
        super().__init__()
        self.towers = towers
        self.max_batch = max_batch

    def forward(self, inputs: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        out = {}
        for key, value in inputs.items():
            if key not in self.towers:
                raise KeyError(f"modality {key!r} is not built on this device (have {list(self.towers)})")
            out[key] = self.towers[key](value, self.max_batch.get(key, 256))
        return out


def find_checkpoint(model_path: str) -> Optional[Path]:
    for cand in (Path(model_path), Path(model_path) / "imagebind_huge.pth", Path(".checkpoints") / "imagebind_huge.pth",
                 Path(os.environ.get("IMAGEBIND_CKPT", "/nonexistent"))):
        if cand.is_file():
            return cand
    return None


def synthetic_state_dict(towers: Iterable[str] = ("vision",), seed: int = 1234, depth: Optional[Dict[str, int]] = None,
                         device: Optional[torch.device] = None, std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Random-init weights of the imagebind_huge architecture under upstream key names, generated
    on the GPU (benchmarks; there is no checkpoint in this environment).  N(0, std) matrices and
    embeddings, zero biases, unit LayerNorm gains."""
    dev = device or _lib.require_gpu()
    g = torch.Generator(device=dev).manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def rn(*shape):
        return torch.randn(*shape, generator=g, device=dev) * std

    for m in towers:
        if m == "text":
            D, mlp = 1024, 4096
            pp, tr, hd = "modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text."
            sd[pp + "token_embedding.weight"] = rn(49408, D)
            sd[pp + "pos_embed"] = rn(1, 77, D)
            for i in range((depth or {}).get(m, FULL_DEPTH[m])):
                b = f"{tr}blocks.{i}."
                for ln in ("norm_1", "norm_2"):
                    sd[b + ln + ".weight"] = torch.ones(D, device=dev)
                    sd[b + ln + ".bias"] = torch.zeros(D, device=dev)
                sd[b + "attn.in_proj_weight"] = rn(3 * D, D)
                sd[b + "attn.in_proj_bias"] = torch.zeros(3 * D, device=dev)
                sd[b + "attn.out_proj.weight"] = rn(D, D)
                sd[b + "attn.out_proj.bias"] = torch.zeros(D, device=dev)
                sd[b + "mlp.fc1.weight"] = rn(mlp, D)
                sd[b + "mlp.fc1.bias"] = torch.zeros(mlp, device=dev)
                sd[b + "mlp.fc2.weight"] = rn(D, mlp)
                sd[b + "mlp.fc2.bias"] = torch.zeros(D, device=dev)
            sd[hd + "proj.0.weight"] = torch.ones(D, device=dev)
            sd[hd + "proj.0.bias"] = torch.zeros(D, device=dev)
            sd[hd + "proj.1.weight"] = rn(1024, D)
            sd["modality_postprocessors.text.1.log_logit_scale"] = torch.full((1,), float(np.log(1 / 0.07)), device=dev)
            continue
        D, mlp, T = (1280, 5120, 257) if m == "vision" else (768, 3072, 229)
        n_blk = (depth or {}).get(m, FULL_DEPTH[m])
        pp, tr, hd = (f"modality_preprocessors.{m}.", f"modality_trunks.{m}.", f"modality_heads.{m}.")
        sd[pp + "cls_token"] = rn(1, 1, D)
        sd[pp + "pos_embedding_helper.pos_embed"] = rn(1, T, D)
        if m == "vision":
            sd[pp + "rgbt_stem.proj.1.weight"] = rn(D, 3, 2, 14, 14)
            sd[tr + "pre_transformer_layer.0.weight"] = torch.ones(D, device=dev)
            sd[tr + "pre_transformer_layer.0.bias"] = torch.zeros(D, device=dev)
        else:
            sd[pp + "rgbt_stem.proj.weight"] = rn(D, 1, 16, 16)
            sd[pp + "rgbt_stem.norm_layer.weight"] = torch.ones(D, device=dev)
            sd[pp + "rgbt_stem.norm_layer.bias"] = torch.zeros(D, device=dev)
            sd["modality_postprocessors.audio.1.log_logit_scale"] = torch.full((1,), float(np.log(20.0)), device=dev)
        for i in range(n_blk):
            b = f"{tr}blocks.{i}."
            for ln in ("norm_1", "norm_2"):
                sd[b + ln + ".weight"] = torch.ones(D, device=dev)
                sd[b + ln + ".bias"] = torch.zeros(D, device=dev)
            sd[b + "attn.in_proj_weight"] = rn(3 * D, D)
            sd[b + "attn.in_proj_bias"] = torch.zeros(3 * D, device=dev)
            if m == "audio":
                sd[b + "attn.bias_k"] = rn(1, 1, D)
                sd[b + "attn.bias_v"] = rn(1, 1, D)
            sd[b + "attn.out_proj.weight"] = rn(D, D)
            sd[b + "attn.out_proj.bias"] = torch.zeros(D, device=dev)
            sd[b + "mlp.fc1.weight"] = rn(mlp, D)
            sd[b + "mlp.fc1.bias"] = torch.zeros(mlp, device=dev)
            sd[b + "mlp.fc2.weight"] = rn(D, mlp)
            sd[b + "mlp.fc2.bias"] = torch.zeros(D, device=dev)
        sd[hd + "0.weight"] = torch.ones(D, device=dev)
        sd[hd + "0.bias"] = torch.zeros(D, device=dev)
        sd[hd + "2.weight"] = rn(1024, D)
    return sd


class ImageBind(nn.Module):
    """ImageBind model for multimodal feature extraction (MI355X towers)."""

    def __init__(self, model_path: str = "pretrained/imagebind", *,
                 state_dict: Optional[Mapping[str, torch.Tensor]] = None,
                 towers: Iterable[str] = ("vision", "audio"),
                 depth: Optional[Dict[str, int]] = None,
                 max_batch: Optional[Dict[str, int]] = None,
                 tokenizer=None):
        super().__init__()
        self.tokenizer = tokenizer                  # str list -> (B,77) int64 (imagebind.data.load_and_transform_text)
        if self.tokenizer is None and "text" in tuple(towers):
            from .tokenizer import SimpleTokenizer, find_bpe_vocab     # upstream's merge table, if it is on disk
            bpe = find_bpe_vocab(model_path)
            if bpe is not None:
                self.tokenizer = SimpleTokenizer(str(bpe))
        self.device = _lib.require_gpu()            # reference :26 falls back to cpu then calls .cuda() anyway (:33)
        self.model = self._load_model(model_path, state_dict, tuple(towers), depth or {},
                                      max_batch or {"vision": 256, "audio": 128})

    def _load_model(self, model_path, state_dict, towers, depth, max_batch) -> nn.Module:
        """Build the HIP towers from upstream-named weights (reference :31-35)."""
        if state_dict is None:
            ckpt = find_checkpoint(model_path)
            if ckpt is None:
                raise FileNotFoundError(
                    f"no imagebind_huge.pth under {model_path!r} or .checkpoints/ and no state_dict= given; "
                    "hippomm_amd does not download weights")
            state_dict = torch.load(str(ckpt), map_location="cpu")
        built = {name: HipTower(name, state_dict, depth.get(name), self.device) for name in towers}
        model = HipImageBindModel(built, max_batch)
        model.eval()
        return model

    def _load_audio_file(self, audio_path: str) -> str:
        path = Path(audio_path)
        if not path.exists():
            logger.error(f"Failed to verify audio file {audio_path}: not found")
            raise FileNotFoundError(f"Audio file not found: {audio_path}")
        return str(path)

    def load_data(self, inputs: Dict[str, Union[List[str], List, np.ndarray, torch.Tensor]], modalities):
        """Raw inputs -> model tensors per modality (reference :48-114).  Errors are logged and
        the modality is skipped, as in the reference (:110-112).

        vision: list of image paths / PIL images (opened by path via ``.filename``, :83-86): decoded on the host,
        then resized (Pillow-identical bicubic, short side 224), centre-cropped and CLIP-normalised on the GPU
        (hippomm_amd/preprocess.py); or an already preprocessed (B,3,224,224) tensor.  audio: 16 kHz wav paths (three
        2-second clips -> kaldi log-mel filterbank on the GPU, hmm_audio_fbank) or a preprocessed (B,3,1,128,204) tensor."""
        transformed = {}
        for modality in modalities:
            if modality not in inputs:
                continue
            try:
                value = inputs[modality]
                if isinstance(value, torch.Tensor):
                    transformed[modality] = value.to(self.device)
                elif modality == ModalityType.VISION:
                    paths = [img if isinstance(img, str) else img.filename for img in value]
                    from .preprocess import load_and_transform_vision_data_device      # decode on host, rest on GPU
                    transformed[modality] = load_and_transform_vision_data_device(paths, self.device)
                elif modality == ModalityType.AUDIO:
                    if not all(isinstance(x, str) for x in value):
                        raise ValueError("Audio inputs must be file paths. Direct tensor/array inputs are not supported.")
                    paths = [self._load_audio_file(p) for p in value]
                    from .preprocess import load_and_transform_audio_data_device       # wav read on host, fbank on GPU
                    transformed[modality] = load_and_transform_audio_data_device(paths, self.device)
                elif modality == ModalityType.TEXT:
                    if self.tokenizer is None:
                        raise NotImplementedError("text needs the CLIP merge table bpe_simple_vocab_16e6.txt.gz (under "
                                                  "model_path, .checkpoints/, bpe/ or $IMAGEBIND_BPE), a tokenizer= "
                                                  "callable, or a (B,77) int64 token tensor")
                    transformed[modality] = self.tokenizer(list(value)).to(self.device)
                else:
                    raise NotImplementedError(f"modality {modality!r} is not built")
            except Exception as e:  # noqa: BLE001 - mirror of the reference's catch-all
                logger.error(f"Error processing {modality}: {str(e)}")
                continue
        return transformed

    def forward(self, inputs):
        """{modality: tensor} -> {modality: (B,1024) fp32} (reference :116-133)."""
        with torch.no_grad():
            embeddings = self.model(inputs)
        return embeddings

    def _vision_from_files(self, images) -> torch.Tensor:
        """Image paths -> (B,1024) embeddings as ONE pipeline: the host decodes while the GPU uploads, resizes and already
        embeds the frames that are ready (hippomm_amd.preprocess.vision_pipeline).  Same bits as forward(load_data(...)):
        every range the tower sees has at least two frames unless the call has one, and the tower's embedding of a frame
        does not depend on the batch it rides in within that regime (tests/test_gpu_encoder_batch.py)."""
        from .preprocess import vision_pipeline
        paths = [img if isinstance(img, str) else img.filename for img in images]
        tower = self.model.towers[ModalityType.VISION]
        step = int(self.model.max_batch.get(ModalityType.VISION, 256))
        emb = torch.empty(len(paths), 1024, dtype=torch.float32, device=self.device)

        def embed(x, lo, hi):
            for s in range(lo, hi, step):
                tower.forward_into(x[s:min(s + step, hi)], emb[s:min(s + step, hi)])

        vision_pipeline(paths, self.device, embed, max_chunk=step)
        return emb

    def extract_features(self, inputs, modalities):
        """load_data + forward (reference :135-151).  Vision given as files (the reference's formation calls,
        hippocampal_memory.py:1180-1186, :1328-1335) runs decode, upload, resize and the tower as one overlapped pipeline;
        a failure is logged and the modality left out, exactly as load_data does (:110-112)."""
        fused = {}
        rest = list(modalities)
        v = ModalityType.VISION
        if v in rest and v in inputs and not isinstance(inputs[v], torch.Tensor) and v in self.model.towers:
            rest.remove(v)
            try:
                with torch.no_grad():
                    fused[v] = self._vision_from_files(inputs[v])
            except Exception as e:  # noqa: BLE001 - mirror of the reference's catch-all
                logger.error(f"Error processing {v}: {str(e)}")
        out = self.forward(self.load_data(inputs, rest)) if rest else {}
        return {m: (fused[m] if m in fused else out[m]) for m in modalities if m in fused or m in out}
