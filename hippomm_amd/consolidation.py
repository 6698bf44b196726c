"""Consolidation similarity on MI355X -- host-side mirror of
``HippocampalMemory._select_key_frames`` (hippomm/core/hippocampal_memory.py:944-967),
bound to ``hmm_gram_select`` (fp32 normalise, fp64-accumulated gram via f64 MFMA, greedy scan).
"""
from __future__ import annotations

import threading
from typing import Optional, Union

import numpy as np
import torch

from . import _lib

FEATURE_DIM = 1024


# (device, stream, host thread, n) -> (kept int64[n], n_kept int32[1], workspace): select_key_frames_async allocates nothing
# after its first call.  The stream and the thread are part of the key: two selections of the same n on different streams or
# threads never share a workspace.  Only the async entry point uses the cache; the synchronous wrappers own their buffers.
_BUFFERS = {}
_BUFFERS_LOCK = threading.Lock()     # lookup / evict / insert as one step; an evicted entry stays alive with whoever holds it


def _new_buffers(lib, n, dev):
    need = lib.hmm_gram_select_workspace_bytes(n)
    return (torch.empty(max(n, 1), dtype=torch.int64, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
            torch.empty(max(need, 256), dtype=torch.uint8, device=dev))


def _prepare(features: torch.Tensor) -> torch.Tensor:
    if features.dim() != 2 or features.shape[1] != FEATURE_DIM:
        raise ValueError(f"features must be (n,{FEATURE_DIM}), got {tuple(features.shape)}")
    return features if features.dtype == torch.float32 and features.is_contiguous() else features.to(dtype=torch.float32).contiguous()


def _launch(lib, f, similarity_threshold, buf):
    kept, n_kept, ws = buf
    thr = float(np.float32(similarity_threshold))     # the reference compares in float32 (:960)
    _lib.check(lib.hmm_gram_select(f.data_ptr(), f.shape[0], FEATURE_DIM, thr, kept.data_ptr(), n_kept.data_ptr(),
                                   ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "hmm_gram_select")
    return kept, n_kept


def select_key_frames_async(features: torch.Tensor, similarity_threshold: float = 0.9):
    """features: (n,1024) fp32 CUDA tensor -> (kept int64[n], n_kept int32[1]) CUDA tensors, WITHOUT synchronising: the
    kept indices are ``kept[:n_kept]`` once the stream has run.  Launch-only (no allocation after the first call for a
    given n on this stream and thread, no read-back); the two tensors are reused by the next such call, so consume or copy
    them before that."""
    lib = _lib.load()
    f = _prepare(features)
    n, dev = f.shape[0], f.device
    key = (dev.index, int(_lib.stream_ptr().value or 0), threading.get_ident(), n)
    with _BUFFERS_LOCK:
        buf = _BUFFERS.get(key)
        if buf is None:
            if len(_BUFFERS) >= 16:                         # a handful of sizes recur (frame buffer, per-video totals)
                _BUFFERS.pop(next(iter(_BUFFERS)), None)    # the evicted tuple lives on in the caller that still uses it
            buf = _BUFFERS[key] = _new_buffers(lib, n, dev)
    return _launch(lib, f, similarity_threshold, buf)


def select_key_frames_device(features: torch.Tensor, similarity_threshold: float = 0.9) -> torch.Tensor:
    """features: (n,1024) fp32 CUDA tensor -> kept indices int64 CUDA tensor (synchronises once to read the count; buffers
    and result are this call's own)."""
    lib = _lib.load()
    f = _prepare(features)
    kept, n_kept = _launch(lib, f, similarity_threshold, _new_buffers(lib, f.shape[0], f.device))
    return kept[: int(n_kept.item())]


def select_key_frames(features: Union[np.ndarray, torch.Tensor], times: Optional[np.ndarray] = None,
                      similarity_threshold: float = 0.9) -> np.ndarray:
    """Same contract as the reference method: int64 indices, increasing, first is 0; ``times`` is
    accepted and unused, as in the reference.  Features are taken as float32, which is what the encoder hands over
    (hippocampal_memory.py:1186, :842).  A float64 matrix (what ``load_theta_event`` yields) holds exactly-float32 values
     and is converted without loss; the reference would then run its gram in float64, which can differ from the
    float32 comparison only for pairs within ~1e-7 of the threshold -- the same band in which its float32 answer already
    depends on the host's BLAS (tests/test_gpu_select.py pins that band to the fp64-accumulated definition)."""
    if len(features) <= 2:                              # :947-948
        return np.arange(len(features))
    dev = _lib.require_gpu()
    if isinstance(features, np.ndarray):
        f = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(dev)
    else:
        f = features.to(dev)
    return select_key_frames_device(f, similarity_threshold).cpu().numpy()


def _select_key_frames(self, features: np.ndarray, times: np.ndarray,
                       similarity_threshold: float = 0.9) -> np.ndarray:
    """Drop-in for ``HippocampalMemory._select_key_frames`` (assign it on the class; ``self`` is
    unused there as well)."""
    return select_key_frames(features, times, similarity_threshold)
