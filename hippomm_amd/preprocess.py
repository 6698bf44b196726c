"""Device-side vision preprocessing: decoded RGB uint8 frames -> (B,3,224,224) fp32 on the GPU,
bit-identical to the reference's host path (Pillow BICUBIC resize of the short side to 224, centre crop,
ToTensor, CLIP Normalize -- ``imagebind.data.load_and_transform_vision_data`` [upstream, recalled], called at
hippomm/models/foundation_models.py:87-90).

The host only computes Pillow's coefficient tables (a few hundred rows of taps per frame size, cached);
``hmm_preprocess_vision_u8`` does the two resample passes and the normalisation.
"""
from __future__ import annotations

import math
from functools import lru_cache
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib

OUT = 224
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x: float) -> float:
    a = -0.5                                     # Pillow's bicubic_filter
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def resample_coeffs(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray]:
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for BICUBIC, in the same double-precision order:
    -> (kk int32 [out_size][ksize], bounds int32 [out_size][2] = (first input index, tap count))."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return kk, bounds


def resized_shape(h: int, w: int) -> Tuple[int, int]:
    """torchvision Resize(224): short side -> 224, long side int(224 * long / short)."""
    if w <= h:
        return int(OUT * h / w), OUT
    return OUT, int(OUT * w / h)


@lru_cache(maxsize=64)
def _plan(h: int, w: int):
    nh, nw = resized_shape(h, w)
    left, top = int(round((nw - OUT) / 2.0)), int(round((nh - OUT) / 2.0))       # torchvision CenterCrop
    kh, bh = resample_coeffs(w, nw)
    kv, bv = resample_coeffs(h, nh)
    kh, bh = np.ascontiguousarray(kh[left:left + OUT]), np.ascontiguousarray(bh[left:left + OUT])
    kv, bv = np.ascontiguousarray(kv[top:top + OUT]), np.ascontiguousarray(bv[top:top + OUT])
    row_first = int(bv[:, 0].min())
    row_last = int((bv[:, 0] + bv[:, 1]).max())
    return kh, bh, kv, bv, row_first, row_last


_dev_plans = {}


def preprocess_frames_device(frames_u8: torch.Tensor) -> torch.Tensor:
    """frames_u8: (B,H,W,3) uint8 CUDA tensor of decoded RGB frames (all the same size, e.g. one video) ->
    (B,3,224,224) fp32 CUDA tensor, bit-identical to the Pillow/torchvision host pipeline."""
    lib = _lib.load()
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise ValueError(f"frames must be (B,H,W,3) uint8, got {tuple(frames_u8.shape)} {frames_u8.dtype}")
    frames_u8 = frames_u8.contiguous()
    B, H, W, _ = frames_u8.shape
    dev = frames_u8.device
    key = (H, W, dev)
    if key not in _dev_plans:
        kh, bh, kv, bv, r0, r1 = _plan(H, W)
        _dev_plans[key] = (tuple(torch.from_numpy(a).to(dev) for a in (kh, bh, kv, bv)), kh.shape[1], kv.shape[1], r0, r1)
    (kh_d, bh_d, kv_d, bv_d), ksh, ksv, r0, r1 = _dev_plans[key]
    out = torch.empty(B, 3, OUT, OUT, dtype=torch.float32, device=dev)
    need = lib.hmm_preprocess_vision_workspace_bytes(B, r1 - r0)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    _lib.check(lib.hmm_preprocess_vision_u8(frames_u8.data_ptr(), B, H, W, kh_d.data_ptr(), bh_d.data_ptr(), ksh,
                                            kv_d.data_ptr(), bv_d.data_ptr(), ksv, r0, r1, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "hmm_preprocess_vision_u8")
    return out


def load_and_transform_vision_data_device(image_paths: Sequence[str], device=None) -> torch.Tensor:
    """Decode on the host (PIL), resize / crop / normalise on the GPU.  Frames are grouped by size so that the
    frames of one video go through a single launch."""
    from PIL import Image
    dev = device or _lib.require_gpu()
    decoded: List[np.ndarray] = []
    for path in image_paths:
        with open(path, "rb") as fh:
            decoded.append(np.asarray(Image.open(fh).convert("RGB"), dtype=np.uint8))
    out = torch.empty(len(decoded), 3, OUT, OUT, dtype=torch.float32, device=dev)
    groups = {}
    for i, a in enumerate(decoded):
        groups.setdefault(a.shape[:2], []).append(i)
    for _, idxs in groups.items():
        batch = torch.from_numpy(np.stack([decoded[i] for i in idxs])).to(dev)
        out[torch.tensor(idxs, device=dev)] = preprocess_frames_device(batch)
    return out
