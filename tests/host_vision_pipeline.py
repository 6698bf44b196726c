"""Test witness (not product): the HOST image pipeline the reference goes through -- Pillow BICUBIC resize of the short
side to 224, centre crop, ToTensor, CLIP Normalize (foundation_models.py:87-90 -> imagebind.data.load_and_transform_vision_data
[upstream, recalled]).  The device pipeline (hippomm_amd/preprocess.py, preprocess.hip) must reproduce it bit for bit."""
from typing import List

import numpy as np
import torch

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def load_and_transform_vision_data(image_paths: List[str], device) -> torch.Tensor:
    """Resize(224, bicubic) -> CenterCrop(224) -> ToTensor -> Normalize(CLIP mean/std), as upstream
    ``imagebind.data.load_and_transform_vision_data`` [recalled]; PIL does the resampling."""
    from PIL import Image
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    batch = []
    for path in image_paths:
        with open(path, "rb") as fh:
            img = Image.open(fh).convert("RGB")
        w, h = img.size
        if w <= h:
            nw, nh = 224, int(224 * h / w)
        else:
            nw, nh = int(224 * w / h), 224
        img = img.resize((nw, nh), Image.BICUBIC)
        left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
        img = img.crop((left, top, left + 224, top + 224))
        arr = np.asarray(img, dtype=np.float32).transpose(2, 0, 1) / 255.0
        batch.append((arr - mean) / std)
    return torch.from_numpy(np.stack(batch)).to(device)
