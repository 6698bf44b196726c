"""Vision preprocessing.  Oracle: Pillow itself -- the resampler upstream ImageBind relies on through
torchvision -- available in this image.  CPU: the host-computed coefficient tables reproduce Pillow's resize
bit for bit (numpy two-pass).  GPU: hmm_preprocess_vision_u8 equals the host PIL pipeline exactly."""
import numpy as np
import pytest
import torch
from PIL import Image

from hippomm_amd import preprocess as pp

SIZES = [(1080, 1920), (480, 640), (224, 224), (100, 180), (721, 333), (225, 300), (64, 64), (500, 224), (360, 202)]


def _pil_pipeline(img: np.ndarray) -> np.ndarray:
    """Resize(224, BICUBIC) -> CenterCrop(224) -> ToTensor -> Normalize(CLIP), as hippomm_amd.encoder's host path."""
    from hippomm_amd.encoder import CLIP_MEAN, CLIP_STD
    h, w = img.shape[:2]
    nh, nw = pp.resized_shape(h, w)
    im = Image.fromarray(img).resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
    im = im.crop((left, top, left + 224, top + 224))
    arr = np.asarray(im, dtype=np.float32).transpose(2, 0, 1) / 255.0
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None, None]
    return (arr - mean) / std


def _numpy_two_pass(img, kh, bh, kv, bv, r0, r1):
    P = pp.PRECISION_BITS
    tmp = np.zeros((r1 - r0, 224, 3), np.uint8)
    for xx in range(224):
        xmin, xmax = bh[xx]
        acc = (img[r0:r1, xmin:xmin + xmax].astype(np.int64) * kh[xx, :xmax, None].astype(np.int64)).sum(1) + (1 << (P - 1))
        tmp[:, xx] = np.clip(acc >> P, 0, 255)
    out = np.zeros((224, 224, 3), np.uint8)
    for yy in range(224):
        ymin, ymax = bv[yy]
        acc = (tmp[ymin - r0:ymin - r0 + ymax].astype(np.int64) * kv[yy, :ymax, None, None].astype(np.int64)).sum(0) + (1 << (P - 1))
        out[yy] = np.clip(acc >> P, 0, 255)
    return out


@pytest.mark.parametrize("h,w", SIZES)
def test_coefficient_tables_reproduce_pillow(h, w):
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    kh, bh, kv, bv, r0, r1 = pp._plan(h, w)
    got = _numpy_two_pass(img, kh, bh, kv, bv, r0, r1)
    nh, nw = pp.resized_shape(h, w)
    ref = Image.fromarray(img).resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - 224) / 2.0)), int(round((nh - 224) / 2.0))
    ref = np.asarray(ref.crop((left, top, left + 224, top + 224)))
    assert np.array_equal(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w", SIZES)
def test_device_preprocess_is_bit_identical_to_the_host_path(h, w):
    rng = np.random.default_rng(h + w)
    frames = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    frames[1] = (np.linspace(0, 255, w)[None, :, None] * np.ones((h, 1, 3))).astype(np.uint8)    # smooth ramp
    got = pp.preprocess_frames_device(torch.from_numpy(frames).cuda()).cpu().numpy()
    for i in range(3):
        want = _pil_pipeline(frames[i])
        assert np.array_equal(got[i], want), f"frame {i}: max diff {np.abs(got[i] - want).max()}"


@pytest.mark.gpu
def test_image_files_mixed_sizes(tmp_path):
    from host_vision_pipeline import load_and_transform_vision_data
    rng = np.random.default_rng(5)
    paths = []
    for i, (h, w) in enumerate([(300, 400), (300, 400), (256, 256), (400, 300)]):
        p = tmp_path / f"f{i}.png"                              # PNG: lossless, so decode is exact
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    dev = pp.load_and_transform_vision_data_device(paths).cpu()
    host = load_and_transform_vision_data(paths, "cpu")
    assert torch.equal(dev, host)


def test_threaded_decode_keeps_order_and_pixels(tmp_path):
    from PIL import Image
    from hippomm_amd.preprocess import decode_rgb
    rng = np.random.default_rng(0)
    paths = []
    for i in range(9):
        a = rng.integers(0, 256, (40 + i, 60, 3), dtype=np.uint8)
        p = tmp_path / f"f{i}.png"
        Image.fromarray(a).save(p)
        paths.append(str(p))
    seq = decode_rgb(paths, workers=1)
    par = decode_rgb(paths, workers=4)
    assert [a.shape for a in seq] == [(40 + i, 60, 3) for i in range(9)]
    assert all(np.array_equal(a, b) for a, b in zip(seq, par))
