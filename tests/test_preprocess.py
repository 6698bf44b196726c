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


def test_decoded_pixels_are_pillows_in_every_mode(tmp_path):
    """decode_rgb hands over Pillow's own pixels (the Arrow export of its RGBX block packed by hmm_host_arrow_rgbx_to_rgb, or the
    copying route): equal to np.asarray(Image.open(p).convert('RGB')) for JPEG, PNG, grey, palette and RGBA files."""
    from hippomm_amd.preprocess import decode_rgb
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    files = {"rgb.png": Image.fromarray(a), "rgb.jpg": Image.fromarray(a), "grey.png": Image.fromarray(a[..., 0]),
             "pal.png": Image.fromarray(a).convert("P"), "rgba.png": Image.fromarray(np.dstack([a, a[..., :1]]))}
    paths = []
    for name, im in files.items():
        im.save(tmp_path / name)
        paths.append(str(tmp_path / name))
    for got, p in zip(decode_rgb(paths, workers=3), paths):
        assert np.array_equal(got, np.asarray(Image.open(p).convert("RGB"))), p


def test_rgbx_packing_writes_three_bytes_per_pixel_and_nothing_else():
    from hippomm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(1)
    for n in (1, 2, 15, 16, 17, 31, 32, 33, 1000, 1280 * 7 + 5):
        src = rng.integers(0, 256, (n, 4), dtype=np.uint8)
        dst = np.full(3 * n + 32, 0xA5, dtype=np.uint8)
        assert lib.hmm_host_rgbx_to_rgb(src.ctypes.data, n, dst.ctypes.data) == 0
        assert np.array_equal(dst[:3 * n].reshape(n, 3), src[:, :3]) and (dst[3 * n:] == 0xA5).all(), n
    assert lib.hmm_host_rgbx_to_rgb(None, 4, None) == -1 and b"null pointer" in lib.hmm_last_error()
    assert lib.hmm_host_arrow_rgbx_to_rgb(None, 2, 2, 0, 0, 2, 2, None) == -1


def _write_pngs(tmp_path, sizes, seed=9):
    rng = np.random.default_rng(seed)
    paths = []
    for i, (h, w) in enumerate(sizes):
        p = tmp_path / f"g{i:03d}.png"
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(str(p))
    return paths


@pytest.mark.gpu
@pytest.mark.parametrize("workers,first_chunk,upload_min,depth,ring_bytes",
                         [(0, 0, 8, 1, None), (1, 4, 3, 1, None), (3, 2, 1, 2, None), (4, 3, 2, 1, 1), (6, 5, 4, 3, 1)])
def test_pipeline_ranges_ring_wrap_and_odd_sizes(tmp_path, monkeypatch, workers, first_chunk, upload_min, depth, ring_bytes):
    """vision_pipeline against the host chain, bit for bit: every cut of the call into ranges, a pinned ring much smaller than
    the call (slots reused while the call runs), frames of other sizes in the middle, and the ranges handed to the consumer
    cover [0, n) in order with no single-frame range."""
    from host_vision_pipeline import load_and_transform_vision_data
    if ring_bytes is not None:
        monkeypatch.setattr(pp, "STAGING_BYTES", ring_bytes)
        pp._staging.clear()
    sizes = [(96, 128)] * 23
    sizes[5], sizes[6], sizes[17] = (128, 96), (128, 96), (100, 100)
    paths = _write_pngs(tmp_path, sizes)
    seen, stats = [], {}

    def consume(x, lo, hi):
        seen.append((lo, hi, x[lo:hi].clone()))

    got = pp.vision_pipeline(paths, "cuda", consume, workers=workers, first_chunk=first_chunk, upload_min=upload_min, depth=depth,
                             stats=stats)
    want = load_and_transform_vision_data(paths, "cpu")
    assert torch.equal(got.cpu(), want)
    assert [lo for lo, _, _ in seen] == [0] + [hi for _, hi, _ in seen[:-1]] and seen[-1][1] == len(paths)
    assert all(hi - lo >= 2 for lo, hi, _ in seen) and sum(stats["chunks"]) == len(paths) and stats["odd_sized"] == 3
    for lo, hi, rows in seen:                                    # what the consumer saw on ITS stream is the final content
        assert torch.equal(rows.cpu(), want[lo:hi])
    if ring_bytes is not None:
        assert stats["ring_frames"] < len(paths)
    # a second call reuses the ring; one path only: a single range of one frame
    again = pp.vision_pipeline(paths[:1], "cuda", consume, workers=workers)
    assert torch.equal(again.cpu(), want[:1]) and seen[-1][:2] == (0, 1)
    pp._staging.clear()


@pytest.mark.gpu
def test_pipeline_reports_a_bad_file_and_stays_usable(tmp_path):
    paths = _write_pngs(tmp_path, [(64, 80)] * 12)
    (tmp_path / "broken.png").write_bytes(b"not an image")
    bad = paths[:7] + [str(tmp_path / "broken.png")] + paths[7:]
    with pytest.raises(Exception):
        pp.vision_pipeline(bad, "cuda", workers=4, first_chunk=2, upload_min=2)
    with pytest.raises(FileNotFoundError):
        pp.vision_pipeline([str(tmp_path / "missing.png")] + paths, "cuda")
    from host_vision_pipeline import load_and_transform_vision_data
    assert torch.equal(pp.vision_pipeline(paths, "cuda", workers=4, first_chunk=2).cpu(), load_and_transform_vision_data(paths, "cpu"))
    assert pp.vision_pipeline([], "cuda").shape == (0, 3, 224, 224)


def test_direct_jpeg_route_gives_pillows_pixels_and_declines_everything_else(tmp_path):
    """_decode_file: plain three-component JPEGs of the expected size go through Pillow's libjpeg decoder object driven directly
    (one reused image per thread); the pixels equal Image.open(...).convert('RGB') for baseline / progressive / optimised /
    4:4:4 / 4:2:2 files.  Grey, CMYK and PNG files, another size, and a truncated file take Pillow's ordinary route (same
    pixels, same errors)."""
    import io
    from hippomm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(4)
    a = np.asarray(Image.fromarray(rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)).resize((130, 90), Image.BICUBIC))
    variants = {"base.jpg": dict(quality=90), "prog.jpg": dict(quality=85, progressive=True), "opt444.jpg": dict(optimize=True, subsampling=0),
                "s422.jpg": dict(subsampling=1), "q30.jpg": dict(quality=30)}
    for name, kw in variants.items():
        Image.fromarray(a).save(tmp_path / name, **kw)
        data = (tmp_path / name).read_bytes()
        assert pp._jpeg_geometry(data) == (130, 90), name
        dst = np.zeros((90, 130, 3), np.uint8)
        assert pp._decode_file(str(tmp_path / name), 130, 90, dst, lib) is None
        assert np.array_equal(dst, np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))), name
    assert pp._direct["ok"] is True
    for name, im in {"grey.jpg": Image.fromarray(a).convert("L"), "cmyk.jpg": Image.fromarray(a).convert("CMYK"), "x.png": Image.fromarray(a)}.items():
        im.save(tmp_path / name)
        assert pp._jpeg_geometry((tmp_path / name).read_bytes()) is None, name
        dst = np.zeros((90, 130, 3), np.uint8)
        assert pp._decode_file(str(tmp_path / name), 130, 90, dst, lib) is None
        assert np.array_equal(dst, np.asarray(Image.open(tmp_path / name).convert("RGB"))), name
    other = pp._decode_file(str(tmp_path / "base.jpg"), 64, 64, np.zeros((64, 64, 3), np.uint8), lib)      # not the expected size
    assert other.shape == (90, 130, 3) and np.array_equal(other, np.asarray(Image.open(tmp_path / "base.jpg")))
    (tmp_path / "cut.jpg").write_bytes((tmp_path / "base.jpg").read_bytes()[:600])
    with pytest.raises(OSError):
        pp._decode_file(str(tmp_path / "cut.jpg"), 130, 90, np.zeros((90, 130, 3), np.uint8), lib)
    assert pp._direct["ok"] is True
    assert pp._jpeg_geometry(b"\xff\xd8\xff") is None and pp._jpeg_geometry(b"") is None
    # a window of the frame only (what the pipeline keeps): both routes, dense (h, w, 3) block
    win = (17, 5, 60, 40)
    for name in ("base.jpg", "grey.jpg"):
        dst = np.zeros((40, 60, 3), np.uint8)
        assert pp._decode_file(str(tmp_path / name), 130, 90, dst, lib, win) is None
        assert np.array_equal(dst, np.asarray(Image.open(tmp_path / name).convert("RGB"))[5:45, 17:77]), name
    im = Image.open(tmp_path / "base.jpg")
    im.load()
    import ctypes as C
    getp = C.pythonapi.PyCapsule_GetPointer
    getp.restype, getp.argtypes = C.c_void_p, [C.py_object, C.c_char_p]
    _, cap = im.__arrow_c_array__()
    assert lib.hmm_host_arrow_rgbx_to_rgb(getp(cap, b"arrow_array"), 130, 90, 100, 0, 40, 90, dst.ctypes.data) == -1   # window outside
    assert lib.hmm_host_arrow_rgbx_to_rgb(getp(cap, b"arrow_array"), 131, 90, 0, 0, 40, 40, dst.ctypes.data) == -1     # wrong image size


@pytest.mark.parametrize("h,w", SIZES + [(720, 1280), (1280, 720), (224, 224), (250, 230)])
def test_needed_window_holds_every_tap(h, w):
    """needed_window is exactly the bounding box of the taps of the cropped output rows / columns, inside the frame."""
    kh, bh, kv, bv, r0, r1 = pp._plan(h, w)
    x0, y0, ww, wh = pp.needed_window(h, w)
    assert 0 <= x0 and x0 + ww <= w and 0 <= y0 and y0 + wh <= h
    assert x0 == bh[:, 0].min() and x0 + ww == (bh[:, 0] + bh[:, 1]).max() and (y0, y0 + wh) == (r0, r1)
    if (h, w) == (720, 1280):
        assert (x0, ww, y0, wh) == (275, 730, 0, 720)


def test_cpu_quota_reads_the_cgroup_and_workers_follow_it(tmp_path, monkeypatch):
    """cpu_quota(): scheduler affinity cut by the container's CFS quota (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us`); decode_workers
    = that less two (the thread that feeds the GPU and the HIP runtime's own), HMM_DECODE_WORKERS overrides."""
    import builtins
    import os
    real_open = builtins.open
    files = {}

    def fake_open(path, *a, **kw):
        if isinstance(path, str) and path.startswith("/sys/fs/cgroup/"):
            if path in files:
                return real_open(files[path], *a, **kw)
            raise OSError(path)
        return real_open(path, *a, **kw)

    def put(path, text):
        f = tmp_path / path.strip("/").replace("/", "_")
        f.write_text(text)
        files[path] = str(f)

    monkeypatch.setattr(builtins, "open", fake_open)
    monkeypatch.setattr(os, "sched_getaffinity", lambda _pid: set(range(256)), raising=False)
    monkeypatch.delenv("HMM_DECODE_WORKERS", raising=False)
    assert pp.cpu_quota() == 256.0 and pp.decode_workers() == 16                  # no cgroup files: the affinity; threads capped at 16
    put("/sys/fs/cgroup/cpu.max", "1600000 100000\n")
    assert pp.cpu_quota() == 16.0 and pp.decode_workers() == 14                   # the pool's pods
    put("/sys/fs/cgroup/cpu.max", "max 100000\n")
    assert pp.cpu_quota() == 256.0
    files.clear()
    put("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "350000\n")
    put("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "100000\n")
    assert pp.cpu_quota() == 3.5 and pp.decode_workers() == 3                     # cgroup v1, a fractional quota
    put("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "-1\n")
    assert pp.cpu_quota() == 256.0
    monkeypatch.setenv("HMM_DECODE_WORKERS", "5")
    assert pp.decode_workers() == 5


def test_a_frame_pillow_keeps_in_several_blocks_takes_the_copying_route(tmp_path):
    """Beyond 16 MB of pixels Pillow has no single block to export: the direct decoder declines THAT frame (without switching itself
    off once it has been verified) and the ordinary route copies -- the same pixels."""
    from hippomm_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(6)
    small = np.asarray(Image.fromarray(rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)).resize((130, 90), Image.BICUBIC))
    Image.fromarray(small).save(tmp_path / "small.jpg")
    assert pp._decode_file(str(tmp_path / "small.jpg"), 130, 90, np.zeros((90, 130, 3), np.uint8), lib) is None and pp._direct["ok"] is True
    big = np.asarray(Image.fromarray(rng.integers(0, 256, (18, 26, 3), dtype=np.uint8)).resize((2600, 1800), Image.BICUBIC))
    Image.fromarray(big).save(tmp_path / "big.jpg", quality=80)
    dst = np.zeros((1800, 2600, 3), np.uint8)
    assert pp._decode_file(str(tmp_path / "big.jpg"), 2600, 1800, dst, lib) is None
    assert np.array_equal(dst, np.asarray(Image.open(tmp_path / "big.jpg"))) and pp._direct["ok"] is True
    win = (100, 50, 700, 600)
    dst = np.zeros((600, 700, 3), np.uint8)
    assert pp._decode_file(str(tmp_path / "big.jpg"), 2600, 1800, dst, lib, win) is None
    assert np.array_equal(dst, np.asarray(Image.open(tmp_path / "big.jpg"))[50:650, 100:800])


def test_decode_workers_share_the_quota_among_the_ranks_of_a_node(monkeypatch):
    monkeypatch.delenv("HMM_DECODE_WORKERS", raising=False)
    monkeypatch.setattr(pp, "cpu_quota", lambda: 16.0)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert pp.decode_workers() == 14
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")                  # torchrun --nproc-per-node 8 on a 16-CPU pod: two CPUs per rank
    assert pp.decode_workers() == 2
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "2")
    assert pp.decode_workers() == 6
