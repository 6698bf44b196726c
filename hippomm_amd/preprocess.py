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


def _device_plan(h: int, w: int, dev):
    """Pillow's coefficient tables for an h x w frame, uploaded once per (size, device)."""
    key = (h, w, str(dev))
    if key not in _dev_plans:
        kh, bh, kv, bv, r0, r1 = _plan(h, w)
        _dev_plans[key] = (tuple(torch.from_numpy(a).to(dev) for a in (kh, bh, kv, bv)), kh.shape[1], kv.shape[1], r0, r1)
    return _dev_plans[key]


@lru_cache(maxsize=64)
def needed_window(h: int, w: int) -> Tuple[int, int, int, int]:
    """(x0, y0, width, height) of the pixels of an h x w frame that the resampling taps of the 224 x 224 centre crop touch:
    the only part of a decoded frame that has to reach the GPU (a 1280x720 frame: 730 of its 1280 columns)."""
    _, bh, _, _, r0, r1 = _plan(h, w)
    c0, c1 = int(bh[:, 0].min()), int((bh[:, 0] + bh[:, 1]).max())
    return c0, r0, c1 - c0, r1 - r0


def _device_plan_window(h: int, w: int, dev):
    """_device_plan for frames uploaded as their needed_window only: tap positions counted from the window's corner."""
    key = (h, w, str(dev), "window")
    if key not in _dev_plans:
        kh, bh, kv, bv, r0, r1 = _plan(h, w)
        x0, y0, ww, wh = needed_window(h, w)
        bh, bv = bh.copy(), bv.copy()
        bh[:, 0] -= x0
        bv[:, 0] -= y0
        _dev_plans[key] = (tuple(torch.from_numpy(a).to(dev) for a in (kh, bh, kv, bv)), kh.shape[1], kv.shape[1], 0, wh)
    return _dev_plans[key]


def _preprocess_into(frames_u8: torch.Tensor, out: torch.Tensor, full_hw: Tuple[int, int] = None) -> torch.Tensor:
    """frames_u8 (B,H,W,3) uint8 contiguous on the GPU -> out (B,3,224,224) fp32 (rows of an existing result); current stream.
    full_hw: the frames are the needed_window cut-outs of full_hw = (height, width) frames."""
    lib = _lib.load()
    B, H, W, _ = frames_u8.shape
    dev = frames_u8.device
    (kh_d, bh_d, kv_d, bv_d), ksh, ksv, r0, r1 = _device_plan(H, W, dev) if full_hw is None else _device_plan_window(*full_hw, dev)
    ws = torch.empty(lib.hmm_preprocess_vision_workspace_bytes(B, r1 - r0), dtype=torch.uint8, device=dev)
    _lib.check(lib.hmm_preprocess_vision_u8(frames_u8.data_ptr(), B, H, W, kh_d.data_ptr(), bh_d.data_ptr(), ksh,
                                            kv_d.data_ptr(), bv_d.data_ptr(), ksv, r0, r1, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), _lib.stream_ptr()), "hmm_preprocess_vision_u8")
    return out


def preprocess_frames_device(frames_u8: torch.Tensor) -> torch.Tensor:
    """frames_u8: (B,H,W,3) uint8 CUDA tensor of decoded RGB frames (all the same size, e.g. one video) ->
    (B,3,224,224) fp32 CUDA tensor, bit-identical to the Pillow/torchvision host pipeline."""
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[3] != 3:
        raise ValueError(f"frames must be (B,H,W,3) uint8, got {tuple(frames_u8.shape)} {frames_u8.dtype}")
    frames_u8 = frames_u8.contiguous()
    out = torch.empty(frames_u8.shape[0], 3, OUT, OUT, dtype=torch.float32, device=frames_u8.device)
    return _preprocess_into(frames_u8, out)


# ---- host side of the vision path: files -> decoded RGB in pinned memory, overlapped with the GPU -------------------------------
# The reference decodes and resizes frame by frame on one thread (imagebind.data.load_and_transform_vision_data [upstream,
# recalled], foundation_models.py:87-90).  Here JPEG decoding is the only step left on the host (Pillow's libjpeg, so the pixels
# are the reference's); everything around it is arranged so that the decode threads hardly ever hold the interpreter lock:
#   * one persistent thread pool sized to the host (os.sched_getaffinity), not a fixed 8;
#   * a frame is decoded and then packed straight into its slot of a pinned ring (hmm_host_arrow_rgbx_to_rgb: Pillow's pixel
#     block through the Arrow C interface -- no tobytes / np.asarray / np.stack, no per-frame allocation);
#   * the main thread hands every run of decoded frames to the GPU as soon as it exists: H2D + resize on a side stream, the
#     consumer (the tower's forward) on the caller's stream behind an event, while the pool decodes the next frames.
STAGING_BYTES = 512 << 20                          # pinned ring per frame size (more only to hold workers + 2 chunks)
MAX_DECODE_THREADS = 16                            # see decode_workers
_capsule_pointer = None
_pools = {}
_staging = {}
_side_streams = {}
_pipeline_lock = __import__("threading").Lock()     # one call at a time owns the ring, the side stream and the decode pool


def cpu_quota() -> float:
    """CPUs this process may really use: the scheduler affinity, cut by the container's CFS quota (cgroup v2 cpu.max / v1
    cpu.cfs_quota_us) when there is one.  A pod on a 256-thread host with `cpu.max = 1600000 100000` has 16 CPUs: more busy
    threads than that are not faster, they are throttled -- every thread of the process, the one feeding the GPU included."""
    import os
    try:
        n = float(len(os.sched_getaffinity(0)))
    except AttributeError:
        n = float(os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, float(quota) / float(period))
    except (OSError, ValueError):
        try:
            quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0 and period > 0:
                n = min(n, quota / period)
        except (OSError, ValueError):
            pass
    return max(1.0, n)


def decode_workers() -> int:
    """Decode threads: the CPUs this process may use (cpu_quota) -- its share of them when torchrun started several ranks on the node
    (LOCAL_WORLD_SIZE) -- less two for the thread that feeds the GPU and the HIP runtime's own, and at most MAX_DECODE_THREADS;
    HMM_DECODE_WORKERS overrides."""
    import os
    env = os.environ.get("HMM_DECODE_WORKERS")
    if env:
        return max(1, int(env))
    try:
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    q = int(cpu_quota() / ranks)
    # never more than 16: the threads share one interpreter lock, and past ~16 of them the hand-overs cost more than the cores give
    # (256 frames: 105 ms on 16 threads, 122-139 ms on 32-128, profiles/r6_formation_probe_first.json; one burst of 32 frames: 6.1 ms
    # on 16, 5.7 on 32, profiles/r6_decode_burst.json)
    return max(1, min(q - 2 if q > 4 else q, MAX_DECODE_THREADS))


def _decode_pool(workers: int):
    """Persistent pool (threads are started as tasks arrive, up to `workers`).  The threads only ever run Pillow and the
    packing call; a fork()ed child (the reference forks multiprocessing pools nearby) starts without them and must not decode."""
    if workers not in _pools:
        from concurrent.futures import ThreadPoolExecutor
        _pools[workers] = ThreadPoolExecutor(max_workers=workers, thread_name_prefix="hmm-decode")
    return _pools[workers]


def _open_rgb(fh):
    """File object -> loaded RGB Pillow image.  The file is read in one piece and handed to the decoder in one call
    (decodermaxblock; Pillow's default feeds it 64 KB at a time from a Python loop): the same pixels, fewer hand-overs of
    the interpreter lock per frame."""
    return _open_rgb_bytes(fh.read())


def _open_rgb_bytes(data: bytes):
    import io
    from PIL import Image
    im = Image.open(io.BytesIO(data))
    im.decodermaxblock = max(len(data), 1 << 16)
    im.load()
    return im if im.mode == "RGB" else im.convert("RGB")


def _jpeg_geometry(data: bytes):
    """(width, height) of a plain 8-bit three-component Huffman JPEG (baseline, extended or progressive; what a video frame
    extractor writes), None for anything else -- no Adobe marker, no CMYK / grey / lossless / arithmetic coding: those keep
    Pillow's own opening logic."""
    if len(data) < 4 or data[0] != 0xFF or data[1] != 0xD8:
        return None
    i, end = 2, len(data) - 9
    while i < end:
        if data[i] != 0xFF:
            return None
        m = data[i + 1]
        if m in (0xC0, 0xC1, 0xC2):
            if data[i + 4] != 8 or data[i + 9] != 3:
                return None
            return (data[i + 7] << 8) | data[i + 8], (data[i + 5] << 8) | data[i + 6]
        if m == 0xEE or m == 0xDA or 0xC3 <= m <= 0xCF and m not in (0xC4, 0xC8, 0xCC):
            return None
        if m == 0xFF:                                              # fill byte
            i += 1
            continue
        if 0xD0 <= m <= 0xD9 or m == 0x01:
            i += 2
            continue
        i += 2 + ((data[i + 2] << 8) | data[i + 3])
    return None


_direct = {"ok": None, "fails": 0}                                 # ok None: not verified yet; False: this Pillow does not play along
_thread_images = None


def _decode_jpeg_direct(data: bytes, w: int, h: int, dst: np.ndarray, lib, window=None) -> bool:
    """A w x h plain JPEG -> dst through Pillow's libjpeg decoder object driven directly: Image.open's marker parsing, the
    per-frame 4-bytes-per-pixel image allocation (3.7 MB of fresh pages per 720p frame) and its release are what the decode
    threads spent their time under the interpreter lock on; here a thread keeps ONE image core of the call's size and decodes
    every frame into it.  The decoder is the one Image.open(...).load() would build (same mode, raw mode and defaults), so the
    pixels are Pillow's; the first use in a process is checked against Image.open on the same bytes, and any surprise
    (another Pillow, an exotic file) turns this path off or hands the frame to the ordinary route.  -> False: not done."""
    global _thread_images
    if _direct["ok"] is False:
        return False
    import threading
    from PIL import Image
    try:
        if _thread_images is None:
            _thread_images = threading.local()
        core = getattr(_thread_images, "core", None)
        if core is None or getattr(_thread_images, "size", None) != (w, h):
            core = Image.core.new("RGB", (w, h))
            _thread_images.core, _thread_images.size = core, (w, h)
        decoder = Image._getdecoder("RGB", "jpeg", ("RGB", ""), ())
        try:
            decoder.setimage(core, (0, 0, w, h))
            n, err = decoder.decode(data)
        finally:
            decoder.cleanup()
        if n >= 0 or err < 0:
            return False                                           # truncated / broken: let Pillow's route raise its error
        global _capsule_pointer
        if _capsule_pointer is None:
            import ctypes as C
            fn = C.pythonapi.PyCapsule_GetPointer
            fn.restype, fn.argtypes = C.c_void_p, [C.py_object, C.c_char_p]
            _capsule_pointer = fn
        x0, y0, ww, wh = window if window is not None else (0, 0, w, h)
        capsule = core.__arrow_c_array__()                         # must outlive the call: dropping it releases the export
        if lib.hmm_host_arrow_rgbx_to_rgb(_capsule_pointer(capsule, b"arrow_array"), w, h, x0, y0, ww, wh, dst.ctypes.data) != 0:
            raise RuntimeError(lib.hmm_last_error().decode())
        del capsule
        if _direct["ok"] is None:                                  # first frame of the process: Pillow's own route must agree
            _direct["ok"] = bool(np.array_equal(dst, np.asarray(_open_rgb_bytes(data))[y0:y0 + wh, x0:x0 + ww]))
            return _direct["ok"]
        return True
    except Exception:                                              # noqa: BLE001 - this frame takes the ordinary route ...
        _direct["fails"] += 1                                      # (an image Pillow keeps in several blocks -- beyond 16 MB -- has no export)
        if _direct["ok"] is not True and _direct["fails"] >= 3:    # ... and if it never worked, so does every later one
            _direct["ok"] = False
        return False


def _decode_file(path: str, w: int, h: int, dst: np.ndarray, lib, window=None):
    """One image file -> dst (its `window` (x0, y0, w, h) when given) when it is w x h (returns None), else -> its own (H,W,3) array."""
    with open(path, "rb") as fh:
        data = fh.read()
    if _jpeg_geometry(data) == (w, h) and _decode_jpeg_direct(data, w, h, dst, lib, window):
        return None
    im = _open_rgb_bytes(data)
    if im.size == (w, h):
        _pack_into(im, dst, lib, window)
        return None
    arr = np.empty((im.size[1], im.size[0], 3), dtype=np.uint8)
    _pack_into(im, arr, lib)
    return arr


def _pack_into(im, dst: np.ndarray, lib, window=None) -> None:
    """Pixels of a loaded RGB Pillow image (or its window (x0, y0, w, h)) -> dst (h,w,3) uint8; without the interpreter lock
    when Pillow exports its block."""
    global _capsule_pointer
    if _capsule_pointer is None:
        import ctypes as C
        fn = C.pythonapi.PyCapsule_GetPointer
        fn.restype, fn.argtypes = C.c_void_p, [C.py_object, C.c_char_p]
        _capsule_pointer = fn
    x0, y0, ww, wh = window if window is not None else (0, 0, im.size[0], im.size[1])
    export = getattr(im, "__arrow_c_array__", None)               # Pillow >= 11.2
    if export is not None:
        try:
            _, array = export()                                   # refused for an image kept in several blocks (> 16 MB)
        except Exception:                                         # noqa: BLE001 - any refusal: the copying route below
            array = None
        if array is not None and lib.hmm_host_arrow_rgbx_to_rgb(_capsule_pointer(array, b"arrow_array"), im.size[0], im.size[1],
                                                                x0, y0, ww, wh, dst.ctypes.data) == 0:
            return
    np.copyto(dst, np.asarray(im, dtype=np.uint8)[y0:y0 + wh, x0:x0 + ww])


def decode_rgb(image_paths: Sequence[str], workers: int = 0) -> List[np.ndarray]:
    """Paths -> uint8 (H,W,3) arrays in the given order (Pillow's decoders on a thread pool over the host's cores).
    workers = 0: decode_workers(); 1: sequential."""
    lib = _lib.load()

    def one(path):
        with open(path, "rb") as fh:
            im = _open_rgb(fh)
        out = np.empty((im.size[1], im.size[0], 3), dtype=np.uint8)
        _pack_into(im, out, lib)
        return out

    if workers <= 0:
        workers = decode_workers()
    if workers == 1 or len(image_paths) < 2:
        return [one(p) for p in image_paths]
    return list(_decode_pool(workers).map(one, image_paths))


class _Staging:
    """Pinned ring of `capacity` frame slots of one frame size (one per device and size, reused by every call)."""

    def __init__(self, h: int, w: int, capacity: int):
        self.capacity = capacity
        self.pinned = torch.empty(capacity, h, w, 3, dtype=torch.uint8, pin_memory=True)       # h x w: the needed_window
        self.host = self.pinned.numpy()
        self.last_upload = None                    # event behind the last H2D issued from this ring


def _get_staging(h: int, w: int, need: int, dev) -> _Staging:
    key = (h, w, str(dev))
    st = _staging.get(key)
    if st is None or st.capacity < need:
        if st is not None and st.last_upload is not None:
            st.last_upload.synchronize()
        others = [k for k in _staging if k != key]
        for k in others[: max(0, len(others) - 2)]:                     # at most three frame sizes stay pinned
            if _staging[k].last_upload is not None:
                _staging[k].last_upload.synchronize()
            del _staging[k]
        st = _staging[key] = _Staging(h, w, need)
    return st


def _side_stream(dev):
    key = str(dev)
    if key not in _side_streams:
        # high priority: a resize launch must not queue behind the tower's kernels of the previous range, it is what the next one waits for
        _side_streams[key] = torch.cuda.Stream(device=dev, priority=-1)
    return _side_streams[key]


def vision_pipeline(image_paths: Sequence[str], device=None, consume=None, workers: int = 0, first_chunk: int = 0,
                    upload_min: int = 8, depth: int = 1, max_chunk: int = 256, stats: dict = None,
                    poll_s: float = 0.0003, tail_wait: int = 0) -> torch.Tensor:
    """Image files -> (B,3,224,224) fp32 on `device`, bit-identical to the Pillow / torchvision host chain, as a pipeline:
    decode (host threads) | H2D + resize / crop / normalise (side stream) | consume(x, lo, hi) (caller's stream).

    Decoded frames are uploaded and resized eagerly, `upload_min` at a time.  `consume`, when given, is called on the caller's
    stream for consecutive row ranges [lo, hi) of the result (the tower's forward: ImageBind.extract_features): the first
    range as soon as `first_chunk` frames are there (0: eight, or one round of the decode threads if that is fewer -- the tower starts
    ~2 ms into the call; 32 paths 20.4 -> 19.4-19.9 ms), every later one when fewer than
    `depth` earlier ranges are still running on the GPU, and it takes everything uploaded by then -- a forward of few frames is
    latency-bound (~2.7 ms + 0.29 ms per frame on the ViT-H tower), so ranges queued ahead of need only add their fixed part
    (profiles/r6_formation_probe.json); once everything is uploaded the rest is queued at once.  A range is at least two
    frames unless the call has a single path, so a tower stays in one arithmetic regime whatever the
    timing: the embedding bits do not depend on how the call happened to be cut.  Frames whose size differs from the first
    file's take the same route one by one.  Returns when everything has been ISSUED; the result is ordered on the caller's
    stream like any kernel output."""
    import threading
    import time
    from collections import deque
    from PIL import Image

    dev = torch.device(device) if device is not None else _lib.require_gpu()
    lib = _lib.load()
    paths = [str(p) for p in image_paths]
    n = len(paths)
    x = torch.empty(n, 3, OUT, OUT, dtype=torch.float32, device=dev)
    if n == 0:
        return x
    workers = min(workers if workers > 0 else decode_workers(), n)
    first_chunk = max(2, min(first_chunk if first_chunk > 0 else min(workers, 8), n)) if n > 1 else 1
    upload_min = max(1, upload_min)
    max_chunk = max(max_chunk, 3)
    with _pipeline_lock, torch.cuda.device(dev):
        with open(paths[0], "rb") as fh:
            W, H = Image.open(fh).size                                    # header only
        window = needed_window(H, W)                                      # only these pixels are kept, uploaded and read
        WW, WH = window[2], window[3]
        cap = min(n, max(STAGING_BYTES // (WH * WW * 3), workers + 2 * max(first_chunk, upload_min)))
        st = _get_staging(WH, WW, cap, dev)
        cap = st.capacity
        if st.last_upload is not None:
            st.last_upload.synchronize()                                  # the previous call's uploads have left the ring
        side, cur = _side_stream(dev), torch.cuda.current_stream(dev)
        side.wait_stream(cur)                                             # x (and whatever memory it reuses) is ours from here

        cond = threading.Condition()
        done = [False] * n
        odd = {}                                                          # frame -> its own array (a size other than H x W)
        slot_free = [None] * n                                            # frame -> event behind the H2D that emptied its slot
        state = {"prefix": 0, "uploaded": 0, "error": None, "abort": False, "odd": 0}

        def work(i):
            try:
                if i >= cap:                                              # the slot still belongs to frame i - cap
                    with cond:
                        while slot_free[i - cap] is None and not state["abort"]:
                            cond.wait()
                        ev = slot_free[i - cap]
                    if state["abort"]:
                        return
                    if ev is not True:
                        ev.synchronize()
                if state["abort"]:
                    return
                arr = _decode_file(paths[i], W, H, st.host[i % cap], lib, window)
                with cond:
                    if arr is not None:
                        odd[i] = arr
                        state["odd"] += 1
                    done[i] = True
                    old = p = state["prefix"]
                    while p < n and done[p]:
                        p += 1
                    if p != old:
                        state["prefix"] = p
                        if p == n or p - state["uploaded"] >= upload_min or old < first_chunk <= p:
                            cond.notify_all()                             # the main thread has something to do
            except BaseException as exc:                                  # noqa: BLE001 - handed to the caller below
                with cond:
                    state["error"] = state["error"] or exc
                    state["abort"] = True
                    cond.notify_all()

        def upload(a, hi):
            """H2D + resize of the decoded frames [a, hi) on the side stream -> event behind their resize."""
            with torch.cuda.stream(side):
                while a < hi:
                    if a in odd:
                        _preprocess_into(torch.from_numpy(odd.pop(a)).unsqueeze(0).to(dev), x[a:a + 1])
                        ev, b = True, a + 1
                    else:
                        b = a + 1
                        while b < hi and b not in odd and b % cap != 0:
                            b += 1
                        d = torch.empty(b - a, WH, WW, 3, dtype=torch.uint8, device=dev)
                        d.copy_(st.pinned[a % cap:a % cap + (b - a)], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(side)
                        st.last_upload = ev
                        _preprocess_into(d, x[a:b], (H, W))
                        del d
                    with cond:
                        for i in range(a, b):
                            slot_free[i] = ev
                        cond.notify_all()
                    a = b
                ready = torch.cuda.Event(enable_timing=trace is not None)
                ready.record(side)
            return ready

        futures = None
        if workers > 1:
            pool = _decode_pool(workers)
            futures = [pool.submit(work, i) for i in range(n)]
        uploaded, issued, sequential_next = 0, 0, 0
        batches = deque()                                                 # (end frame, event behind its resize), in order
        running = deque()                                                 # events behind the ranges handed to `consume`
        chunks = []
        trace = [] if stats is not None and stats.get("trace") else None  # per range: host ms since the call began
        t_begin = time.perf_counter()
        try:
            while issued < n if consume is not None else uploaded < n:
                if futures is None and sequential_next < n:               # workers = 1: decode here, a few frames at a time
                    for _ in range(min(upload_min if uploaded else first_chunk, n - sequential_next)):
                        work(sequential_next)
                        sequential_next += 1
                with cond:
                    if state["error"] is not None:
                        raise state["error"]
                    prefix = state["prefix"]
                progress = False
                if prefix > uploaded and (prefix - uploaded >= upload_min or prefix == n
                                          or (uploaded < first_chunk <= prefix)):
                    batches.append((prefix, upload(uploaded, prefix)))
                    uploaded = state["uploaded"] = prefix
                    progress = True
                if consume is not None and uploaded > issued:
                    while running and running[0].query():
                        running.popleft()
                    need = first_chunk if issued == 0 else 2
                    hi = 0
                    if uploaded == n:
                        hi = min(n, issued + max_chunk)                   # everything is on its way: queue the rest behind the running range
                    elif n - uploaded <= tail_wait and issued > 0:
                        pass                                              # the last few frames are about to arrive: one range, not two
                    elif len(running) < depth and uploaded - issued >= need:
                        # the GPU has room: hand it the frames whose resize has FINISHED (a range waits for its last upload; while the
                        # tower is busy that wait is free, on an idle GPU it is not)
                        ready_upto = issued
                        for end, ev in batches:
                            if end <= issued:
                                continue
                            if not ev.query():
                                break
                            ready_upto = end
                        if ready_upto - issued >= need:
                            hi = min(ready_upto, issued + max_chunk)
                        elif not running:
                            hi = min(uploaded, issued + max_chunk)        # an idle GPU and nothing finished: it waits either way
                    if hi:
                        if n - hi == 1:                                   # never leave a single frame for the last range
                            hi = hi - 1 if hi - issued >= 3 else 0
                        if hi:
                            while batches[0][0] < hi:
                                batches.popleft()
                            cur.wait_event(batches[0][1])
                            consume(x, issued, hi)
                            fin = torch.cuda.Event(enable_timing=trace is not None)
                            fin.record(cur)
                            running.append(fin)
                            chunks.append(hi - issued)
                            if trace is not None:
                                trace.append({"frames": hi - issued, "issued_at": round((time.perf_counter() - t_begin) * 1e3, 2),
                                              "decoded_then": state["prefix"], "_ready": batches[0][1], "_fin": fin})
                            issued = hi
                            progress = True
                if not progress and (futures is not None or sequential_next >= n):
                    with cond:                                            # decoders or the GPU have to move first
                        if state["prefix"] == prefix and state["error"] is None:
                            cond.wait(poll_s if running else 0.002)
            if consume is None:
                cur.wait_event(batches[-1][1])
        finally:
            if (issued < n if consume is not None else uploaded < n):     # an error: nobody may still be writing into the ring
                with cond:
                    state["abort"] = True
                    cond.notify_all()
            if futures is not None:
                for f in futures:
                    f.cancel()
                for f in futures:
                    if not f.cancelled():
                        f.result()                                        # work() never raises: errors are in state["error"]
        if stats is not None:
            stats.update(chunks=chunks, workers=workers, ring_frames=cap, frame_hw=(H, W), window=window, odd_sized=state["odd"],
                         uploads=len(batches) if consume is None else None)
            if trace is not None and trace:                               # diagnostic: synchronises
                torch.cuda.synchronize(dev)
                first = trace[0]["_ready"]
                for r in trace:
                    ready, fin = r.pop("_ready"), r.pop("_fin")
                    r["gpu_ready_at"] = round(first.elapsed_time(ready), 2)          # ms after the first range was resized
                    r["gpu_done_at"] = round(first.elapsed_time(fin), 2)
                stats["trace"] = trace
    return x


def load_and_transform_vision_data_device(image_paths: Sequence[str], device=None, workers: int = 0) -> torch.Tensor:
    """Drop-in for imagebind.data.load_and_transform_vision_data(image_paths, device): decode on the host (Pillow, every core),
    resize / crop / normalise on the GPU, the two overlapped (vision_pipeline)."""
    return vision_pipeline(image_paths, device, None, workers)


# =====================================================================================================
# Audio (audio half): wav samples -> (B,3,1,128,204) normalised log-mel clips on the GPU.
# Mirrors imagebind.data.load_and_transform_audio_data [upstream, recalled] (called at
# hippomm/models/foundation_models.py:106-109): three 2-second clips spread evenly over the file
# (pytorchvideo ConstantClipsPerVideoSampler), per clip `waveform -= waveform.mean()`, kaldi fbank (128 mel bins,
# 25 ms / 10 ms, Hann, 16 kHz), zero-pad to 204 frames, Normalize(-4.268, 9.138).  The host reads the wav and slices
# the clips; hmm_audio_fbank does everything else.
# =====================================================================================================
AUDIO_SAMPLE_RATE = 16000
AUDIO_CLIP_DURATION = 2
AUDIO_CLIPS_PER_VIDEO = 3
AUDIO_MEAN, AUDIO_STD = -4.268, 9.138
AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH = 128, 204


@lru_cache(maxsize=512)
def _audio_clip_bounds(n_samples: int, sample_rate: int) -> Tuple[Tuple[int, int], ...]:
    from fractions import Fraction
    duration = n_samples / sample_rate
    max_start = Fraction(max(duration - AUDIO_CLIP_DURATION, 0))
    step = Fraction(max_start, max(AUDIO_CLIPS_PER_VIDEO - 1, 1))
    out = []
    for i in range(AUDIO_CLIPS_PER_VIDEO):
        start, end = step * i, step * i + AUDIO_CLIP_DURATION
        out.append((int(start * sample_rate), min(int(end * sample_rate), n_samples)))
    return tuple(out)


def audio_clip_bounds(n_samples: int, sample_rate: int = AUDIO_SAMPLE_RATE) -> List[Tuple[int, int]]:
    """Sample ranges of the 3 clips: starts spread evenly over [0, duration - 2 s] in exact rational arithmetic,
    then `int(t * sample_rate)` as upstream slices the waveform.  Cached per length (every 10-s segment asks the same question)."""
    return list(_audio_clip_bounds(int(n_samples), int(sample_rate)))


_WAV_DTYPES = {(1, 8): np.uint8, (1, 16): np.dtype("<i2"), (1, 32): np.dtype("<i4"), (3, 32): np.dtype("<f4")}


def _read_wav_raw(path: str) -> Tuple[np.ndarray, int]:
    """-> (samples as stored, shaped (n,) or (n, channels); sample_rate): what scipy.io.wavfile.read returns (same dtype, shape and
    values), by one read and a walk over the RIFF chunks for the files the reference writes (:1219 float32, ffmpeg pcm_s16le
    :1386-1394; also 8 / 32-bit PCM); anything else (extensible format tags, 24-bit, float64, RF64, a truncated file) goes to scipy,
    which also produces the error messages."""
    import struct
    try:
        with open(path, "rb") as fh:
            raw = fh.read()
        if raw[:4] == b"RIFF" and raw[8:12] == b"WAVE":
            pos, fmt, n = 12, None, len(raw)
            while pos + 8 <= n:
                tag, size = raw[pos:pos + 4], struct.unpack_from("<I", raw, pos + 4)[0]
                body = pos + 8
                if tag == b"fmt " and size >= 16:
                    fmt = struct.unpack_from("<HHIIHH", raw, body)          # format tag, channels, rate, byte rate, block align, bits
                elif tag == b"data":
                    if fmt is None or (fmt[0], fmt[5]) not in _WAV_DTYPES or fmt[1] < 1 or body + size > n:
                        break
                    dtype = np.dtype(_WAV_DTYPES[(fmt[0], fmt[5])])
                    if fmt[4] != fmt[1] * dtype.itemsize or size % fmt[4]:
                        break
                    data = np.frombuffer(raw, dtype=dtype, count=size // dtype.itemsize, offset=body)
                    return (data if fmt[1] == 1 else data.reshape(-1, fmt[1])), int(fmt[2])
                pos = body + size + (size & 1)
    except (OSError, struct.error):
        pass
    from scipy.io import wavfile
    rate, data = wavfile.read(path)
    return data, int(rate)


def _pcm_to_float(data: np.ndarray, out: np.ndarray = None) -> np.ndarray:
    """torchaudio.load's normalisation of integer PCM to [-1, 1) (float files pass through), optionally straight into `out`."""
    if data.dtype == np.float32:
        x = data
    elif data.dtype == np.int16:
        if out is not None:
            return np.divide(data, np.float32(32768.0), out=out)        # int16 -> float32 exactly, / 2^15 exactly
        x = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        x = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        x = (data.astype(np.float32) - 128.0) / 128.0
    else:
        x = data.astype(np.float32)
    if out is None:
        return x
    np.copyto(out, x)
    return out


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (samples float32 in [-1, 1] shaped (channels, n), sample_rate).  The reference writes its segment files with
    scipy.io.wavfile (hippocampal_memory.py:1219, float32) or ffmpeg pcm_s16le (:1386-1394); torchaudio.load
    normalises integer PCM to [-1, 1), which is reproduced here."""
    data, rate = _read_wav_raw(path)
    if data.ndim == 1:
        data = data[:, None]
    return np.ascontiguousarray(_pcm_to_float(data).T), rate


@lru_cache(maxsize=8)
def _resample_kernel(orig: int, new: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
    """Polyphase windowed-sinc kernels of torchaudio.functional.resample (sinc_interp_hann, its defaults), which
    upstream's load_and_transform_audio_data applies when a file is not at 16 kHz [upstream, recalled:
    torchaudio/functional/functional.py _get_sinc_resample_kernel].  orig / new are already divided by their gcd.
    -> (kernels (new, 1, 2*width + orig) fp32, width)."""
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kernels = torch.where(t == 0, torch.tensor(1.0, dtype=torch.float64), t.sin() / t) * window * scale
    return kernels.to(torch.float32), width


def resample_waveform(waveform: torch.Tensor, orig_freq: int, new_freq: int = AUDIO_SAMPLE_RATE) -> torch.Tensor:
    """(channels, n) fp32 at orig_freq -> (channels, ceil(n * new / orig)) at new_freq; runs where `waveform` lives.
    One strided conv1d against `new` phase kernels (polyphase form), as torchaudio's _apply_sinc_resample_kernel."""
    if orig_freq == new_freq:
        return waveform
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    kernels, width = _resample_kernel(orig, new)
    x = waveform.to(torch.float32)
    length = x.shape[-1]
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernels.to(x.device), stride=orig)      # (channels, new, frames)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    return y[..., : math.ceil(new * length / orig)]


@lru_cache(maxsize=4)
def _fbank_tables(device_str: str) -> Tuple[torch.Tensor, torch.Tensor]:
    """Hann window (400) and kaldi mel banks (128, 257) in the float32 torch CPU arithmetic torchaudio uses
    (torchaudio.compliance.kaldi.get_mel_banks / _feature_window_function [upstream, recalled]), uploaded once."""
    window = torch.hann_window(400, periodic=False)
    num_bins, n_fft_bins, fft_bin_width = AUDIO_MEL_BINS, 256, AUDIO_SAMPLE_RATE / 512
    mel_low = 1127.0 * math.log(1.0 + 20.0 / 700.0)
    mel_high = 1127.0 * math.log(1.0 + 0.5 * AUDIO_SAMPLE_RATE / 700.0)
    delta = (mel_high - mel_low) / (num_bins + 1)
    b = torch.arange(num_bins).unsqueeze(1)
    left, center, right = mel_low + b * delta, mel_low + (b + 1.0) * delta, mel_low + (b + 2.0) * delta
    mel = (1127.0 * (1.0 + (fft_bin_width * torch.arange(n_fft_bins)) / 700.0).log()).unsqueeze(0)
    banks = torch.max(torch.zeros(1), torch.min((mel - left) / (center - left), (right - mel) / (right - center)))
    banks = torch.nn.functional.pad(banks, (0, 1)).contiguous()
    return window.to(device_str), banks.to(device_str)


def melspec_clips_device(clips: torch.Tensor, mean: float = AUDIO_MEAN, std: float = AUDIO_STD,
                         out: torch.Tensor = None) -> torch.Tensor:
    """clips: (n_clips, clip_len) fp32 CUDA, mono 16 kHz -> (n_clips, 128, 204) fp32 CUDA (asynchronous); `out`: write there."""
    lib = _lib.load()
    _lib.require_gpu()
    if clips.dim() != 2 or clips.dtype != torch.float32 or not clips.is_cuda:
        raise ValueError("clips must be a 2-D float32 CUDA tensor (n_clips, clip_len)")
    clips = clips.contiguous()
    n, clip_len = clips.shape
    if out is None:
        out = torch.empty(n, AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH, dtype=torch.float32, device=clips.device)
    elif out.numel() != n * AUDIO_MEL_BINS * AUDIO_TARGET_LENGTH or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of n_clips x 128 x 204 elements")
    ws = torch.empty(lib.hmm_audio_fbank_workspace_bytes(n), dtype=torch.uint8, device=clips.device)
    window, banks = _fbank_tables(str(clips.device))
    _lib.check(lib.hmm_audio_fbank(clips.data_ptr(), n, clip_len, clip_len, window.data_ptr(), banks.data_ptr(),
                                   float(mean), float(std), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()),
               "hmm_audio_fbank")
    return out


_audio_staging = {}                                 # device -> [pinned (rows, len) fp32, event behind its last upload]


def _clips_to_melspec(clips: Sequence[np.ndarray], n_files: int, device) -> torch.Tensor:
    """clips: 3 per file in (file, clip) order, each a 1-D view of the file's samples as stored (int16 / float32 / ...), channel 0,
    16 kHz -> (n_files,3,1,128,204) on `device`.  The usual case -- every clip equally long (2 s whenever the file has 2 s) -- is
    one pinned buffer filled in place (PCM normalised on the way), ONE upload and ONE launch writing the result where it
    belongs; files shorter than a clip are grouped by clip length."""
    dev = torch.device(device)
    out = torch.empty(n_files, AUDIO_CLIPS_PER_VIDEO, 1, AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH, dtype=torch.float32, device=dev)
    lengths = {c.shape[0] for c in clips}
    if len(lengths) == 1:
        n, length = len(clips), lengths.pop()
        st = _audio_staging.get(str(dev))
        if st is None or st[0].shape[0] < n or st[0].shape[1] < length:
            if st is not None and st[1] is not None:
                st[1].synchronize()
            st = _audio_staging[str(dev)] = [torch.empty(max(n, 48), max(length, AUDIO_CLIP_DURATION * AUDIO_SAMPLE_RATE),
                                                         dtype=torch.float32, pin_memory=True), None]
        if st[1] is not None:
            st[1].synchronize()                      # the previous call's upload has left the buffer
        host = st[0].numpy()
        for r, c in enumerate(clips):
            _pcm_to_float(c, host[r, :length])
        with torch.cuda.device(dev):
            batch = torch.empty(n, length, dtype=torch.float32, device=dev)
            batch.copy_(st[0][:n, :length], non_blocking=True)
            st[1] = torch.cuda.Event()
            st[1].record()
            melspec_clips_device(batch, out=out)
        return out
    groups = {}                                      # clip length -> positions in `clips`
    for pos, c in enumerate(clips):
        groups.setdefault(c.shape[0], []).append(pos)
    flat = out.view(n_files * AUDIO_CLIPS_PER_VIDEO, AUDIO_MEL_BINS, AUDIO_TARGET_LENGTH)
    for length, positions in groups.items():
        batch = torch.from_numpy(np.stack([_pcm_to_float(clips[p]) for p in positions])).to(dev)
        flat[torch.tensor(positions, device=dev)] = melspec_clips_device(batch)
    return out


def transform_waveforms_device(waveforms: Sequence, device, sample_rate: int = AUDIO_SAMPLE_RATE) -> torch.Tensor:
    """waveforms: per file a (channels, n) or (n,) float array / tensor -> (B,3,1,128,204) fp32 on `device`.  Input at
    another rate than 16 kHz is resampled first (torchaudio's windowed-sinc polyphase filter, resample_waveform), as
    upstream does.  Channel 0 is analysed (kaldi.fbank's default channel).  Upstream's `waveform -= waveform.mean()` runs over all
    channels, but any constant offset is removed again per frame (remove_dc_offset), so for multi-channel input the
    result differs from the mono rule only in rounding; the reference always writes mono (:1204-1207, ffmpeg -ac 1)."""
    rates = list(sample_rate) if isinstance(sample_rate, (list, tuple)) else [sample_rate] * len(waveforms)
    clips = []
    for fi, w in enumerate(waveforms):
        if isinstance(w, torch.Tensor):
            w = w.detach().cpu().numpy()
        w = np.asarray(w)
        if w.dtype != np.float32:
            w = w.astype(np.float32)
        mono = w if w.ndim == 1 else w[0]
        if rates[fi] != AUDIO_SAMPLE_RATE:
            mono = resample_waveform(torch.from_numpy(np.ascontiguousarray(mono))[None].to(device), int(rates[fi]),
                                     AUDIO_SAMPLE_RATE)[0].cpu().numpy()
        clips += [mono[s:e] for s, e in _audio_clip_bounds(mono.shape[0], AUDIO_SAMPLE_RATE)]
    return _clips_to_melspec(clips, len(waveforms), device)


def load_and_transform_audio_data_device(audio_paths: Sequence[str], device) -> torch.Tensor:
    """Drop-in for imagebind.data.load_and_transform_audio_data(audio_paths, device) on wav files (any sample rate;
    the reference itself writes 16 kHz, hippocampal_memory.py:1219, :1386-1394).  A 16 kHz file goes from the samples as
    stored to the pinned upload buffer in one pass per clip (no float copy of the whole file, no transpose)."""
    clips = []
    for p in audio_paths:
        data, rate = _read_wav_raw(p)
        mono = data if data.ndim == 1 else data[:, 0]
        if rate != AUDIO_SAMPLE_RATE:
            x = torch.from_numpy(np.ascontiguousarray(_pcm_to_float(mono)))[None].to(device)
            mono = resample_waveform(x, rate, AUDIO_SAMPLE_RATE)[0].cpu().numpy()
        clips += [mono[s:e] for s, e in _audio_clip_bounds(mono.shape[0], AUDIO_SAMPLE_RATE)]
    return _clips_to_melspec(clips, len(audio_paths), device)
