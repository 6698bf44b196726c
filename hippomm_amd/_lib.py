"""ctypes binding of libhippomm_hip.so (C ABI: include/hippomm_hip.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libhippomm_hip.so"

_lib = None

c_f32p = C.c_void_p       # device pointers travel as integers (tensor.data_ptr())
c_ptr = C.c_void_p

_SIGNATURES = {
    "hmm_abi_version": (C.c_int, []),
    "hmm_last_error": (C.c_char_p, []),
    "hmm_device_supported": (C.c_int, []),
    "hmm_cosine_topk_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "hmm_cosine_topk": (C.c_int, [c_ptr, C.c_int64, C.c_int, c_ptr, C.c_int, c_ptr, c_ptr, c_ptr,
                                  c_ptr, C.c_size_t, c_ptr]),
    "hmm_cosine_topk_keys": (C.c_int, [c_ptr, C.c_int64, C.c_int, c_ptr, C.c_int, c_ptr,
                                       c_ptr, C.c_size_t, c_ptr]),
    "hmm_topk_merge_keys": (C.c_int, [c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "hmm_shadow_store_bytes": (C.c_size_t, [C.c_int64]),
    "hmm_shadow_store_build": (C.c_int, [c_ptr, C.c_int64, C.c_int, c_ptr, C.c_size_t, c_ptr]),
    "hmm_cosine_topk_prefilter_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
    "hmm_cosine_topk_prefilter": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_int, c_ptr, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                            c_ptr, C.c_size_t, c_ptr]),
    "hmm_cosine_topk_segmented_prefilter_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "hmm_cosine_topk_segmented_prefilter": (C.c_int, [c_ptr, c_ptr, C.c_int64, C.c_int, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr,
                                                      c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "hmm_cosine_topk_segmented_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "hmm_cosine_topk_segmented": (C.c_int, [c_ptr, C.c_int64, C.c_int, c_ptr, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr,
                                            c_ptr, C.c_size_t, c_ptr]),
    "hmm_rank_segment_hits": (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "hmm_cosine_topk_multi_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
    "hmm_cosine_topk_multi": (C.c_int, [c_ptr, C.c_int64, C.c_int, c_ptr, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr,
                                        C.c_size_t, c_ptr]),
    "hmm_gram_select_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hmm_gram_select": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_float, c_ptr, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "hmm_preprocess_vision_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "hmm_preprocess_vision_u8": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, C.c_int, c_ptr, c_ptr, C.c_int,
                                           C.c_int, C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "hmm_host_rgbx_to_rgb": (C.c_int, [c_ptr, C.c_size_t, c_ptr]),
    "hmm_host_arrow_rgbx_to_rgb": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "hmm_audio_fbank_workspace_bytes": (C.c_size_t, [C.c_int]),
    "hmm_audio_fbank": (C.c_int, [c_ptr, C.c_int, C.c_int, C.c_int64, c_ptr, c_ptr, C.c_float, C.c_float, c_ptr, c_ptr,
                                  C.c_size_t, c_ptr]),
    "hmm_encoder_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "hmm_encoder_destroy": (None, [c_ptr]),
    "hmm_encoder_load_param": (C.c_int, [c_ptr, C.c_char_p, c_ptr, C.c_int64, c_ptr]),
    "hmm_encoder_missing_params": (C.c_int, [c_ptr]),
    "hmm_encoder_workspace_bytes": (C.c_size_t, [c_ptr, C.c_int]),
    "hmm_encoder_forward": (C.c_int, [c_ptr, c_ptr, C.c_int, c_ptr, c_ptr, C.c_size_t, c_ptr]),
    "hmm_encoder_flops": (C.c_double, [c_ptr, C.c_int]),
    "hmm_encoder_flops_executed": (C.c_double, [c_ptr, C.c_int]),
    "hmm_encoder_set_streams": (C.c_int, [c_ptr, C.c_int]),
    "hmm_encoder_set_fused_attention": (C.c_int, [c_ptr, C.c_int]),
    "hmm_op_gemm_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "hmm_op_layernorm_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr]),
    "hmm_op_gemm_bf16_tile": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "hmm_op_gemm_bf16_splitk": (C.c_int, [c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "hmm_op_layernorm_reduce_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, C.c_int, C.c_float, c_ptr]),
    "hmm_op_attention_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr, c_ptr, c_ptr]),
    "hmm_op_qkv_attention_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, c_ptr]),
    "hmm_op_qkv_attention_audio_bf16": (C.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, C.c_int, c_ptr]),
    "hmm_op_attention_causal_bf16": (C.c_int, [c_ptr, c_ptr, C.c_int, C.c_int, C.c_int, C.c_int, c_ptr]),
    "hmm_op_scan_topk_only": (C.c_int, [c_ptr, C.c_int64, c_ptr, C.c_int, c_ptr, c_ptr]),
    "hmm_op_scan_sims": (C.c_int, [c_ptr, C.c_int64, c_ptr, c_ptr, c_ptr]),
    "hmm_json_find_matrices": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, c_ptr, C.c_int, C.POINTER(C.c_int)]),
    "hmm_json_parse_matrix_f32": (C.c_int, [C.c_char_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, c_ptr]),
    "hmm_json_matrix_text_bound": (C.c_size_t, [C.c_size_t, C.c_size_t, C.c_int]),
    "hmm_json_write_matrix_f64": (C.c_int, [c_ptr, C.c_size_t, C.c_size_t, C.c_int, c_ptr, C.c_size_t, C.POINTER(C.c_size_t)]),
}


class HippoMMHipError(RuntimeError):
    pass


def bind(path):
    """dlopen `path` and apply the binding table (also used by tools/ for the probe build of the same sources)."""
    # PyTorch-ROCm ships its own HIP / HSA runtime libraries.  They must be in the process BEFORE this library pulls in the
    # ones under /opt/rocm: loaded the other way round (library first, torch later) the library's runtime finds no device
    # ("no ROCm-capable device is detected") although torch sees the GPU.
    import torch  # noqa: F401
    lib = C.CDLL(str(path))
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the ABI and the binding disagree
        fn.restype = res
        fn.argtypes = args
    return lib


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise HippoMMHipError(
            f"{LIB_PATH} is missing: build the HIP extension first (python -m hippomm_amd.build). "
            "hippomm_amd has no CPU fallback.")
    _lib = bind(LIB_PATH)
    return _lib


def check(status: int, what: str):
    if status != 0:
        msg = load().hmm_last_error().decode("utf-8", "replace")
        raise HippoMMHipError(f"{what} failed (status {status}): {msg}")


_checked_devices = set()


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise HippoMMHipError("hippomm_amd needs a ROCm GPU (MI355X / gfx950); no CPU fallback exists")
    dev = torch.cuda.current_device()
    if dev not in _checked_devices:                      # once per device: gfx950 with 256 CUs, or refuse loudly
        check(load().hmm_device_supported(), "hmm_device_supported")
        _checked_devices.add(dev)
    return torch.device("cuda", dev)


def stream_ptr():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
