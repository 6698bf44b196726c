"""hippomm_amd -- MI355X-native (gfx950) hot path of HippoMM.

Host-side mirror of the three reference call sites, bound to hand-written HIP kernels
through the C ABI in include/hippomm_hip.h:

    hippomm_amd.encoder.ImageBind                    <- hippomm/models/foundation_models.py:21-151
    hippomm_amd.consolidation._select_key_frames     <- hippomm/core/hippocampal_memory.py:944-967
    hippomm_amd.vector_ops.top_k_cosine_similarity   <- hippomm/utils/vector_ops.py:151-188
    hippomm_amd.sharding                             one-process-per-GPU sharding (RCCL all-gather)
"""
__version__ = "0.1.0"
