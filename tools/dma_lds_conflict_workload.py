"""Workload for ONE PMC pass (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE): the LDS-DMA staging stream alone (tools/csrc/dma_stream.hip:
nothing but global_load_lds_dwordx4 into LDS, no ds_read) and the 16-query scan -- do the DMA writes themselves count as bank
conflicts?  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python3 tools/dma_lds_conflict_workload.py"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from probe_common import load_probe

import torch

L, lib = load_probe()
lib.hmm_probe_dma_stream.restype = C.c_int
lib.hmm_probe_dma_stream.argtypes = [C.c_void_p, C.c_uint, C.c_int, C.c_void_p, C.c_void_p]
src = torch.randn(256 << 20 >> 2, device="cuda")                       # 256 MiB
ticks = torch.zeros(4096, dtype=torch.int64, device="cuda")
for _ in range(3):
    L.check(lib.hmm_probe_dma_stream(src.data_ptr(), 2 << 20, 64, ticks.data_ptr(), L.stream_ptr()), "dma_stream")
torch.cuda.synchronize()
from hippomm_amd.vector_ops import FeatureStore
rows = torch.randn(200_000, 1024, device="cuda")
store = FeatureStore(rows)
q16 = torch.randn(16, 1024, device="cuda")
for _ in range(3):
    store.search_multi_device(q16, 32)
torch.cuda.synchronize()
