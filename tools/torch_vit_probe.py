"""Reference point only (never on the product path): the same ViT-H/14 vision tower written with stock PyTorch-ROCm
operators in bf16 and in fp32 (what the reference runs: no autocast), timed on the same GPU and batch as bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import torch_vit_lib as lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for dtype, iters in ((torch.bfloat16, 5), (torch.float32, 2)):
    ms = lib.time_forward(B, dtype, iters)
    print(f"stock PyTorch-ROCm ViT-H/14 {str(dtype).split('.')[1]:8s} B={B}: {ms:8.1f} ms/forward  {B/ms*1e3:8.1f} frames/s  "
          f"{B*334.98e9/ms/1e9:6.0f} TFLOP/s", flush=True)
