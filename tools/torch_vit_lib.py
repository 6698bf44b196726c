"""Stock PyTorch-ROCm ViT-H/14 vision tower (vendor GEMMs, F.scaled_dot_product_attention, F.layer_norm, F.gelu): a
reference point for bench.py and tools/torch_vit_probe.py.  Never on the product path."""
import time

import torch
import torch.nn.functional as F

D, H, MLP, L, T = 1280, 16, 5120, 32, 257


def make(dtype):
    g = torch.Generator(device="cuda").manual_seed(0)
    r = lambda *s: (torch.randn(*s, device="cuda", generator=g) * 0.02).to(dtype)
    ones = lambda: torch.ones(D, device="cuda", dtype=dtype)
    zeros = lambda: torch.zeros(D, device="cuda", dtype=dtype)
    blocks = [dict(n1w=ones(), n1b=zeros(), qkv=r(3 * D, D), qkvb=r(3 * D), o=r(D, D), ob=r(D), n2w=ones(), n2b=zeros(),
                   f1=r(MLP, D), f1b=r(MLP), f2=r(D, MLP), f2b=r(D)) for _ in range(L)]
    return dict(patch=r(D, 3, 14, 14), cls=r(1, 1, D), pos=r(1, T, D), blocks=blocks, hw=r(1024, D), pre_w=ones(), pre_b=zeros())


@torch.no_grad()
def forward(x, p):
    b = x.shape[0]
    t = F.conv2d(x, p["patch"], stride=14).flatten(2).transpose(1, 2)
    t = torch.cat([p["cls"].expand(b, -1, -1), t], 1) + p["pos"]
    t = F.layer_norm(t, (D,), p["pre_w"], p["pre_b"], 1e-6)
    for w in p["blocks"]:
        y = F.layer_norm(t, (D,), w["n1w"], w["n1b"], 1e-6)
        q, k, v = F.linear(y, w["qkv"], w["qkvb"]).view(b, T, 3, H, D // H).permute(2, 0, 3, 1, 4)
        a = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(b, T, D)
        t = t + F.linear(a, w["o"], w["ob"])
        y = F.layer_norm(t, (D,), w["n2w"], w["n2b"], 1e-6)
        t = t + F.linear(F.gelu(F.linear(y, w["f1"], w["f1b"])), w["f2"], w["f2b"])
    y = F.layer_norm(t[:, 0], (D,), p["pre_w"], p["pre_b"], 1e-6)
    return F.normalize(F.linear(y, p["hw"]).float(), dim=-1)


def time_forward(batch, dtype, iters):
    p = make(dtype)
    x = torch.randn(batch, 3, 224, 224, device="cuda").to(dtype)
    for _ in range(2):
        forward(x, p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        forward(x, p)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    del p, x
    torch.cuda.empty_cache()
    return ms
