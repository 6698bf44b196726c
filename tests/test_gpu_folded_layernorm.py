"""GPU parity of the folded-LayerNorm path of the vision tower (hmm_encoder_set_folded_layernorm, include/hippomm_hip.h):

    LN(x) W^T + b  =  rstd * (xb W'^T) - rstd * mean * rowsum(W') + (W beta + b),   xb = bf16(x),  W' = bf16(gamma (.) W)

piece by piece against torch references built from the same bf16-rounded operands, bitwise across tile geometries and
between the fused in_proj + attention kernel and GEMM + attention, and as a tower against the fp32 oracle
(upstream: nn.LayerNorm(eps=1e-6) -> nn.MultiheadAttention in_proj / Mlp.fc1, restated in oracle/imagebind_oracle.py)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import imagebind_oracle as ib

pytestmark = pytest.mark.gpu


def _lib():
    from hippomm_amd import _lib as L
    return L, L.load()


def _bf16(t):
    return t.to(torch.bfloat16)


def _rowstat(L, lib, xb, eps=1e-6):
    rs = torch.empty(xb.shape[0], 2, device="cuda")
    L.check(lib.hmm_op_rowstat_bf16(xb.data_ptr(), rs.data_ptr(), xb.shape[0], xb.shape[1], eps, L.stream_ptr()), "rowstat")
    return rs


def _fold(L, lib, w0, gamma, beta, bias):
    n, d = w0.shape
    wf = torch.empty(n, d, dtype=torch.bfloat16, device="cuda")
    c1, c2 = torch.empty(n, device="cuda"), torch.empty(n, device="cuda")
    L.check(lib.hmm_op_fold_ln_weights(w0.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(), wf.data_ptr(),
                                       c1.data_ptr(), c2.data_ptr(), n, d, L.stream_ptr()), "fold")
    return wf, c1, c2


@pytest.mark.parametrize("dim", [768, 1024, 1280])
def test_rowstat_matches_two_pass_statistics(dim):
    L, lib = _lib()
    g = torch.Generator().manual_seed(dim)
    x = torch.randn(777, dim, generator=g) * 3.0 + torch.randn(777, 1, generator=g)
    x[5] = 0.0                                            # constant row: variance 0, eps decides
    x[6] = 7.25
    xb = _bf16(x).cuda()
    rs = _rowstat(L, lib, xb).cpu().double()
    v = xb.cpu().double()
    mean = v.mean(1)
    rstd = 1.0 / torch.sqrt(v.var(1, unbiased=False) + 1e-6)
    assert torch.allclose(rs[:, 0], rstd, rtol=2e-6, atol=0)
    assert torch.allclose(rs[:, 1], rstd * mean, rtol=2e-6, atol=1e-6 * float((rstd * mean).abs().max()))


def test_fold_weights():
    L, lib = _lib()
    g = torch.Generator().manual_seed(3)
    n, d = 640, 1280
    w0 = (torch.randn(n, d, generator=g) * 0.03).cuda()
    gamma = (1 + 0.2 * torch.randn(d, generator=g)).cuda()
    beta = (0.1 * torch.randn(d, generator=g)).cuda()
    bias = (0.05 * torch.randn(n, generator=g)).cuda()
    wf, c1, c2 = _fold(L, lib, w0, gamma, beta, bias)
    assert torch.equal(wf, _bf16(w0 * gamma))
    assert torch.allclose(c1.double(), wf.double().sum(1), rtol=1e-6, atol=1e-6)
    assert torch.allclose(c2.double(), w0.double() @ beta.double() + bias.double(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("gelu", [0, 1])
@pytest.mark.parametrize("M", [257, 771, 8224 + 130])
def test_gemm_ln_epilogue_against_layernorm_then_linear(M, gelu):
    """The folded GEMM against (a) its own definition evaluated in fp64 on the same bf16 operands and (b) LayerNorm followed
    by the Linear in fp64 on the unrounded inputs, within the bf16 rounding of the result; every tile geometry gives the
    same bits."""
    L, lib = _lib()
    N, K = 1280, 1280
    g = torch.Generator().manual_seed(M + gelu)
    x = torch.randn(M, K, generator=g) * 2.0 + 0.3
    x[:, ::160] *= 25.0                                   # a few outlier channels, as residual streams have
    w0 = torch.randn(N, K, generator=g) * 0.03
    gamma, beta = 1 + 0.2 * torch.randn(K, generator=g), 0.1 * torch.randn(K, generator=g)
    bias = 0.1 * torch.randn(N, generator=g)
    xb = _bf16(x).cuda()
    wf, c1, c2 = _fold(L, lib, w0.cuda(), gamma.cuda(), beta.cuda(), bias.cuda())
    rs = _rowstat(L, lib, xb)
    outs = []
    for tile in (0, 1, 2, 3, 4):
        c = torch.full((M + 2, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        L.check(lib.hmm_op_gemm_bf16_ln(xb.data_ptr(), wf.data_ptr(), c2.data_ptr(), c.data_ptr(), M, N, K, gelu,
                                        rs.data_ptr(), 1, c1.data_ptr(), tile, L.stream_ptr()), "gemm_ln")
        assert torch.isnan(c[M:].float()).all()
        outs.append(c[:M])
    for o in outs[1:]:
        assert torch.equal(o, outs[0]), "tile geometries disagree"
    got = outs[0].double().cpu()
    act = (lambda t: F.gelu(t)) if gelu else (lambda t: t)
    rsd = rs.double().cpu()
    own = act(rsd[:, :1] * (xb.double().cpu() @ wf.double().cpu().T) - rsd[:, 1:] * c1.double().cpu() + c2.double().cpu())
    assert ((got - own).abs() <= 2.0 ** -8 * own.abs() * 1.01 + 2e-5 * own.abs().max()).all()
    ref = act(F.layer_norm(x.double(), (K,), gamma.double(), beta.double(), 1e-6) @ w0.double().T + bias.double())
    rel = float((got - ref).norm() / ref.norm())
    print(f"M={M} gelu={gelu}: relative error vs fp64 LayerNorm -> Linear {rel:.2e}")
    assert rel < 6e-3                                     # two bf16 operand roundings: ~2.5e-3 as for LN -> bf16 -> GEMM


@pytest.mark.parametrize("dim", [768, 1024, 1280])
def test_rowstat_finalize_on_rows_with_large_means(dim):
    """Chunk statistics -> row statistics (Chan's combination) on rows whose mean is up to ~30 standard deviations: the
    centred chunk sums keep the variance accurate where sum(x^2) - sum(x)^2 / n would not."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(dim + 1)
    x = torch.randn(300, dim, generator=g) * torch.rand(300, 1, generator=g) * 3 + 30.0 * torch.randn(300, 1, generator=g)
    x[7] = 0.0
    x[8] = -3.5
    xb = _bf16(x).double()
    v = xb.view(300, dim // 64, 64)
    part = torch.stack([v.sum(-1), ((v - v.mean(-1, keepdim=True)) ** 2).sum(-1)], dim=-1).float().cuda().contiguous()
    rs = torch.empty(300, 2, device="cuda")
    L.check(lib.hmm_op_rowstat_finalize(part.data_ptr(), rs.data_ptr(), 300, dim, 1e-6, L.stream_ptr()), "rowstat_finalize")
    mean = xb.mean(1)
    rstd = 1.0 / torch.sqrt(xb.var(1, unbiased=False) + 1e-6)
    rs = rs.double().cpu()
    assert torch.allclose(rs[:, 0], rstd, rtol=2e-5, atol=0)
    assert torch.allclose(rs[:, 1], rstd * mean, rtol=2e-5, atol=1e-6 * float((rstd * mean).abs().max()))


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4])
def test_gemm_resid_xb_epilogue(tile):
    """C += A W^T + bias exactly as HMM_EPI_BIAS_RESID_F32, plus xb = bf16(C) and the chunk statistics of xb."""
    L, lib = _lib()
    M, N, K = 771, 1280, 1280
    g = torch.Generator().manual_seed(9)
    a = _bf16(torch.randn(M, K, generator=g)).cuda()
    w = _bf16(torch.randn(N, K, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    c0 = torch.randn(M, N, generator=g).cuda()
    plain = c0.clone()
    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), plain.data_ptr(), M, N, K, 2, tile,
                                      L.stream_ptr()), "gemm resid")
    c = c0.clone()
    xb = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((M + 1, N // 64, 2), float("nan"), device="cuda")
    L.check(lib.hmm_op_gemm_bf16_resid_xb(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), xb.data_ptr(),
                                          part.data_ptr(), M, N, K, tile, L.stream_ptr()), "gemm resid xb")
    assert torch.equal(c, plain)
    assert torch.equal(xb[:M], _bf16(c))
    assert torch.isnan(xb[M:].float()).all() and torch.isnan(part[M:]).all()
    # chunk statistics: (sum, sum of squares about the chunk mean) of every 64 columns of the stored bf16 rows
    v = xb[:M].double().view(M, N // 64, 64)
    s, q = v.sum(-1), ((v - v.mean(-1, keepdim=True)) ** 2).sum(-1)
    assert torch.allclose(part[:M, :, 0].double(), s, rtol=1e-6, atol=1e-5)
    assert torch.allclose(part[:M, :, 1].double(), q, rtol=1e-5, atol=1e-6)
    # ... combined per row they are the statistics of the pass over xb, and without the buffer nothing else changes
    rs = torch.empty(M, 2, device="cuda")
    L.check(lib.hmm_op_rowstat_finalize(part.data_ptr(), rs.data_ptr(), M, N, 1e-6, L.stream_ptr()), "rowstat_finalize")
    ref_rs = _rowstat(L, lib, xb[:M].contiguous())
    assert torch.allclose(rs, ref_rs, rtol=3e-6, atol=1e-6)
    c2_, xb2 = c0.clone(), torch.empty_like(xb)
    L.check(lib.hmm_op_gemm_bf16_resid_xb(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c2_.data_ptr(), xb2.data_ptr(), None,
                                          M, N, K, tile, L.stream_ptr()), "gemm resid xb, no statistics")
    assert torch.equal(c2_, c) and torch.equal(xb2[:M], xb[:M])
    if tile:                                              # every tile geometry reduces a chunk in the same order: same bits
        p0, x0, c3 = torch.empty_like(part), torch.empty_like(xb), c0.clone()
        L.check(lib.hmm_op_gemm_bf16_resid_xb(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c3.data_ptr(), x0.data_ptr(),
                                              p0.data_ptr(), M, N, K, 0, L.stream_ptr()), "gemm resid xb 128")
        assert torch.equal(p0[:M], part[:M])
    if tile:                                              # and the same bits as the 128x128 tiles
        ref = c0.clone()
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), ref.data_ptr(), M, N, K, 2, 0,
                                          L.stream_ptr()), "gemm resid 128")
        assert torch.equal(c, ref)


@pytest.mark.parametrize("n_img", [1, 3, 9])
def test_fused_qkv_attention_ln_equals_gemm_ln_plus_attention(n_img):
    L, lib = _lib()
    T, D, H, dh = 257, 1280, 16, 80
    g = torch.Generator().manual_seed(40 + n_img)
    x = torch.randn(n_img * T, D, generator=g) * 1.5 + 0.2
    w0 = torch.randn(3 * D, D, generator=g) * 0.03
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.1 * torch.randn(D, generator=g)
    bias = 0.1 * torch.randn(3 * D, generator=g)
    xb = _bf16(x).cuda()
    wf, c1, c2 = _fold(L, lib, w0.cuda(), gamma.cuda(), beta.cuda(), bias.cuda())
    rs = _rowstat(L, lib, xb)
    qkv = torch.empty(n_img * T, 3 * D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_ln(xb.data_ptr(), wf.data_ptr(), c2.data_ptr(), qkv.data_ptr(), n_img * T, 3 * D, D, 0,
                                    rs.data_ptr(), 1, c1.data_ptr(), -1, L.stream_ptr()), "gemm_ln")
    two = torch.empty(n_img * T, D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_attention_bf16(qkv.data_ptr(), two.data_ptr(), n_img, T, H, dh, None, None, L.stream_ptr()), "attention")
    # cls rows: gathered rows, their own statistics (same kernel, same values), small-M GEMM
    cls_rows = xb.view(n_img, T, D)[:, 0].contiguous()
    rsc = _rowstat(L, lib, cls_rows)
    assert torch.equal(rsc, rs.view(n_img, T, 2)[:, 0])
    qkv_cls = torch.empty(n_img, 3 * D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_ln(cls_rows.data_ptr(), wf.data_ptr(), c2.data_ptr(), qkv_cls.data_ptr(), n_img, 3 * D, D, 0,
                                    rsc.data_ptr(), 1, c1.data_ptr(), -1, L.stream_ptr()), "gemm_ln cls")
    assert torch.equal(qkv_cls, qkv.view(n_img, T, 3 * D)[:, 0])
    one = torch.full((n_img * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_qkv_attention_ln_bf16(xb.data_ptr(), wf.data_ptr(), c2.data_ptr(), qkv_cls.data_ptr(), one.data_ptr(),
                                             n_img, rs.data_ptr(), c1.data_ptr(), L.stream_ptr()), "qkv_attention_ln")
    assert torch.equal(one, two), f"max diff {(one.float() - two.float()).abs().max().item()}"
    # and the whole thing against fp64 LayerNorm -> in_proj -> attention
    q, k, v = (F.layer_norm(x.double(), (D,), gamma.double(), beta.double(), 1e-6) @ w0.double().T
               + bias.double()).reshape(n_img, T, 3, H, dh).unbind(2)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    want = (torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_img * T, D)
    rel = float((one.double().cpu() - want).norm() / want.norm())
    print(f"n_img={n_img}: relative error vs fp64 LN -> in_proj -> attention {rel:.2e}")
    assert rel < 1e-2


def _check(got, want, what):
    got, want = got.float().cpu(), want.float().cpu()
    cos = F.cosine_similarity(got, want, dim=1)
    err = (got - want).abs().max().item()
    print(f"{what}: min cos {cos.min().item():.7f}  max|diff| {err:.3e}")
    assert (1 - cos).max().item() <= 2e-4 and err <= 2e-2


@pytest.mark.parametrize("batch,depth", [(5, 3), (70, 3), (33, 6)])
def test_tower_folded_and_plain_layernorm_both_match_the_oracle(batch, depth):
    """The vision tower with LayerNorm folded (opt-in) and as its own kernel (default): both within the encoder tolerance of the fp32
    oracle ('rich' weights: random gamma / beta / biases), close to each other, and the folded path bitwise independent of
    the batch it rides in, the stream count and the fused-attention switch."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, depth)
    st = ib.synthetic_state(spec, seed=77, init="rich")
    x = torch.randn(batch, 3, 224, 224, generator=torch.Generator().manual_seed(batch))
    pick = sorted({0, batch // 2, batch - 1})
    want = ib.vision_forward(x[pick], st, spec)
    tower = HipTower("vision", st, depth=depth)
    tower.set_folded_layernorm(True)                      # opt-in: the LayerNorm kernel is the default path
    folded = tower(x)
    _check(folded[pick], want, f"folded LayerNorm, depth {depth}, B={batch}")
    assert torch.equal(folded, tower(x, max_batch=4))
    tower.set_fused_attention(False)
    assert torch.equal(folded, tower(x))
    tower.set_fused_attention(True)
    tower.set_streams(1)
    assert torch.equal(folded, tower(x))
    tower.set_streams(2)
    tower.set_folded_layernorm(False)
    plain = tower(x)
    _check(plain[pick], want, f"LayerNorm kernel, depth {depth}, B={batch}")
    cos = F.cosine_similarity(folded.float(), plain.float(), dim=1)
    assert (1 - cos).max().item() <= 2e-4
    tower.set_folded_layernorm(True)
    assert torch.equal(folded, tower(x))
