"""GPU parity of each encoder kernel on its own against a torch fp32 reference of the same op
(computed on the CPU from the same bf16-rounded operands)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_RESID_F32, EPI_F32 = 0, 1, 2, 3
BF16_EPS = 2.0 ** -8          # half an ulp of bf16, relative


def _lib():
    from hippomm_amd import _lib as L
    lib = L.load()
    return L, lib


def _bf16(t):
    return t.to(torch.bfloat16)


def _close_bf16(got, want, extra_atol=0.0):
    """got is bf16-rounded output of an fp32-accumulated op; want is the fp32 reference."""
    got, want = got.float().cpu(), want.float().cpu()
    tol = BF16_EPS * want.abs() * 1.01 + 1e-5 * want.abs().max() + extra_atol
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())} / {bad.numel()} off; worst {float((got - want).abs().max()):.4g}"


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("M,N,K", [(257, 1280, 1280), (514, 3840, 1280), (300, 5120, 1280), (257, 1280, 5120),
                                   (1, 1024, 1280), (130, 768, 256), (512, 1280, 640), (1000, 2304, 768)])
def test_gemm_bias_bf16(variant, M, N, K):
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf16(torch.randn(M, K, generator=g))
    w = _bf16(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    want = a.float() @ w.float().T + bias
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    c = torch.full((M + 3, N), float("nan"), dtype=torch.bfloat16, device="cuda")   # canary rows past M
    L.check(lib.hmm_op_gemm_bf16_tile(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K,
                                          EPI_BIAS_BF16, variant, L.stream_ptr()), "gemm")
    _close_bf16(c[:M], want)
    assert torch.isnan(c[M:].float()).all(), "rows past M were written"


def test_gemm_epilogues_bitwise_equal_across_tile_geometries():
    """A row must get the same bits whichever tile geometry computes it (the tower picks the geometry by batch size and peels
    the last row tile into 128x128 tiles): same K order in every main loop and the same sum order in every epilogue --
    (accumulator + bias) + residual -- checked with a non-zero bias and residual."""
    L, lib = _lib()
    M, N, K = 1000, 1280, 1280
    g = torch.Generator().manual_seed(77)
    a = _bf16(torch.randn(M, K, generator=g)).cuda()
    w = _bf16(torch.randn(N, K, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    c0 = (torch.randn(M, N, generator=g) * 3.0).cuda()
    for epi, dtype in ((EPI_BIAS_RESID_F32, torch.float32), (EPI_BIAS_GELU_BF16, torch.bfloat16), (EPI_BIAS_BF16, torch.bfloat16),
                       (EPI_F32, torch.float32)):
        outs = []
        for variant in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
            c = c0.clone().to(dtype)
            L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, variant,
                                              L.stream_ptr()), "gemm")
            outs.append(c)
        for v, o in enumerate(outs[1:], 1):
            assert torch.equal(o, outs[0]), f"epilogue {epi}: tile geometry {v} differs from 128x128"


def _dispatch_fuzz_shapes():
    """Seeded shapes around every row / tile-count limit of the dispatcher (gemm_bf16.hip: sliver vs ring, 32 / 64 / 128-row ring
    tiles and their deep-K forms, the 128 x 64 rule up to 320 rows, one 128 x 128 ring tile per CU, ping-pong with peeled tails)."""
    rng = np.random.default_rng(20261003)
    Ns, Ks = (128, 256, 768, 1024, 1280, 2304, 3072, 3840, 4096, 5120), (64, 128, 192, 256, 768, 1024, 1280, 3072, 4096, 5120)
    edge_m = (1, 15, 16, 17, 31, 33, 63, 64, 65, 77, 127, 128, 129, 229, 255, 256, 257, 319, 320, 321, 514, 640, 687, 693, 771, 1028,
              2056, 2570, 3084, 3341, 4112)
    shapes = []
    for i in range(56):
        M = int(edge_m[i % len(edge_m)]) if i < 40 else int(np.exp(rng.uniform(0, np.log(5000))))
        shapes.append((M, int(rng.choice(Ns)), int(rng.choice(Ks)), int(rng.integers(0, 4))))
    # the ring tail peel: three frames' fc1 (7 x 40 ring tiles for 6 x 40 + 3 rows), four frames' qkv, and its limits (16 / 17 tail rows)
    shapes += [(771, 5120, 1280, 1), (1028, 3840, 1280, 0), (771, 5120, 256, 2), (768 + 16, 5120, 128, 3), (768 + 17, 5120, 128, 0)]
    # the audit rules of round 5: 128 x 64 up to 700 rows (six questions' fc1, one segment's qkv) and up to 1536 rows for K = 5120 (four frames'
    # fc2); 64 x 128 from 450 small tiles on (six frames' fc2 / out-proj, 24 questions' fc2); ping-pong for qkv from 80 tiles on
    shapes += [(462, 4096, 1024, 1), (687, 2304, 768, 0), (700, 3072, 1024, 0), (701, 3072, 1024, 0), (1028, 1280, 5120, 2), (1536, 1280, 5120, 2),
               (1537, 1280, 5120, 2), (1542, 1280, 5120, 2), (1542, 1280, 1280, 2), (1848, 1024, 4096, 2), (1694, 1024, 1024, 2),
               (1285, 3840, 1280, 0), (1540, 3072, 1024, 0), (1028, 3840, 1280, 0), (1285, 5120, 1280, 1)]
    return shapes


@pytest.mark.parametrize("M,N,K,epi", _dispatch_fuzz_shapes())
def test_gemm_dispatcher_fuzz_auto_choice_has_the_bits_of_the_reference_tile(M, N, K, epi):
    """Whatever kernel the dispatcher picks for a shape (AUTO: incl. the sliver kernel; AUTO_TILED: what the towers use) -- the
    output has the bits of the double-buffered 128 x 128 kernel and is close to fp32 torch; rows past M stay untouched."""
    L, lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N + K + epi)
    a = _bf16(torch.randn(M, K, device="cuda", generator=g))
    w = _bf16(torch.randn(N, K, device="cuda", generator=g) * 0.05)
    bias = torch.randn(N, device="cuda", generator=g)
    dtype = torch.float32 if epi in (EPI_BIAS_RESID_F32, EPI_F32) else torch.bfloat16
    c0 = (torch.randn(M + 2, N, device="cuda", generator=g) * 2.0).to(dtype)
    outs = {}
    for tile in (0, -1, -2):
        c = c0.clone()
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, tile, L.stream_ptr()), "gemm")
        outs[tile] = c
    assert torch.equal(outs[-1], outs[0]), "AUTO differs from the reference tile"
    assert torch.equal(outs[-2], outs[0]), "AUTO_TILED differs from the reference tile"
    assert torch.equal(outs[0][M:], c0[M:]), "rows past M were written"
    want = a.float() @ w.float().T + bias                         # EPI_F32 adds the bias too when it is non-null
    if epi == EPI_BIAS_GELU_BF16:
        want = F.gelu(want)
    if epi == EPI_BIAS_RESID_F32:
        want = want + c0[:M]
    if dtype == torch.bfloat16:
        _close_bf16(outs[0][:M], want, extra_atol=2e-3)
    else:
        assert torch.allclose(outs[0][:M], want, rtol=2e-4, atol=2e-3 * float(want.abs().max()) / 10 + 1e-3)


@pytest.mark.parametrize("M,N,K", [(77, 3072, 1024), (77, 1024, 4096), (1, 1024, 1024), (128, 5120, 1280), (16, 1280, 5120),
                                   (100, 768, 3072), (257, 3840, 1280), (640, 3840, 1280)])
def test_gemm_few_rows_sliver_kernel_bitwise_equal_to_the_tiled_kernels(M, N, K):
    """One question is 77 token rows: the dispatcher sends few-row products to the one-wave sliver kernel (16, 32 or 64 rows per
    wave by shape -- all three classes are in the list).  Same bits as the 128x128 tiles for every epilogue it has, rows past M
    untouched, and AUTO agrees with both."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    a = _bf16(torch.randn(M, K, generator=g)).cuda()
    w = _bf16(torch.randn(N, K, generator=g) * 0.05).cuda()
    bias = torch.randn(N, generator=g).cuda()
    c0 = (torch.randn(M + 2, N, generator=g) * 3.0).cuda()
    for epi, dtype in ((EPI_BIAS_RESID_F32, torch.float32), (EPI_BIAS_GELU_BF16, torch.bfloat16), (EPI_BIAS_BF16, torch.bfloat16),
                       (EPI_F32, torch.float32)):
        outs = []
        for variant in (0, 5, -1, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
            c = c0.clone().to(dtype)
            L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K, epi, variant,
                                              L.stream_ptr()), "gemm")
            outs.append(c)
        assert torch.equal(outs[1], outs[0]), f"epilogue {epi}: the sliver kernel differs from 128x128 tiles"
        assert torch.equal(outs[2], outs[0]), f"epilogue {epi}: the default dispatch differs from 128x128 tiles"
        assert torch.equal(outs[3], outs[0]), f"epilogue {epi}: the 4-deep ring differs from the double-buffered 128x128 tiles"
        assert torch.equal(outs[4], outs[0]), f"epilogue {epi}: 64x64 tiles behind the ring differ from 128x128 tiles"
        assert torch.equal(outs[5], outs[0]), f"epilogue {epi}: 32x32 tiles behind the ring differ from 128x128 tiles"
        assert torch.equal(outs[6], outs[0]), f"epilogue {epi}: the deep-K ring (2 K-tiles per stage) differs from 128x128 tiles"
        assert torch.equal(outs[7], outs[0]), f"epilogue {epi}: the deep-K ring (4 K-tiles per stage) differs from 128x128 tiles"
        assert torch.equal(outs[8], outs[0]), f"epilogue {epi}: the 64x64 deep-K ring differs from 128x128 tiles"
        assert torch.equal(outs[9], outs[0]), f"epilogue {epi}: 128x64 tiles behind the ring differ from 128x128 tiles"
        assert torch.equal(outs[10], outs[0]), f"epilogue {epi}: 64x128 tiles behind the ring differ from 128x128 tiles"
        assert torch.equal(outs[11], outs[0]), f"epilogue {epi}: the eight-wave 128x128 ring differs from 128x128 tiles"
        assert torch.equal(outs[12], outs[0]), f"epilogue {epi}: the eight-wave 128x64 ring differs from 128x128 tiles"
        assert torch.equal(outs[1][M:], c0[M:].to(dtype)), "rows past M were written"
    want = a.float().cpu() @ w.float().cpu().T + bias.cpu()
    _close_bf16(outs[1][:M], want)                     # the last epilogue of the loop is plain fp32 + bias: a value check on top


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_gemm_gelu_resid_f32_epilogues(variant):
    L, lib = _lib()
    M, N, K = 771, 1280, 1280
    g = torch.Generator().manual_seed(5)
    a = _bf16(torch.randn(M, K, generator=g))
    w = _bf16(torch.randn(N, K, generator=g) * 0.05)
    bias = torch.randn(N, generator=g)
    lin = a.float() @ w.float().T + bias
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    # GELU(erf)
    c = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_tile(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), c.data_ptr(), M, N, K,
                                          EPI_BIAS_GELU_BF16, variant, L.stream_ptr()), "gemm gelu")
    _close_bf16(c, F.gelu(lin), extra_atol=2e-5)
    # fp32 residual, in place
    x0 = torch.randn(M, N, generator=g)
    x = x0.clone().cuda()
    L.check(lib.hmm_op_gemm_bf16_tile(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), x.data_ptr(), M, N, K,
                                          EPI_BIAS_RESID_F32, variant, L.stream_ptr()), "gemm resid")
    torch.testing.assert_close(x.cpu(), x0 + lin, rtol=2e-5, atol=2e-4)
    # plain fp32, no bias
    y = torch.empty(M, N, dtype=torch.float32, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_tile(ad.data_ptr(), wd.data_ptr(), None, y.data_ptr(), M, N, K,
                                          EPI_F32, variant, L.stream_ptr()), "gemm f32")
    torch.testing.assert_close(y.cpu(), lin - bias, rtol=2e-5, atol=2e-4)


@pytest.mark.parametrize("epi", [EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_RESID_F32, EPI_F32])
def test_gemm_many_tiles_ragged_m(epi):
    """1300 tiles (several rounds on 256 CUs), last M-tile ragged (M % 256 = 57): clamped staging rows and the
    masked LDS-transposed epilogue of every epilogue kind."""
    L, lib = _lib()
    M, N, K = 256 * 259 + 57, 1280, 256
    g = torch.Generator(device="cuda").manual_seed(epi)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    lin = a.float() @ w.float().T
    if epi in (EPI_BIAS_BF16, EPI_BIAS_GELU_BF16):
        c = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K,
                                              epi, 4, L.stream_ptr()), "gemm")
        want = lin + bias
        _close_bf16(c, F.gelu(want) if epi == EPI_BIAS_GELU_BF16 else want, extra_atol=2e-5)
    elif epi == EPI_BIAS_RESID_F32:
        x0 = torch.randn(M, N, device="cuda", generator=g)
        x = x0.clone()
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K,
                                              epi, 4, L.stream_ptr()), "gemm")
        torch.testing.assert_close(x, x0 + lin + bias, rtol=2e-5, atol=2e-4)
    else:
        y = torch.empty(M, N, dtype=torch.float32, device="cuda")
        L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), None, y.data_ptr(), M, N, K,
                                              epi, 4, L.stream_ptr()), "gemm")
        torch.testing.assert_close(y, lin, rtol=2e-5, atol=2e-4)


def test_gemm_cfg2_rows_peeled_tail():
    """M = 65792 (256 frames x 257 tokens): the default path peels the 257th M-tile into a second
    launch; check both launches against torch on sampled rows, including the seam and the last row."""
    L, lib = _lib()
    M, N, K = 65792, 1280, 1280
    g = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.randn(M, K, device="cuda", generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    c = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), c.data_ptr(), M, N, K,
                                 EPI_BIAS_BF16, L.stream_ptr()), "gemm")
    rows = torch.cat([torch.arange(0, 300), torch.arange(65280, 65792), torch.randint(0, M, (500,))]).cuda()
    want = a[rows].float() @ w.float().T + bias
    _close_bf16(c[rows], want)
    x0 = torch.randn(M, N, device="cuda", generator=g)
    x = x0.clone()
    L.check(lib.hmm_op_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), x.data_ptr(), M, N, K,
                                 EPI_BIAS_RESID_F32, L.stream_ptr()), "gemm")
    torch.testing.assert_close(x[rows], x0[rows] + want, rtol=2e-5, atol=3e-4)


@pytest.mark.parametrize("M", [256 * 50 + 31, 256 * 61])
def test_gemm_block_walk_covers_every_tile(M):
    """N = 5120 with >= 48 row tiles takes the 6 x 5 block walk (gemm_bf16.hip pick_walk): 50 / 61 row tiles leave a last
    row group of 2 / 1 tiles and M % 256 = 31 a ragged last tile.  Every output element must be written (the buffer starts
    as NaN) and equal the 128 x 128 tile kernel's, which walks plain rows."""
    L, lib = _lib()
    N, K = 5120, 256
    g = torch.Generator(device="cuda").manual_seed(M)
    a = torch.randn(M, K, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    bias = torch.randn(N, device="cuda", generator=g)
    got = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), got.data_ptr(), M, N, K, EPI_BIAS_BF16, 3,
                                      L.stream_ptr()), "gemm pp")
    ref = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_tile(a.data_ptr(), w.data_ptr(), bias.data_ptr(), ref.data_ptr(), M, N, K, EPI_BIAS_BF16, 0,
                                      L.stream_ptr()), "gemm 128")
    assert torch.isnan(got[M].float()).all() and not torch.isnan(got[:M].float()).any()
    assert torch.equal(got[:M], ref)


def test_gemm_identity_asymmetric():
    """A = I against an asymmetric W catches a transposed / permuted accumulator mapping exactly."""
    L, lib = _lib()
    K = N = 256
    a = torch.eye(K).to(torch.bfloat16)
    w = (torch.arange(N * K).reshape(N, K) % 251 - 125).float().to(torch.bfloat16)   # exact in bf16
    y = torch.empty(K, N, dtype=torch.float32, device="cuda")
    ad, wd = a.cuda(), w.cuda()
    for variant in (0, 1, 2, 3, 4):
        y.zero_()
        L.check(lib.hmm_op_gemm_bf16_tile(ad.data_ptr(), wd.data_ptr(), None, y.data_ptr(), K, N, K,
                                              EPI_F32, variant, L.stream_ptr()), "gemm")
        assert torch.equal(y.cpu(), w.float().T)


@pytest.mark.parametrize("rows,D", [(1, 768), (5, 1280), (1029, 1280), (700, 768), (333, 1024)])
def test_layernorm(rows, D):
    L, lib = _lib()
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.7
    gamma, beta = 1 + 0.2 * torch.randn(D, generator=g), 0.3 * torch.randn(D, generator=g)
    want = F.layer_norm(x, (D,), gamma, beta, 1e-6)
    y = torch.empty(rows, D, dtype=torch.bfloat16, device="cuda")
    xd, gd, bd = x.cuda(), gamma.cuda(), beta.cuda()
    L.check(lib.hmm_op_layernorm_bf16(xd.data_ptr(), gd.data_ptr(), bd.data_ptr(),
                                      y.data_ptr(), rows, D, 1e-6, L.stream_ptr()), "layernorm")
    _close_bf16(y, want, extra_atol=1e-5)


@pytest.mark.parametrize("M,N,K,splits", [(77, 1024, 4096, 2), (257, 1280, 5120, 2), (514, 1280, 5120, 2), (687, 768, 3072, 2),
                                           (77, 1024, 4096, 4), (300, 1280, 1280, 2), (40, 768, 3072, 8), (1030, 1280, 5120, 4),
                                           (462, 1024, 4096, 4), (539, 1024, 4096, 4), (693, 1024, 4096, 4)])      # six / seven / nine questions: the 128-row split tiles
def test_gemm_split_k_partials_and_the_layernorm_that_reduces_them(M, N, K, splits):
    """Few-row forwards run fc2 as a split-K launch: slab s = A[:, K_s] W[:, K_s]^T in fp32 (no bias), and the LayerNorm behind the
    GEMM adds the slabs in split order, the bias and the residual.  Checked: every slab against a torch fp32 product of the same
    bf16 operands; the slabs bitwise equal across every ring geometry and the choice by shape (an element's bits depend on
    (K, splits) only); x and LayerNorm(x) against the torch composition; rows past M untouched; and that with splits = 1 the
    pair is bit-identical to the residual epilogue followed by the plain LayerNorm kernel."""
    L, lib = _lib()
    g = torch.Generator().manual_seed(M + N + K + splits)
    a = _bf16(torch.randn(M, K, generator=g))
    w = _bf16(torch.randn(N, K, generator=g) * 0.05)
    bias, gamma, beta = torch.randn(N, generator=g), 1 + 0.2 * torch.randn(N, generator=g), 0.3 * torch.randn(N, generator=g)
    x0 = torch.randn(M, N, generator=g) * 2
    ad, wd, bd, gd, btd = a.cuda(), w.cuda(), bias.cuda(), gamma.cuda(), beta.cuda()
    kl = K // splits
    want_parts = torch.stack([a[:, s * kl:(s + 1) * kl].float() @ w[:, s * kl:(s + 1) * kl].float().T for s in range(splits)])
    slabs = {}
    for tile in (-1, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16):
        if tile in (9, 11) and kl % 128 or tile == 10 and kl % 256:
            continue
        part = torch.full((splits * M + 2, N), float("nan"), device="cuda")
        L.check(lib.hmm_op_gemm_bf16_splitk(ad.data_ptr(), wd.data_ptr(), part.data_ptr(), M, N, K, splits, tile, L.stream_ptr()), "splitk")
        assert torch.isnan(part[splits * M:]).all(), "rows past the last slab were written"
        slabs[tile] = part[:splits * M].reshape(splits, M, N)
    ref = slabs[-1]
    err = (ref.cpu() - want_parts).abs().max().item()
    assert err <= 2e-4 * max(1.0, want_parts.abs().max().item()), err
    for tile, p in slabs.items():
        assert torch.equal(p, ref), f"split-K slabs of tile geometry {tile} differ from the default choice"
    # the reducing LayerNorm
    x = torch.cat([x0, torch.full((1, N), float("nan"))]).cuda()
    y = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_layernorm_reduce_bf16(x.data_ptr(), ref.data_ptr(), splits, bd.data_ptr(), gd.data_ptr(), btd.data_ptr(),
                                             y.data_ptr(), M, N, 1e-6, L.stream_ptr()), "layernorm_reduce")
    acc = ref[0].clone()
    for s in range(1, splits):
        acc += ref[s]
    want_x = (acc + bd) + x0.cuda()                    # the kernel's own order: slabs, bias, residual
    assert torch.equal(x[:M], want_x) and torch.isnan(x[M:]).all()
    _close_bf16(y[:M], F.layer_norm(want_x.cpu(), (N,), gamma, beta, 1e-6), extra_atol=1e-5)
    assert torch.isnan(y[M:].float()).all()
    full = a.float() @ w.float().T + bias + x0        # and the value of the whole thing
    assert (x[:M].cpu() - full).abs().max().item() <= 3e-4 * max(1.0, full.abs().max().item())
    # one split == residual epilogue + LayerNorm kernel, bit for bit
    part1 = torch.empty(M, N, device="cuda")
    L.check(lib.hmm_op_gemm_bf16_splitk(ad.data_ptr(), wd.data_ptr(), part1.data_ptr(), M, N, K, 1, -1, L.stream_ptr()), "splitk")
    x1, y1 = x0.cuda(), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_layernorm_reduce_bf16(x1.data_ptr(), part1.data_ptr(), 1, bd.data_ptr(), gd.data_ptr(), btd.data_ptr(),
                                             y1.data_ptr(), M, N, 1e-6, L.stream_ptr()), "layernorm_reduce")
    x2, y2 = x0.cuda(), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), x2.data_ptr(), M, N, K, EPI_BIAS_RESID_F32, L.stream_ptr()), "gemm")
    L.check(lib.hmm_op_layernorm_bf16(x2.data_ptr(), gd.data_ptr(), btd.data_ptr(), y2.data_ptr(), M, N, 1e-6, L.stream_ptr()), "layernorm")
    assert torch.equal(x1, x2) and torch.equal(y1, y2)


def test_gemm_split_k_argument_errors():
    L, lib = _lib()
    a = torch.zeros(64, 1024, dtype=torch.bfloat16, device="cuda")
    w = torch.zeros(1024, 1024, dtype=torch.bfloat16, device="cuda")
    part = torch.zeros(8 * 64, 1024, device="cuda")
    st = L.stream_ptr()
    assert lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), 64, 1024, 1024, 3, -1, st) == -1    # 1024 / 3
    assert lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), 64, 1024, 1024, 9, -1, st) == -1
    assert lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), 64, 1024, 1024, 2, 3, st) == -1     # ping-pong tile: no split launch
    assert lib.hmm_op_gemm_bf16_splitk(a.data_ptr(), w.data_ptr(), part.data_ptr(), 64, 1024, 1024, 8, 10, st) == -1    # K/8 = 128: no K4 ring
    assert lib.hmm_op_gemm_bf16_splitk(None, w.data_ptr(), part.data_ptr(), 64, 1024, 1024, 2, -1, st) == -1
    assert lib.hmm_op_layernorm_reduce_bf16(part.data_ptr(), part.data_ptr(), 0, part.data_ptr(), part.data_ptr(), part.data_ptr(),
                                            a.data_ptr(), 64, 1024, 1e-6, st) == -1


def _attention_ref(qkv, B, T, H, dh, bias_k=None, bias_v=None):
    D = H * dh
    q, k, v = qkv.float().reshape(B, T, 3, H, dh).unbind(2)          # (B,T,H,dh)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))             # (B,H,T,dh)
    if bias_k is not None:
        bk = bias_k.to(torch.bfloat16).float().reshape(1, H, 1, dh).expand(B, -1, -1, -1)
        bv = bias_v.to(torch.bfloat16).float().reshape(1, H, 1, dh).expand(B, -1, -1, -1)
        k, v = torch.cat([k, bk], 2), torch.cat([v, bv], 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    o = torch.softmax(s, dim=-1) @ v
    return o.permute(0, 2, 1, 3).reshape(B * T, D)


@pytest.mark.parametrize("B,T,H,dh,bias,scale", [(2, 257, 16, 80, False, 1.0), (3, 229, 12, 64, True, 1.0),
                                                 (1, 257, 16, 80, False, 6.0), (2, 229, 12, 64, True, 6.0),
                                                 (1, 40, 2, 80, False, 2.0), (1, 33, 3, 64, True, 2.0),
                                                 (1, 1, 1, 64, False, 1.0)])
def test_attention(B, T, H, dh, bias, scale):
    """scale > 1 makes the softmax peaky so that masking / max / row-sum mistakes show."""
    L, lib = _lib()
    D = H * dh
    g = torch.Generator().manual_seed(B * 1000 + T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * scale).to(torch.bfloat16)
    bk = torch.randn(D, generator=g) * scale if bias else None
    bv = torch.randn(D, generator=g) * scale if bias else None
    want = _attention_ref(qkv, B, T, H, dh, bk, bv)
    out = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    qd = qkv.cuda()
    bkd, bvd = (bk.cuda(), bv.cuda()) if bias else (None, None)
    L.check(lib.hmm_op_attention_bf16(qd.data_ptr(), out.data_ptr(), B, T, H, dh,
                                      bkd.data_ptr() if bias else None,
                                      bvd.data_ptr() if bias else None, L.stream_ptr()), "attention")
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    # P is rounded to bf16 before P.V (as in every flash kernel): error <= 2^-9 * sum|p v| per element
    tol = 2.0 ** -8 * want.abs() + 2.0 ** -8 * scale + 1e-4
    bad = (got - want).abs() > tol
    assert not bad.any(), f"{int(bad.sum())}/{bad.numel()} off; worst {float((got - want).abs().max()):.4g}"


@pytest.mark.parametrize("B,T,H", [(3, 77, 16), (1, 33, 2), (2, 150, 4)])
def test_attention_causal(B, T, H):
    """Causal variant (text tower): query i sees keys 0..i."""
    L, lib = _lib()
    dh, D = 64, H * 64
    g = torch.Generator().manual_seed(T)
    qkv = (torch.randn(B * T, 3 * D, generator=g) * 3.0).to(torch.bfloat16)
    q, k, v = qkv.float().reshape(B, T, 3, H, dh).unbind(2)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh) + torch.full((T, T), float("-inf")).triu_(1)
    want = (torch.softmax(s, dim=-1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    out = torch.full((B * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    qd = qkv.cuda()
    L.check(lib.hmm_op_attention_causal_bf16(qd.data_ptr(), out.data_ptr(), B, T, H, dh, L.stream_ptr()), "attention")
    got = out.float().cpu()
    tol = 2.0 ** -8 * want.abs() + 2.0 ** -8 * 3.0 + 1e-4
    bad = (got - want).abs() > tol
    assert torch.isfinite(got).all() and not bad.any(), f"{int(bad.sum())}/{bad.numel()} off; worst {float((got - want).abs().max()):.4g}"


def test_attention_one_hot_rows_pick_the_right_value():
    """Exact structural check.  Key j lies on axis j % dh with length 1 + j // dh; a query of length
    300 along axis a therefore scores 300 * len / sqrt(dh) on that axis' keys and 0 elsewhere, so the
    longest key of the axis wins by > 30 nats and the output row must be exactly that key's V row.
    Covers the last (257th) key and the add_bias_kv position, which are the tops of their axes."""
    L, lib = _lib()
    for (T, H, dh, bias) in [(257, 16, 80, False), (229, 12, 64, True)]:
        D, Lk = H * dh, T + (1 if bias else 0)
        g = torch.Generator().manual_seed(T)
        k = torch.zeros(Lk, H, dh)
        for j in range(Lk):
            k[j, :, j % dh] = 1.0 + j // dh
        top = {j % dh: j for j in range(Lk)}                      # longest key per axis
        v = torch.randn(Lk, H, dh, generator=g).to(torch.bfloat16).float()
        axis = torch.arange(T) % dh
        axis[0] = (Lk - 1) % dh                                   # query 0 -> the very last key
        q = torch.zeros(T, H, dh)
        q[torch.arange(T), :, axis] = 300.0
        qkv = torch.zeros(T, 3, H, dh)
        qkv[:, 0], qkv[:, 1], qkv[:, 2] = q, k[:T], v[:T]
        bk = k[T].reshape(D).cuda() if bias else None
        bv = v[T].reshape(D).cuda() if bias else None
        out = torch.empty(T, D, dtype=torch.bfloat16, device="cuda")
        qd = qkv.reshape(T, 3 * D).to(torch.bfloat16).cuda()
        L.check(lib.hmm_op_attention_bf16(qd.data_ptr(), out.data_ptr(),
                                          1, T, H, dh, bk.data_ptr() if bias else None,
                                          bv.data_ptr() if bias else None, L.stream_ptr()), "attention")
        want = v[torch.tensor([top[int(a)] for a in axis])].reshape(T, D)
        assert torch.equal(out.float().cpu(), want)


@pytest.mark.parametrize("n_img", [1, 3, 9])
def test_fused_qkv_attention_equals_gemm_plus_attention(n_img):
    """hmm_op_qkv_attention_bf16 (vision tower: in_proj + attention in one kernel per (image, head)) against the two
    kernels it replaces on the same operands: BITWISE equal, and both close to the fp32 torch reference."""
    L, lib = _lib()
    T, D, H, dh = 257, 1280, 16, 80
    g = torch.Generator().manual_seed(100 + n_img)
    a = _bf16(torch.randn(n_img * T, D, generator=g))
    w = _bf16(torch.randn(3 * D, D, generator=g) * 0.03)
    bias = torch.randn(3 * D, generator=g) * 0.1
    ad, wd, bd = a.cuda(), w.cuda(), bias.cuda()
    qkv = torch.empty(n_img * T, 3 * D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), qkv.data_ptr(), n_img * T, 3 * D, D,
                                 EPI_BIAS_BF16, L.stream_ptr()), "gemm")
    two = torch.full((n_img * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_attention_bf16(qkv.data_ptr(), two.data_ptr(), n_img, T, H, dh, None, None, L.stream_ptr()), "attention")
    # fused: the cls rows are projected separately (M = n_img GEMM on the gathered rows)
    cls_rows = ad.view(n_img, T, D)[:, 0].contiguous()
    qkv_cls = torch.empty(n_img, 3 * D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16(cls_rows.data_ptr(), wd.data_ptr(), bd.data_ptr(), qkv_cls.data_ptr(), n_img, 3 * D, D,
                                 EPI_BIAS_BF16, L.stream_ptr()), "gemm cls")
    assert torch.equal(qkv_cls, qkv.view(n_img, T, 3 * D)[:, 0]), "small-M GEMM differs from the large one on the cls rows"
    one = torch.full((n_img * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_qkv_attention_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), qkv_cls.data_ptr(), one.data_ptr(),
                                          n_img, L.stream_ptr()), "qkv_attention")
    assert torch.isfinite(one.float()).all()
    assert torch.equal(one, two), f"max diff {(one.float() - two.float()).abs().max().item()}"
    # and against fp32 torch on the bf16-rounded qkv (the attention test's tolerance)
    q, k, v = qkv.float().cpu().reshape(n_img, T, 3, H, dh).unbind(2)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    want = (torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_img * T, D)
    tol = 2.0 ** -8 * want.abs() + 2.0 ** -8 * v.abs().max() + 1e-4
    assert not ((one.float().cpu() - want).abs() > tol).any()


@pytest.mark.parametrize("n_clips", [1, 5, 12])
def test_fused_qkv_attention_audio_equals_gemm_plus_attention(n_clips):
    """hmm_op_qkv_attention_audio_bf16 (audio tower: 229 tokens, 12 heads of 64, add_bias_kv; all rows of a clip inside the
    kernel's 256-row tile, the fourth wave-column idle) against the QKV GEMM + attention kernel it replaces: BITWISE equal,
    and close to fp32 torch on the bf16-rounded qkv with the bias_k / bias_v position appended."""
    L, lib = _lib()
    T, D, H, dh = 229, 768, 12, 64
    g = torch.Generator().manual_seed(300 + n_clips)
    a = _bf16(torch.randn(n_clips * T, D, generator=g))
    w = _bf16(torch.randn(3 * D, D, generator=g) * 0.04)
    bias = torch.randn(3 * D, generator=g) * 0.1
    bk, bv = torch.randn(D, generator=g) * 0.5, torch.randn(D, generator=g) * 0.5
    ad, wd, bd, bkd, bvd = a.cuda(), w.cuda(), bias.cuda(), bk.cuda(), bv.cuda()
    qkv = torch.empty(n_clips * T, 3 * D, dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_gemm_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), qkv.data_ptr(), n_clips * T, 3 * D, D,
                                 EPI_BIAS_BF16, L.stream_ptr()), "gemm")
    two = torch.full((n_clips * T, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_attention_bf16(qkv.data_ptr(), two.data_ptr(), n_clips, T, H, dh, bkd.data_ptr(), bvd.data_ptr(),
                                      L.stream_ptr()), "attention")
    one = torch.full((n_clips * T + 1, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    L.check(lib.hmm_op_qkv_attention_audio_bf16(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), bkd.data_ptr(), bvd.data_ptr(),
                                                one.data_ptr(), n_clips, L.stream_ptr()), "qkv_attention_audio")
    assert torch.isnan(one[-1].float()).all()                         # nothing past the last clip's rows is written
    one = one[:-1]
    assert torch.isfinite(one.float()).all()
    assert torch.equal(one, two), f"max diff {(one.float() - two.float()).abs().max().item()}"
    q, k, v = qkv.float().cpu().reshape(n_clips, T, 3, H, dh).unbind(2)
    q, k, v = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    k = torch.cat([k, _bf16(bk).float().reshape(1, H, 1, dh).expand(n_clips, H, 1, dh)], dim=2)
    v = torch.cat([v, _bf16(bv).float().reshape(1, H, 1, dh).expand(n_clips, H, 1, dh)], dim=2)
    want = (torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ v).permute(0, 2, 1, 3).reshape(n_clips * T, D)
    tol = 2.0 ** -8 * want.abs() + 2.0 ** -8 * v.abs().max() + 1e-4
    assert not ((one.float().cpu() - want).abs() > tol).any()
