"""GPU parity of the towers at the batch sizes that production and bench.py actually run.

From batch*clips >= 64 on, ``hmm_encoder_forward`` forks into two half-batches on two streams
(second workspace half, 256x256 ping-pong GEMM with the peeled last row tile, XCD-dealt attention
grid over many images).  The small-batch tests in test_gpu_encoder.py never reach that path, so
these do, at BASELINE cfg 2 (vision B=256) and cfg 3 (128 frame + audio pairs):

  * reduced depth vs the fp32 oracle on EVERY row (tolerance as in test_gpu_encoder.py);
  * full depth: bitwise equality with the same frames pushed through the single-stream small-batch
    path (max_batch=32 / 16), which IS oracle-checked row by row -- a per-frame result must not
    depend on the batch it rides in, the stream it ran on or the GEMM tile geometry;
  * odd splits (70 -> 35/35, 71 -> 35/36) and the single-stream A/B through hmm_encoder_set_streams.

Reference call shapes: foundation_models.py:116-133, hippocampal_memory.py:1328-1331 (32-frame buffer),
:1180-1183 (per segment)."""
import pytest
import torch

from oracle import imagebind_oracle as ib

pytestmark = pytest.mark.gpu

COS_TOL = 5e-5
ABS_TOL = 2e-3


def _check(got, want, scale=1.0, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    assert got.shape == want.shape and torch.isfinite(got).all()
    cos = torch.nn.functional.cosine_similarity(got, want, dim=1)
    err = (got - want).abs().max().item()
    print(f"{what}: rows {got.shape[0]}  min cos {cos.min().item():.7f}  max|diff| {err:.3e}")
    assert (1 - cos).max().item() <= COS_TOL
    assert err <= ABS_TOL * scale


def _frames(n, seed=0):
    return torch.randn(n, 3, 224, 224, generator=torch.Generator().manual_seed(seed))


def test_vision_depth2_batch256_vs_oracle_every_row():
    """BASELINE cfg 2 batch through the two-stream path; oracle on all 256 rows (~10 s of CPU)."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, 2)
    st = ib.synthetic_state(spec, seed=1234, init="rich")
    x = _frames(256)
    want = ib.vision_forward(x, st, spec)
    tower = HipTower("vision", st, depth=2)
    got = tower(x)
    _check(got, want, what="vision depth2 B=256 (two streams)")
    # the same rows through the single-chain small-batch path
    assert torch.equal(got, tower(x, max_batch=32)), "two-stream B=256 differs from single-stream chunks of 32"


def test_audio_depth2_batch128_vs_oracle_every_row():
    """BASELINE cfg 3 audio half: 128 segments = 384 clips, two streams of 64 segments."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.AUDIO_HUGE, 2)
    st = ib.synthetic_state(spec, seed=4321, init="rich")
    mels = torch.randn(128, 3, 1, 128, 204, generator=torch.Generator().manual_seed(1))
    want = ib.audio_forward(mels, st, spec)
    tower = HipTower("audio", st, depth=2)
    got = tower(mels)
    _check(got, want, scale=20.0, what="audio depth2 B=128 (two streams)")
    assert torch.equal(got, tower(mels, max_batch=16))


@pytest.mark.parametrize("batch", [64, 70, 71, 129])
def test_vision_odd_splits_bitwise_and_oracle(batch):
    """The smallest forked batch, even and odd splits, and one frame past a half-batch tile boundary."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, 1)
    st = ib.synthetic_state(spec, seed=9, init="rich")
    x = _frames(batch, seed=5)
    tower = HipTower("vision", st, depth=1)
    got = tower(x)
    assert torch.equal(got, tower(x, max_batch=8))
    pick = sorted({0, 1, batch // 2 - 1, batch // 2, batch - 1})          # both sides of the split and the ends
    _check(got[pick], ib.vision_forward(x[pick], st, spec), what=f"vision depth1 B={batch} rows {pick}")


def test_audio_odd_split():
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.AUDIO_HUGE, 1)
    st = ib.synthetic_state(spec, seed=11, init="rich")
    mels = torch.randn(23, 3, 1, 128, 204, generator=torch.Generator().manual_seed(2))   # 69 clips -> 11 / 12 segments
    tower = HipTower("audio", st, depth=1)
    got = tower(mels)
    assert torch.equal(got, tower(mels, max_batch=4))
    pick = [0, 10, 11, 22]
    _check(got[pick], ib.audio_forward(mels[pick], st, spec), scale=20.0, what="audio depth1 B=23")


def test_single_stream_option_is_bitwise_neutral():
    """hmm_encoder_set_streams(1) runs the same batch as one chain: identical embeddings."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, 2)
    st = ib.synthetic_state(spec, seed=3, init="rich")
    x = _frames(96, seed=7).cuda()
    tower = HipTower("vision", st, depth=2)
    two = tower(x)
    tower.set_streams(1)
    one = tower(x)
    tower.set_streams(2)
    assert torch.equal(one, two)
    assert torch.equal(two, tower(x))


@pytest.mark.parametrize("batch", [5, 70])
def test_fused_attention_option_is_bitwise_neutral(batch):
    """hmm_encoder_set_fused_attention(0) runs a QKV GEMM + the attention kernel instead of the fused kernel."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, 3)
    st = ib.synthetic_state(spec, seed=21, init="rich")
    x = _frames(batch, seed=8).cuda()
    tower = HipTower("vision", st, depth=3)
    fused = tower(x)
    tower.set_fused_attention(False)
    plain = tower(x)
    tower.set_fused_attention(True)
    assert torch.equal(fused, plain)
    _check(fused[:2], ib.vision_forward(x[:2].cpu(), st, spec), what=f"vision depth3 B={batch} fused attention")


@pytest.mark.parametrize("batch", [2, 24])
def test_fused_attention_option_is_bitwise_neutral_audio(batch):
    """Audio tower: in_proj + attention as one kernel per (clip, head) (default) against QKV GEMM + attention kernel."""
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.AUDIO_HUGE, 3)
    st = ib.synthetic_state(spec, seed=22, init="rich")
    mels = torch.randn(batch, 3, 1, 128, 204, generator=torch.Generator().manual_seed(9))
    tower = HipTower("audio", st, depth=3)
    fused = tower(mels)
    tower.set_fused_attention(False)
    plain = tower(mels)
    tower.set_fused_attention(True)
    assert torch.equal(fused, plain)
    _check(fused[:2], ib.audio_forward(mels[:2], st, spec), scale=20.0, what=f"audio depth3 B={batch} fused attention")


def test_full_depth_vision_batch256_bitwise_batch_invariance():
    """32 blocks at BASELINE cfg 2: the benchmarked configuration equals the oracle-checked small-batch path bit for bit,
    and its first frames match the fp32 oracle."""
    from hippomm_amd.encoder import HipTower
    st = ib.synthetic_state(ib.VISION_HUGE, seed=1234, init="survey")
    x = _frames(256)
    tower = HipTower("vision", st)
    big = tower(x)
    assert torch.equal(big, tower(x, max_batch=32))
    assert torch.equal(big, tower(x))                                       # run-to-run
    _check(big[:3], ib.vision_forward(x[:3], st), what="vision full depth B=256, rows 0..2")
    _check(big[253:], ib.vision_forward(x[253:], st), what="vision full depth B=256, rows 253..255")


def test_full_depth_audio_batch128_and_text_batch96_bitwise_batch_invariance():
    from hippomm_amd.encoder import HipTower
    st = ib.synthetic_state(ib.AUDIO_HUGE, seed=1235, init="survey")
    mels = torch.randn(128, 3, 1, 128, 204, generator=torch.Generator().manual_seed(1))
    tower = HipTower("audio", st)
    big = tower(mels)
    assert torch.equal(big, tower(mels, max_batch=16))
    _check(big[[0, 63, 64, 127]], ib.audio_forward(mels[[0, 63, 64, 127]], st), scale=20.0,
           what="audio full depth B=128")
    del tower
    st = ib.synthetic_state(ib.TEXT_HUGE, seed=5, init="survey")
    g = torch.Generator().manual_seed(0)
    tok = torch.randint(1, 49000, (96, 77), generator=g)
    for b in range(96):
        n = 1 + (b * 7) % 76
        tok[b, n] = 49407
        tok[b, n + 1:] = 0
    tower = HipTower("text", st)
    big = tower(tok)
    assert torch.equal(big, tower(tok, max_batch=16))
    _check(big[[0, 47, 48, 95]], ib.text_forward(tok[[0, 47, 48, 95]], st), scale=1.0 / 0.07,
           what="text full depth B=96")


def _towers_for_regimes():
    from hippomm_amd.encoder import HipTower
    spec = ib.reduced(ib.VISION_HUGE, 3)
    yield "vision", HipTower("vision", ib.synthetic_state(spec, seed=31, init="rich"), depth=3), _frames(52, seed=12), 1, 2
    spec = ib.reduced(ib.AUDIO_HUGE, 3)
    mels = torch.randn(9, 3, 1, 128, 204, generator=torch.Generator().manual_seed(13))
    yield "audio", HipTower("audio", ib.synthetic_state(spec, seed=32, init="rich"), depth=3), mels, 1, 2
    spec = ib.reduced(ib.TEXT_HUGE, 3)
    tok = torch.randint(1, 49000, (70, 77), generator=torch.Generator().manual_seed(14))
    for b in range(70):
        n = 1 + (b * 11) % 76
        tok[b, n] = 49407
        tok[b, n + 1:] = 0
    yield "text", HipTower("text", ib.synthetic_state(spec, seed=33, init="rich"), depth=3), tok, 9, 10


def test_a_sample_gets_the_same_bits_at_every_batch_size_within_a_regime():
    """The reference embeds one question, one query frame, one audio segment at a time (hippocampal_memory.py:1222, :2173, :2445)
    and buffers of up to 32 frames (:1328).  Two regimes (hmm_encoder_forward): few-row forwards -- one frame, one audio segment, up to
    nine questions (300 / 700 token rows) -- run fc2 as a deterministic split-K launch reduced inside the next
    LayerNorm; larger forwards keep the residual epilogue.  INSIDE a regime a sample's embedding does not depend on the batch
    it rides in, although the kernels change with the batch size (sliver GEMM / 32x32, 64x64, 128x128 rings / ping-pong,
    projection GEMM + attention kernel / fused kernel, one chain / two): bitwise, depth 3, every dispatch boundary crossed."""
    for name, tower, x, small_max, large_min in _towers_for_regimes():
        # few-row regime: one sample at a time == the largest batches that are still inside it
        one = tower(x, max_batch=1)
        if small_max > 1:
            assert torch.equal(one, tower(x, max_batch=small_max)), name
            assert torch.equal(one, tower(x, max_batch=max(2, small_max // 2))), name
        assert torch.equal(one, tower(x, max_batch=1)), name                  # run to run
        # large regime: the whole batch (two chains; vision: fused attention) == the smallest batches outside the few-row regime
        whole = tower(x)
        m = len(x) // large_min * large_min                                   # no short last chunk: it would be a few-row forward
        assert torch.equal(whole[:m], tower(x[:m], max_batch=large_min)), name
        n = {"vision": 17, "audio": 5, "text": 20}[name]                      # vision: two chains, projection GEMM + attention kernel
        assert torch.equal(whole[:n], tower(x[:n])), name
        del tower


def test_regimes_agree_within_the_stated_tolerance():
    """Across the regime boundary only the fp32 summation order of fc2 differs (two partial sums of K/2 products each, added in
    split order, instead of one walk over K): the embeddings agree to the tolerance every tower test states against the fp32
    oracle (cos >= 1 - 5e-5), and both regimes are oracle-checked themselves."""
    fwd = {"vision": ib.vision_forward, "audio": ib.audio_forward, "text": ib.text_forward}
    scale = {"vision": 1.0, "audio": 20.0, "text": 1.0 / 0.07}
    for name, tower, x, small_max, large_min in _towers_for_regimes():
        one, whole = tower(x, max_batch=1), tower(x)
        assert not torch.equal(one, whole), f"{name}: the few-row regime was not taken"
        _check(one, whole, scale=scale[name], what=f"{name} depth3 few-row regime vs large regime")
        spec = ib.reduced({"vision": ib.VISION_HUGE, "audio": ib.AUDIO_HUGE, "text": ib.TEXT_HUGE}[name], 3)
        st = ib.synthetic_state(spec, seed={"vision": 31, "audio": 32, "text": 33}[name], init="rich")
        want = fwd[name](x[:4], st, spec)
        _check(one[:4], want, scale=scale[name], what=f"{name} depth3 few-row regime vs oracle")
        _check(whole[:4], want, scale=scale[name], what=f"{name} depth3 large regime vs oracle")
        del tower


@pytest.mark.parametrize("batch,streams", [(1, 2), (6, 2), (70, 1), (70, 2)])     # 1 frame: the few-row regime (split-K fc2, depth 3)
def test_forward_is_graph_capturable(batch, streams):
    """include/hippomm_hip.h: every launch goes to the caller's stream and nothing synchronises or allocates, so a forward
    (including its internal fork / join onto the handle's own streams) can be captured into a HIP graph and replayed."""
    from hippomm_amd.encoder import HipTower
    depth = 3 if batch == 1 else 2                      # split-K needs a block behind the split one
    spec = ib.reduced(ib.VISION_HUGE, depth)
    st = ib.synthetic_state(spec, seed=13, init="rich")
    tower = HipTower("vision", st, depth=depth)
    tower.set_streams(streams)
    x = _frames(batch, seed=2).cuda()
    out = torch.empty(batch, 1024, device="cuda")
    eager = tower(x)                                   # also warms up (LDS attributes, workspace) outside the capture
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tower.forward_into(x, out)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tower.forward_into(x, out)
    out.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, eager)
    x.copy_(_frames(batch, seed=3).cuda())             # new input, same graph
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, tower(x))


@pytest.mark.parametrize("tower", ["vision", "audio", "text"])
def test_workspace_bytes_never_shrink_with_the_batch(tower):
    """hmm_encoder_workspace_bytes is non-decreasing in the batch (the few-sample forwards keep split-K slabs that the next larger
    batch does not), so a C caller that sizes its workspace once for its largest batch can run every smaller one."""
    from hippomm_amd.encoder import HipTower, synthetic_state_dict
    t = HipTower(tower, synthetic_state_dict((tower,), depth={tower: 1}), depth=1)
    sizes = [t._lib.hmm_encoder_workspace_bytes(t._h, b) for b in range(1, 70)]
    assert all(a <= b for a, b in zip(sizes, sizes[1:])), sizes[:16]
    big = torch.empty(sizes[-1], dtype=torch.uint8, device="cuda")
    from hippomm_amd.encoder import INPUT_SHAPE, INPUT_DTYPE
    for b in (1, 2, 9, 10, 11):
        x = torch.zeros(b, *INPUT_SHAPE[tower], dtype=INPUT_DTYPE[tower], device="cuda")
        if tower == "text":
            x[:, 0], x[:, 5] = 49406, 49407
        out = torch.empty(b, 1024, device="cuda")
        assert t._lib.hmm_encoder_forward(t._h, x.data_ptr(), b, out.data_ptr(), big.data_ptr(), big.numel(), None) == 0
    torch.cuda.synchronize()
