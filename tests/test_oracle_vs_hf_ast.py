"""Independent cross-check of the AUDIO side of the encoder oracle (oracle/imagebind_oracle.py, unpinned by the reference: the
upstream ImageBind package is not available).  ImageBind's audio tower is an Audio-Spectrogram-Transformer-shaped ViT-B:
Conv2d(1, 768, kernel 16, stride 10) over a (128 mel, 204 frame) image -> 12 x 19 = 228 patches, 12 heads, 3072-wide erf-GELU MLP,
pre-LN blocks.  HuggingFace `transformers`' AST implementation -- written independently of this repository and of ImageBind --
has the same patch stem and the same blocks, so two things can be checked against it with random weights:

* the patch grid: which 16 x 16 window each of the 228 tokens sees and the order they are flattened in (mel-major, then time);
* the trunk at the audio geometry (768 / 12 heads / 3072) on a 230-token sequence.

What AST cannot speak for, because it does not have them: the LayerNorm straight after the patch projection (applied here with
torch's own layer_norm on the HF patches), the single cls token (AST carries a second, distillation token), `add_bias_kv` (the
oracle calls torch's own `multi_head_attention_forward`, the function upstream's `nn.MultiheadAttention` subclass runs, with the
bias rows) and the x20 / three-clip mean of the head.  Not a pin -- nothing here can be -- but it is the audio counterpart of
tests/test_oracle_vs_hf_clip.py."""
import pytest
import torch
import torch.nn.functional as F

from oracle import imagebind_oracle as ib
from test_oracle_vs_hf_clip import _randomise

transformers = pytest.importorskip("transformers")

DEPTH = 2


def _ast(depth=DEPTH):
    from transformers import ASTConfig, ASTModel
    cfg = ASTConfig(hidden_size=768, num_hidden_layers=depth, num_attention_heads=12, intermediate_size=3072, hidden_act="gelu",
                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, layer_norm_eps=1e-6, patch_size=16,
                    frequency_stride=10, time_stride=10, max_length=204, num_mel_bins=128)
    hf = ASTModel(cfg).eval()
    _randomise(hf, 11)
    return hf


def _blocks_from_ast(sd, st, tr, depth=DEPTH):
    for i in range(depth):
        h, b = f"layers.{i}.", f"{tr}blocks.{i}."
        st[b + "norm_1.weight"], st[b + "norm_1.bias"] = sd[h + "layernorm_before.weight"], sd[h + "layernorm_before.bias"]
        st[b + "norm_2.weight"], st[b + "norm_2.bias"] = sd[h + "layernorm_after.weight"], sd[h + "layernorm_after.bias"]
        st[b + "attn.in_proj_weight"] = torch.cat([sd[h + f"attention.{p}_proj.weight"] for p in "qkv"])
        st[b + "attn.in_proj_bias"] = torch.cat([sd[h + f"attention.{p}_proj.bias"] for p in "qkv"])
        st[b + "attn.out_proj.weight"], st[b + "attn.out_proj.bias"] = sd[h + "attention.o_proj.weight"], sd[h + "attention.o_proj.bias"]
        for fc in ("fc1", "fc2"):
            st[b + f"mlp.{fc}.weight"], st[b + f"mlp.{fc}.bias"] = sd[h + f"mlp.{fc}.weight"], sd[h + f"mlp.{fc}.bias"]


def test_audio_patch_grid_matches_hf_ast():
    hf = _ast()
    sd = hf.state_dict()
    with torch.no_grad():
        hf.embeddings.patch_embeddings.projection.bias.zero_()     # ImageBind's stem has no bias
    pp = "modality_preprocessors.audio."
    g = torch.Generator().manual_seed(12)
    gamma, beta = 1.0 + 0.1 * torch.randn(768, generator=g), 0.05 * torch.randn(768, generator=g)
    st = {pp + "rgbt_stem.proj.weight": sd["embeddings.patch_embeddings.projection.weight"],
          pp + "rgbt_stem.norm_layer.weight": gamma, pp + "rgbt_stem.norm_layer.bias": beta,
          pp + "cls_token": torch.zeros(1, 1, 768),
          pp + "pos_embedding_helper.pos_embed": torch.zeros(1, 229, 768)}
    mel = torch.randn(3, 1, 128, 204, generator=g)                  # (clips, 1, mel bins, frames): what the tower is fed
    with torch.no_grad():
        patches = hf.embeddings.patch_embeddings(mel[:, 0].transpose(1, 2))   # AST takes (batch, frames, mel bins)
        want = F.layer_norm(patches, (768,), gamma, beta, 1e-5)
        got = ib.audio_tokens(mel, st, ib.AUDIO_HUGE)
    assert patches.shape == (3, 228, 768) and got.shape == (3, 229, 768)
    assert got[:, 0].abs().max().item() == 0.0                     # the (zero) cls row comes first
    assert (got[:, 1:] - want).abs().max().item() < 2e-5
    # the order matters: the same patches flattened time-major are a different sequence
    other = want.reshape(3, 12, 19, 768).transpose(1, 2).reshape(3, 228, 768)
    assert (got[:, 1:] - other).abs().max().item() > 0.1


@pytest.mark.parametrize("depth", [DEPTH, 12])          # 12 = imagebind_huge's audio depth: the whole trunk, not a 2-block sample
def test_audio_trunk_geometry_matches_hf_ast(depth):
    hf = _ast(depth)
    sd = hf.state_dict()
    tr, hd = "modality_trunks.audio.", "modality_heads.audio."
    st = {}
    _blocks_from_ast(sd, st, tr, depth)
    spec = ib.reduced(ib.AUDIO_HUGE, depth)
    x = torch.randn(2, 204, 128, generator=torch.Generator().manual_seed(13))
    with torch.no_grad():
        tokens = hf.embeddings(x)                                   # (2, 230, 768): cls, distillation, 228 patches, + positions
        want = hf(input_values=x).last_hidden_state                 # blocks, then the final LayerNorm
        got = ib._trunk_head(tokens, st, spec, return_tokens=True)  # no bias_k / bias_v in `st`: plain self-attention
        got = F.layer_norm(got, (768,), sd["layernorm.weight"], sd["layernorm.bias"], 1e-6)
    assert got.shape == want.shape == (2, 230, 768)
    assert (got - want).abs().max().item() < 5e-5
    assert F.cosine_similarity(got.flatten(1), want.flatten(1)).min().item() > 1 - 1e-6
