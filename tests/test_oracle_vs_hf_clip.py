"""Independent cross-check of the encoder oracle (oracle/imagebind_oracle.py, unpinned by the reference: the upstream
ImageBind package is not available).  ImageBind's vision and text towers are CLIP-architecture transformers (OpenCLIP
ViT-H/14 image tower behind a Conv3d patch stem, CLIP text tower), so HuggingFace `transformers`' CLIP implementation
-- written independently of this repository and of ImageBind -- computes the same function once its weights are
renamed to ImageBind's state-dict keys.  Random-init weights, 2 layers, full widths.

This does not pin the oracle to the reference (nothing here can); it shows that the restated architecture -- pre-LN
blocks, packed in-proj attention with 1/sqrt(d) scaling, exact-erf GELU MLP, pre-transformer LayerNorm, cls / EOS
pooling, LayerNorm -> bias-free projection heads, eps 1e-6 -- is the standard one and is evaluated correctly."""
import math

import pytest
import torch

from oracle import imagebind_oracle as ib

transformers = pytest.importorskip("transformers")


def _blocks(hf_sd, hf_prefix, ib_prefix, depth, out):
    for i in range(depth):
        h, b = f"{hf_prefix}.encoder.layers.{i}.", f"{ib_prefix}blocks.{i}."
        out[b + "norm_1.weight"], out[b + "norm_1.bias"] = hf_sd[h + "layer_norm1.weight"], hf_sd[h + "layer_norm1.bias"]
        out[b + "norm_2.weight"], out[b + "norm_2.bias"] = hf_sd[h + "layer_norm2.weight"], hf_sd[h + "layer_norm2.bias"]
        out[b + "attn.in_proj_weight"] = torch.cat([hf_sd[h + f"self_attn.{p}_proj.weight"] for p in "qkv"])
        out[b + "attn.in_proj_bias"] = torch.cat([hf_sd[h + f"self_attn.{p}_proj.bias"] for p in "qkv"])
        out[b + "attn.out_proj.weight"], out[b + "attn.out_proj.bias"] = hf_sd[h + "self_attn.out_proj.weight"], hf_sd[h + "self_attn.out_proj.bias"]
        for fc in ("fc1", "fc2"):
            out[b + f"mlp.{fc}.weight"], out[b + f"mlp.{fc}.bias"] = hf_sd[h + f"mlp.{fc}.weight"], hf_sd[h + f"mlp.{fc}.bias"]


def _randomise(model, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if "norm" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            elif name.endswith("bias") or "norm" in name:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.03 * torch.randn(p.shape, generator=g))


def test_vision_tower_matches_hf_clip_vision():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    depth = 2
    cfg = CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, projection_dim=1024, num_hidden_layers=depth,
                           num_attention_heads=16, image_size=224, patch_size=14, hidden_act="gelu", layer_norm_eps=1e-6,
                           attention_dropout=0.0)
    hf = CLIPVisionModelWithProjection(cfg).eval()
    _randomise(hf, 1)
    sd = hf.state_dict()
    pp, tr, hd = "modality_preprocessors.vision.", "modality_trunks.vision.", "modality_heads.vision."
    st = {}
    st[pp + "cls_token"] = sd["vision_model.embeddings.class_embedding"].reshape(1, 1, -1)
    st[pp + "pos_embedding_helper.pos_embed"] = sd["vision_model.embeddings.position_embedding.weight"][None]
    w2d = sd["vision_model.embeddings.patch_embedding.weight"]                      # (D,3,14,14)
    g = torch.Generator().manual_seed(2)
    split = torch.rand(w2d.shape, generator=g)                                       # any split over the 2 temporal taps
    st[pp + "rgbt_stem.proj.1.weight"] = torch.stack([w2d * split, w2d * (1 - split)], dim=2)   # PadIm2Video repeats the frame
    st[tr + "pre_transformer_layer.0.weight"] = sd["vision_model.pre_layrnorm.weight"]
    st[tr + "pre_transformer_layer.0.bias"] = sd["vision_model.pre_layrnorm.bias"]
    _blocks(sd, "vision_model", tr, depth, st)
    st[hd + "0.weight"], st[hd + "0.bias"] = sd["vision_model.post_layernorm.weight"], sd["vision_model.post_layernorm.bias"]
    st[hd + "2.weight"] = sd["visual_projection.weight"]
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(3))
    with torch.no_grad():
        want = torch.nn.functional.normalize(hf(pixel_values=x).image_embeds, dim=-1)
        got = ib.vision_forward(x, st, ib.reduced(ib.VISION_HUGE, depth))
    assert got.shape == want.shape == (2, 1024)
    assert (got - want).abs().max().item() < 2e-5
    assert torch.nn.functional.cosine_similarity(got, want).min().item() > 1 - 1e-6


def test_text_tower_matches_hf_clip_text():
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    depth = 2
    cfg = CLIPTextConfig(vocab_size=49408, hidden_size=1024, intermediate_size=4096, projection_dim=1024,
                         num_hidden_layers=depth, num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu",
                         layer_norm_eps=1e-6, attention_dropout=0.0, eos_token_id=49407, bos_token_id=49406, pad_token_id=0)
    hf = CLIPTextModelWithProjection(cfg).eval()
    _randomise(hf, 4)
    sd = hf.state_dict()
    pp, tr, hd = "modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text."
    st = {pp + "token_embedding.weight": sd["text_model.embeddings.token_embedding.weight"],
          pp + "pos_embed": sd["text_model.embeddings.position_embedding.weight"][None]}
    _blocks(sd, "text_model", tr, depth, st)
    st[hd + "proj.0.weight"], st[hd + "proj.0.bias"] = sd["text_model.final_layer_norm.weight"], sd["text_model.final_layer_norm.bias"]
    st[hd + "proj.1.weight"] = sd["text_projection.weight"]
    st["modality_postprocessors.text.1.log_logit_scale"] = torch.tensor([math.log(1 / 0.07)])
    g = torch.Generator().manual_seed(5)
    ids = torch.zeros(4, 77, dtype=torch.long)
    for r, n in enumerate([3, 20, 50, 75]):                        # [SOT] tokens [EOT] zero padding
        ids[r, 0] = 49406
        ids[r, 1:n] = torch.randint(1, 49000, (n - 1,), generator=g)
        ids[r, n] = 49407
    with torch.no_grad():
        want = torch.nn.functional.normalize(hf(input_ids=ids).text_embeds, dim=-1) * (1 / 0.07)
        got = ib.text_forward(ids, st, ib.reduced(ib.TEXT_HUGE, depth))
    assert got.shape == want.shape == (4, 1024)
    assert (got - want).abs().max().item() < 3e-4                  # rows have length 14.3
    assert torch.nn.functional.cosine_similarity(got, want).min().item() > 1 - 1e-6
