"""Oracle (test infrastructure, not product): fp32 ImageBind-huge vision + audio towers.

PARITY UNPINNED.  The reference only *calls* this arithmetic
(hippomm/models/foundation_models.py:33 ``imagebind_huge(pretrained=True)``,
:131 ``self.model(inputs)``); the code lives in the third-party package
``imagebind`` (github.com/facebookresearch/ImageBind, un-pinned: installed by
``git clone`` + ``pip install .``, reference README.md:35-48) which is absent
from /root/reference and from this image, as is the ``imagebind_huge.pth``
checkpoint.  The reference has no tests or golden vectors for it.  What follows
restates the published architecture (upstream files named per step) with
``torch.nn.functional`` fp32 CPU ops and keeps the upstream state-dict key
names, so a real checkpoint can be loaded into it later to validate it.
Independent cross-check (not a pin to the reference): tests/test_oracle_vs_hf_clip.py renames the weights of
HuggingFace ``transformers``' CLIP vision / text models -- the same architecture family, written independently --
to these keys and finds this file's ``vision_forward`` / ``text_forward`` equal to them to < 2e-5 on unit rows.
For the audio tower tests/test_oracle_vs_hf_ast.py does the same with ``transformers``' Audio Spectrogram Transformer for what the
two share (the Conv2d patch grid and its flattening order, the blocks at 768 / 12 heads / 3072); add_bias_kv attention, the
stem LayerNorm and the clip averaging have no such counterpart here.

Upstream structure restated here (imagebind/models/imagebind_model.py
``ImageBindModel`` with the ``imagebind_huge()`` overrides; transformer.py
``SimpleTransformer`` / ``BlockWithMasking`` / ``MultiheadAttention`` / ``Mlp``;
multimodal_preprocessors.py ``PadIm2Video`` / ``PatchEmbedGeneric`` /
``RGBDTPreprocessor`` / ``AudioPreprocessor``; helpers.py ``Normalize`` /
``LearnableLogitScaling`` / ``SelectElement``):

vision  (B,3,224,224)
  PadIm2Video(ntimes=2, "repeat") -> (B,3,2,224,224)
  Conv3d(3, 1280, k=s=(2,14,14), bias=False) -> 256 patches
  prepend cls_token, + pos_embed (1,257,1280)
  pre_transformer LayerNorm(1280, eps 1e-6)
  32 x { x += MHA(LN1(x)) ; x += fc2(GELU_erf(fc1(LN2(x)))) }   16 heads, MLP 5120
  head: LayerNorm(eps 1e-6) -> token 0 -> Linear(1280, 1024, bias=False)
  post: L2 normalise

audio   (B,3,1,128,204)  -> clips folded into the batch (3B,1,128,204)
  Conv2d(1, 768, k=16, s=10, bias=False) -> 12x19 = 228 patches
  stem LayerNorm(768) (torch default eps 1e-5)
  prepend cls_token, + pos_embed (1,229,768); no pre-transformer LayerNorm
  12 x block, 12 heads, MLP 3072, nn.MultiheadAttention(add_bias_kv=True):
      learned bias_k / bias_v appended as one extra key/value position
  head: LayerNorm(eps 1e-6) -> token 0 -> Linear(768, 1024, bias=False)
  post: L2 normalise, x exp(log_logit_scale)=20 (fixed), mean over the 3 clips

text    (B,77) int64 BPE token ids (CLIP vocabulary 49408; the tokenizer file is not in this image)
  token embedding + learned pos_embed (1,77,1024); causal attention mask
  24 x block, 16 heads, MLP 4096, LayerNorm eps 1e-6
  head: token at the EOS position (argmax of the ids) -> LayerNorm(eps 1e-6) -> Linear(1024,1024,bias=False)
  post: L2 normalise, x exp(log_logit_scale) (init 1/0.07, learnable, clipped at 100)
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace
from typing import Dict

import torch
import torch.nn.functional as F

OUT_DIM = 1024


@dataclass(frozen=True)
class TowerSpec:
    name: str            # 'vision' | 'audio'
    embed_dim: int
    depth: int
    heads: int
    mlp_dim: int
    n_patches: int       # tokens without cls
    patch_k: int         # im2col width of one patch (after folding for vision)
    pre_ln: bool         # LayerNorm before the blocks (vision)
    stem_ln: bool        # LayerNorm right after the patch projection (audio)
    bias_kv: bool        # add_bias_kv (audio)
    logit_scale: float   # post-normalise multiplier (audio: 20)
    clips: int           # clips averaged per sample (audio: 3)

    @property
    def tokens(self) -> int:
        return self.n_patches + 1

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.heads


VISION_HUGE = TowerSpec("vision", 1280, 32, 16, 5120, 256, 3 * 14 * 14, True, False, False, 1.0, 1)
AUDIO_HUGE = TowerSpec("audio", 768, 12, 12, 3072, 228, 16 * 16, False, True, True, 20.0, 3)
# text: n_patches + 1 = 77 token positions (no cls token; the EOS position is selected by the head)
TEXT_HUGE = TowerSpec("text", 1024, 24, 16, 4096, 76, 0, False, False, False, 1.0 / 0.07, 1)
TEXT_VOCAB = 49408


def reduced(spec: TowerSpec, depth: int) -> TowerSpec:
    """Same tower with fewer blocks (CI-sized fixtures)."""
    return replace(spec, depth=depth)


_FAST_INIT = False


def _tn(gen, shape, std):
    t = torch.empty(shape, dtype=torch.float32)
    if _FAST_INIT:                      # timing-only weights (bench cpu_baseline): plain normal
        return t.normal_(0.0, std, generator=gen)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t


def synthetic_state(spec: TowerSpec, seed: int = 1234, init: str = "survey",
                    w_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights under upstream key names.

    init='survey': Linear/Conv/pos/cls trunc_normal(std), biases 0, LN gamma 1 beta 0
                   (cfg 1/2).
    init='fast'  : as 'survey' with plain normal draws (quick to generate; timing-only).
    init='rich'  : additionally random biases, LN gamma/beta and bias_k/bias_v, so
                   that a dropped bias or affine term is visible in a parity test.
    """
    global _FAST_INIT
    g = torch.Generator().manual_seed(seed)
    rich = init == "rich"
    _FAST_INIT = init == "fast"
    D, m = spec.embed_dim, spec.name
    pp, tr, hd = f"modality_preprocessors.{m}.", f"modality_trunks.{m}.", f"modality_heads.{m}."
    st: Dict[str, torch.Tensor] = {}

    def vec(n, base, spread):
        if rich:
            return base + spread * torch.randn(n, generator=g)
        return torch.full((n,), float(base))

    if m == "text":
        st[pp + "token_embedding.weight"] = _tn(g, (TEXT_VOCAB, D), w_std)
        st[pp + "pos_embed"] = _tn(g, (1, spec.tokens, D), 0.01)
    else:
        st[pp + "cls_token"] = _tn(g, (1, 1, D), w_std)
        st[pp + "pos_embedding_helper.pos_embed"] = _tn(g, (1, spec.tokens, D), w_std)
    if m == "text":
        pass
    elif m == "vision":
        st[pp + "rgbt_stem.proj.1.weight"] = _tn(g, (D, 3, 2, 14, 14), w_std)
    else:
        st[pp + "rgbt_stem.proj.weight"] = _tn(g, (D, 1, 16, 16), w_std)
        st[pp + "rgbt_stem.norm_layer.weight"] = vec(D, 1.0, 0.1)
        st[pp + "rgbt_stem.norm_layer.bias"] = vec(D, 0.0, 0.1)
    if spec.pre_ln:
        st[tr + "pre_transformer_layer.0.weight"] = vec(D, 1.0, 0.1)
        st[tr + "pre_transformer_layer.0.bias"] = vec(D, 0.0, 0.1)
    for i in range(spec.depth):
        b = f"{tr}blocks.{i}."
        st[b + "norm_1.weight"] = vec(D, 1.0, 0.1)
        st[b + "norm_1.bias"] = vec(D, 0.0, 0.1)
        st[b + "attn.in_proj_weight"] = _tn(g, (3 * D, D), w_std)
        st[b + "attn.in_proj_bias"] = vec(3 * D, 0.0, 0.02)
        if spec.bias_kv:
            st[b + "attn.bias_k"] = _tn(g, (1, 1, D), w_std) if not rich else 0.5 * torch.randn(1, 1, D, generator=g)
            st[b + "attn.bias_v"] = _tn(g, (1, 1, D), w_std) if not rich else 0.5 * torch.randn(1, 1, D, generator=g)
        st[b + "attn.out_proj.weight"] = _tn(g, (D, D), w_std)
        st[b + "attn.out_proj.bias"] = vec(D, 0.0, 0.02)
        st[b + "norm_2.weight"] = vec(D, 1.0, 0.1)
        st[b + "norm_2.bias"] = vec(D, 0.0, 0.1)
        st[b + "mlp.fc1.weight"] = _tn(g, (spec.mlp_dim, D), w_std)
        st[b + "mlp.fc1.bias"] = vec(spec.mlp_dim, 0.0, 0.02)
        st[b + "mlp.fc2.weight"] = _tn(g, (D, spec.mlp_dim), w_std)
        st[b + "mlp.fc2.bias"] = vec(D, 0.0, 0.02)
    if m == "text":
        st[hd + "proj.0.weight"] = vec(D, 1.0, 0.1)
        st[hd + "proj.0.bias"] = vec(D, 0.0, 0.1)
        st[hd + "proj.1.weight"] = _tn(g, (OUT_DIM, D), w_std)
        st["modality_postprocessors.text.1.log_logit_scale"] = torch.tensor([math.log(spec.logit_scale)])
        return st
    st[hd + "0.weight"] = vec(D, 1.0, 0.1)
    st[hd + "0.bias"] = vec(D, 0.0, 0.1)
    st[hd + "2.weight"] = _tn(g, (OUT_DIM, D), w_std)
    if m == "audio":
        st["modality_postprocessors.audio.1.log_logit_scale"] = torch.tensor([math.log(spec.logit_scale)])
    return st


def _block(x_lbd, st, prefix, spec: TowerSpec, attn_mask=None):
    """One BlockWithMasking on (L,B,D) fp32, eval mode, no layer scale, no drop path."""
    D = spec.embed_dim
    y = F.layer_norm(x_lbd, (D,), st[prefix + "norm_1.weight"], st[prefix + "norm_1.bias"], 1e-6)
    attn, _ = F.multi_head_attention_forward(
        y, y, y, D, spec.heads,
        st[prefix + "attn.in_proj_weight"], st[prefix + "attn.in_proj_bias"],
        st.get(prefix + "attn.bias_k"), st.get(prefix + "attn.bias_v"),
        False, 0.0,
        st[prefix + "attn.out_proj.weight"], st[prefix + "attn.out_proj.bias"],
        training=False, need_weights=False, attn_mask=attn_mask)
    x_lbd = x_lbd + attn
    y = F.layer_norm(x_lbd, (D,), st[prefix + "norm_2.weight"], st[prefix + "norm_2.bias"], 1e-6)
    y = F.linear(y, st[prefix + "mlp.fc1.weight"], st[prefix + "mlp.fc1.bias"])
    y = F.gelu(y)  # exact erf form (nn.GELU default)
    y = F.linear(y, st[prefix + "mlp.fc2.weight"], st[prefix + "mlp.fc2.bias"])
    return x_lbd + y


def _trunk_head(tokens_bld, st, spec: TowerSpec, return_tokens=False):
    m, D = spec.name, spec.embed_dim
    tr, hd = f"modality_trunks.{m}.", f"modality_heads.{m}."
    x = tokens_bld
    if spec.pre_ln:
        x = F.layer_norm(x, (D,), st[tr + "pre_transformer_layer.0.weight"],
                         st[tr + "pre_transformer_layer.0.bias"], 1e-6)
    x = x.transpose(0, 1)  # b l d -> l b d
    for i in range(spec.depth):
        x = _block(x, st, f"{tr}blocks.{i}.", spec)
    x = x.transpose(0, 1)
    if return_tokens:
        return x
    y = F.layer_norm(x, (D,), st[hd + "0.weight"], st[hd + "0.bias"], 1e-6)
    y = y[:, 0]
    y = F.linear(y, st[hd + "2.weight"])
    return F.normalize(y, dim=-1)


def vision_tokens(frames: torch.Tensor, st, spec: TowerSpec = VISION_HUGE) -> torch.Tensor:
    """Patchify + cls + pos for (B,3,224,224) fp32 -> (B,257,D)."""
    pp = "modality_preprocessors.vision."
    video = frames.unsqueeze(2).repeat(1, 1, 2, 1, 1)              # PadIm2Video(repeat, 2)
    x = F.conv3d(video, st[pp + "rgbt_stem.proj.1.weight"], stride=(2, 14, 14))
    x = x.flatten(2).transpose(1, 2)                               # (B,256,D)
    cls = st[pp + "cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + st[pp + "pos_embedding_helper.pos_embed"]


def audio_tokens(clips: torch.Tensor, st, spec: TowerSpec = AUDIO_HUGE) -> torch.Tensor:
    """Patchify + stem LN + cls + pos for (N,1,128,204) fp32 -> (N,229,D)."""
    pp = "modality_preprocessors.audio."
    x = F.conv2d(clips, st[pp + "rgbt_stem.proj.weight"], stride=10)
    x = x.flatten(2).transpose(1, 2)                               # (N,228,D)
    x = F.layer_norm(x, (spec.embed_dim,), st[pp + "rgbt_stem.norm_layer.weight"],
                     st[pp + "rgbt_stem.norm_layer.bias"], 1e-5)
    cls = st[pp + "cls_token"].expand(x.shape[0], -1, -1)
    x = torch.cat([cls, x], dim=1)
    return x + st[pp + "pos_embedding_helper.pos_embed"]


@torch.no_grad()
def vision_forward(frames: torch.Tensor, st, spec: TowerSpec = VISION_HUGE) -> torch.Tensor:
    """(B,3,224,224) fp32 -> (B,1024) fp32 unit rows."""
    return _trunk_head(vision_tokens(frames.float(), st, spec), st, spec)


@torch.no_grad()
def audio_forward(mels: torch.Tensor, st, spec: TowerSpec = AUDIO_HUGE) -> torch.Tensor:
    """(B,3,1,128,204) fp32 -> (B,1024) fp32 = mean over clips of 20 x unit vectors."""
    B, S = mels.shape[:2]
    clips = mels.float().reshape(B * S, *mels.shape[2:])
    y = _trunk_head(audio_tokens(clips, st, spec), st, spec)
    scale = st["modality_postprocessors.audio.1.log_logit_scale"].exp().clamp(max=100.0)
    y = y * scale
    return y.reshape(B, S, -1).mean(dim=1)


@torch.no_grad()
def text_forward(tokens: torch.Tensor, st, spec: TowerSpec = TEXT_HUGE) -> torch.Tensor:
    """(B,77) int64 BPE token ids -> (B,1024) = exp(log_logit_scale) x unit rows.  [upstream, recalled]:
    TextPreprocessor (token embedding + learned positions, causal mask), 24 blocks, SelectEOSAndProject
    (the EOS token has the largest id, so its position is tokens.argmax(-1)): LayerNorm(eps 1e-6) ->
    Linear(1024,1024,bias=False); Normalize; LearnableLogitScaling(init 1/0.07, learnable, clipped at 100)."""
    pp, tr, hd = "modality_preprocessors.text.", "modality_trunks.text.", "modality_heads.text."
    D, L = spec.embed_dim, spec.tokens
    x = F.embedding(tokens, st[pp + "token_embedding.weight"]) + st[pp + "pos_embed"]
    mask = torch.full((L, L), float("-inf")).triu_(1)
    x = x.transpose(0, 1)
    for i in range(spec.depth):
        x = _block(x, st, f"{tr}blocks.{i}.", spec, attn_mask=mask)
    x = x.transpose(0, 1)
    eos = tokens.argmax(dim=-1)
    y = x[torch.arange(x.shape[0]), eos]
    y = F.layer_norm(y, (D,), st[hd + "proj.0.weight"], st[hd + "proj.0.bias"], 1e-6)
    y = F.linear(y, st[hd + "proj.1.weight"])
    y = F.normalize(y, dim=-1)
    return y * st["modality_postprocessors.text.1.log_logit_scale"].exp().clamp(max=100.0)


@torch.no_grad()
def forward(inputs: Dict[str, torch.Tensor], states: Dict[str, Dict[str, torch.Tensor]],
            specs: Dict[str, TowerSpec] | None = None) -> Dict[str, torch.Tensor]:
    """Mirror of ``ImageBindModel.forward`` for the two modalities on the hot path."""
    specs = specs or {"vision": VISION_HUGE, "audio": AUDIO_HUGE}
    out = {}
    for key, value in inputs.items():
        if key == "vision":
            out[key] = vision_forward(value, states[key], specs[key])
        elif key == "audio":
            out[key] = audio_forward(value, states[key], specs[key])
        elif key == "text":
            out[key] = text_forward(value, states[key], specs.get(key, TEXT_HUGE))
        else:
            raise KeyError(f"oracle covers 'vision', 'audio' and 'text' only, got {key!r}")
    return out
