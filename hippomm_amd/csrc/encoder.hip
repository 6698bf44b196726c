// ImageBind-huge vision / audio tower forward on gfx950: parameter packing by UPSTREAM state-dict
// key, workspace plan and the kernel chain.  Replaces ImageBind._load_model / ImageBind.forward
// (reference hippomm/models/foundation_models.py:31-35, :116-133), i.e. upstream
// ImageBindModel.forward for the 'vision' and 'audio' keys (architecture restated in
// oracle/imagebind_oracle.py).
//
// Per block (all on the caller's stream, no host sync, graph-capturable):
//   LN1 (fp32 -> bf16)            layernorm_bf16_kernel
//   QKV   [R,D]x[3D,D]^T + b      gemm_bf16 (EPI_BIAS_BF16)
//   attention core                attention_kernel
//   out-proj + b + residual       gemm_bf16 (EPI_BIAS_RESID_F32, in place on the fp32 stream)
//   LN2                           layernorm_bf16_kernel
//   fc1 + b + GELU(erf)           gemm_bf16 (EPI_BIAS_GELU_BF16)
//   fc2 + b + residual            gemm_bf16 (EPI_BIAS_RESID_F32)
// HBM layout (workspace, R = images*tokens rows): residual stream x fp32 [R][D]; one bf16 [R][D]
// buffer shared by the LN output and the attention output; one bf16 [R][4D] buffer shared by qkv
// ([R][3D]) and the MLP hidden ([R][4D]) -- the patch im2col matrix and the patch-projection output
// alias it too, since they are dead before block 0.
#include <string>
#include <unordered_map>
#include <vector>
#include "hmm_common.h"
#include "encoder_ops.h"

namespace hmm {

HMM_TUNABLE(int, g_enc_side_priority, 0) // probe build: HIP priority of the second chain's stream (0 normal, 1 low, -1 high), read at create
HMM_TUNABLE(int, g_enc_split_min, 13)    // frames of a vision forward from which it runs as two chains on two streams (round 5, tools/mid_batch_probe.py, profiles/r5_mid_batch.log: 13 / 14 / 15 frames 7.39 / 7.92 / 8.02 -> 6.78 / 6.92 / 6.93 ms, 8 ... 12 frames faster as one chain, 16 ... 56 frames 0 ... -14 %)
HMM_TUNABLE(int, g_enc_split_min_text, 54)  // the same for the text tower: 54 questions = 4158 rows is where one chain's launches cross 16 row tiles of 256 (54 ... 62 questions 4.76 ... 5.28 -> 4.14 ... 4.50 ms as two chains; 24 ... 52 questions 8 ... 13 % slower as two; profiles/r5_split_min_text_ab.json)
HMM_TUNABLE(int, g_enc_two_chain_small_tiles, 64)  // gemm_set_small_tiles of a two-chain forward
HMM_TUNABLE(int, g_enc_split_min_audio, 12) // the audio tower's smaller kernels overlap from 4 segments on (-7 .. -11 %; tools/split_min_probe.py)
HMM_TUNABLE(int, g_enc_audio_one_round, 1) // see split_point
HMM_TUNABLE(int, g_enc_split_num, 128)   // frames of 256 that go to the first of the two chains
// Few-row forwards (one frame, one question, one audio segment -- the reference's own call sizes, hippocampal_memory.py:1222,
// :2173, :2445): fc2 walks K = 4D on few tiles, i.e. a chain of latencies on a mostly idle chip (one frame: 100 tiles of 64 x 64
// x 80 K-tiles each on 256 CUs, 22 us).  Up to g_enc_splitk_rows token rows per forward it runs split-K (gemm_bf16_splitk: fp32
// partial slabs, summed in split order by the LayerNorm that follows -- no launch added): one frame 2.45 -> 2.30 ms, one
// question 1.08 -> 1.00, one segment 0.73 -> 0.70 (profiles/r5_splitk_probe.json).  The split factor is a constant, so a
// sample gets the same bits at every batch size INSIDE this regime; across the regime boundary the fp32 sum order of fc2
// differs (embeddings within the stated tolerance; tests/test_gpu_encoder_batch.py has a bitwise test per regime and a
// tolerance test across).  out-proj (K = D: 16-20 K-tiles) gains nothing from a split (measured) and keeps its epilogue.
HMM_TUNABLE(int, g_enc_splitk_rows, 700)   // token rows (batch x clips x tokens) up to which a text / audio forward is in the split-K regime (one segment = 687 rows, nine questions = 693); 0 = never
HMM_TUNABLE(int, g_enc_splitk_rows_vision, 300)   // the same for the vision tower: one frame (two frames: 2.76 -> 2.84 ms with the split)
HMM_TUNABLE(int, g_enc_splitk_fc2, 2)      // K splits of fc2 (K = 4D), vision and audio towers (four: one frame 2.25 -> 2.27 ms, one segment 0.67 -> 0.69)
HMM_TUNABLE(int, g_enc_splitk_fc2_text, 4) // the same for the text tower (two / four splits: one question 0.988 / 0.992 ms, two 1.07 / 1.04, four 1.27 / 1.23; unsplit 1.04 / 1.10 / 1.37)
HMM_TUNABLE(int, g_enc_splitk_out, 1)      // K splits of out-proj (K = D); 1 = the residual epilogue as in every other regime

enum PackKind { PACK_F32, PACK_BF16, PACK_FOLD_CONV3D };

struct ParamSlot {
    size_t offset;      // bytes into the arena
    int64_t numel_src;  // elements the caller must provide
    PackKind kind;
    bool loaded;
};

struct BlockW {
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b, *qkv_b, *out_b, *fc1_b, *fc2_b, *bias_k, *bias_v;
    bf16_t *qkv_w, *out_w, *fc1_w, *fc2_w;
};

}  // namespace hmm

using namespace hmm;

struct hmm_encoder {
    int tower, D, depth, heads, mlp, n_patches, T, patch_k, patch_k_pad, clips;
    bool pre_ln, stem_ln, bias_kv, scaled;
    bool ready = false;
    int device = 0;                         // the device the handle was created on (weights, side stream, events)
    int streams = 2;                        // 2: half-batches on two streams (default), 1: one chain (hmm_encoder_set_streams)
    bool fused_attention = true;            // vision tower: in_proj + attention as one kernel (hmm_encoder_set_fused_attention)
    hipStream_t side_stream = nullptr;      // second half-batch runs here (see hmm_encoder_forward)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t cls_stream[2] = {nullptr, nullptr};   // per chain: the cls-row projection of the fused attention path
    hipEvent_t ev_x[2] = {nullptr, nullptr}, ev_cls[2] = {nullptr, nullptr};
    char* arena = nullptr;
    size_t arena_bytes = 0;
    std::unordered_map<std::string, ParamSlot> slots;
    // resolved pointers
    float *cls, *pos, *stem_g, *stem_b, *pre_g, *pre_b, *head_g, *head_b, *log_scale, *tok_emb;
    int vocab = 0;
    bf16_t *patch_w, *head_w;
    std::vector<BlockW> blocks;
};

namespace hmm {

struct ArenaBuilder {
    hmm_encoder* e;
    size_t cursor = 0;
    size_t add(const std::string& key, int64_t numel_src, PackKind kind, size_t bytes_dst) {
        const size_t off = cursor;
        e->slots[key] = ParamSlot{off, numel_src, kind, false};
        cursor = align_up(cursor + bytes_dst, 256);
        return off;
    }
};

static void plan_params(hmm_encoder* e, std::vector<std::pair<void**, size_t>>& fix) {
    ArenaBuilder ab{e};
    const std::string m = e->tower == HMM_TOWER_VISION ? "vision" : (e->tower == HMM_TOWER_AUDIO ? "audio" : "text");
    const std::string pp = "modality_preprocessors." + m + ".";
    const std::string tr = "modality_trunks." + m + ".";
    const std::string hd = "modality_heads." + m + ".";
    const int D = e->D;
    auto f32 = [&](const std::string& key, int64_t n, float** dst) {
        fix.push_back({reinterpret_cast<void**>(dst), ab.add(key, n, PACK_F32, (size_t)n * 4)});
    };
    auto b16 = [&](const std::string& key, int64_t n, bf16_t** dst) {
        fix.push_back({reinterpret_cast<void**>(dst), ab.add(key, n, PACK_BF16, (size_t)n * 2)});
    };
    if (e->tower == HMM_TOWER_TEXT) {
        f32(pp + "token_embedding.weight", (int64_t)e->vocab * D, &e->tok_emb);
        f32(pp + "pos_embed", (int64_t)e->T * D, &e->pos);
    } else {
        f32(pp + "cls_token", D, &e->cls);
        f32(pp + "pos_embedding_helper.pos_embed", (int64_t)e->T * D, &e->pos);
    }
    if (e->tower == HMM_TOWER_TEXT) {
    } else if (e->tower == HMM_TOWER_VISION) {
        fix.push_back({reinterpret_cast<void**>(&e->patch_w),
                       ab.add(pp + "rgbt_stem.proj.1.weight", (int64_t)D * 3 * 2 * 14 * 14, PACK_FOLD_CONV3D,
                              (size_t)D * e->patch_k_pad * 2)});
    } else {
        b16(pp + "rgbt_stem.proj.weight", (int64_t)D * 256, &e->patch_w);
        f32(pp + "rgbt_stem.norm_layer.weight", D, &e->stem_g);
        f32(pp + "rgbt_stem.norm_layer.bias", D, &e->stem_b);
    }
    if (e->pre_ln) {
        f32(tr + "pre_transformer_layer.0.weight", D, &e->pre_g);
        f32(tr + "pre_transformer_layer.0.bias", D, &e->pre_b);
    }
    e->blocks.resize(e->depth);
    for (int i = 0; i < e->depth; ++i) {
        const std::string b = tr + "blocks." + std::to_string(i) + ".";
        BlockW& w = e->blocks[i];
        f32(b + "norm_1.weight", D, &w.ln1_g);
        f32(b + "norm_1.bias", D, &w.ln1_b);
        b16(b + "attn.in_proj_weight", (int64_t)3 * D * D, &w.qkv_w);
        f32(b + "attn.in_proj_bias", 3 * D, &w.qkv_b);
        if (e->bias_kv) {
            f32(b + "attn.bias_k", D, &w.bias_k);
            f32(b + "attn.bias_v", D, &w.bias_v);
        } else {
            w.bias_k = w.bias_v = nullptr;
        }
        b16(b + "attn.out_proj.weight", (int64_t)D * D, &w.out_w);
        f32(b + "attn.out_proj.bias", D, &w.out_b);
        f32(b + "norm_2.weight", D, &w.ln2_g);
        f32(b + "norm_2.bias", D, &w.ln2_b);
        b16(b + "mlp.fc1.weight", (int64_t)e->mlp * D, &w.fc1_w);
        f32(b + "mlp.fc1.bias", e->mlp, &w.fc1_b);
        b16(b + "mlp.fc2.weight", (int64_t)D * e->mlp, &w.fc2_w);
        f32(b + "mlp.fc2.bias", D, &w.fc2_b);
    }
    if (e->tower == HMM_TOWER_TEXT) {
        f32(hd + "proj.0.weight", D, &e->head_g);
        f32(hd + "proj.0.bias", D, &e->head_b);
        b16(hd + "proj.1.weight", (int64_t)HMM_FEATURE_DIM * D, &e->head_w);
    } else {
        f32(hd + "0.weight", D, &e->head_g);
        f32(hd + "0.bias", D, &e->head_b);
        b16(hd + "2.weight", (int64_t)HMM_FEATURE_DIM * D, &e->head_w);
    }
    if (e->scaled) f32("modality_postprocessors." + m + ".1.log_logit_scale", 1, &e->log_scale);
    e->arena_bytes = ab.cursor;
}

struct WsPlan { int n_img, R; size_t off_x, off_a, off_big, off_im2col, off_patch, off_hl, off_hv, off_xc, off_ac, off_qc, off_hc, off_sel, off_part, total; };

// Whether a forward of `batch` samples is in the few-row split-K regime (a property of the whole forward, not of a chain).
// (A MID regime -- 8 ... 48 frames, fc2 as split-K on the ping-pong tile -- was built and measured in round 5: -2 ... -8 % at 8-28
// frames with a different best split factor per size, 0 at the reference's 32-frame buffer, losses above; under the 5 % bar it
// was removed: profiles/r5_mid_splitk_probe.json, profiles/LABNOTES_r5.md.)
static int fc2_splits(const hmm_encoder* e) { return e->tower == HMM_TOWER_TEXT ? g_enc_splitk_fc2_text : g_enc_splitk_fc2; }
static int splitk_mode(const hmm_encoder* e, int batch) {
    const int64_t rows = (int64_t)batch * e->clips * e->T;
    const int limit = e->tower == HMM_TOWER_VISION ? g_enc_splitk_rows_vision : g_enc_splitk_rows;
    return limit > 0 && rows <= limit && e->D % (64 * g_enc_splitk_out) == 0 && e->mlp % (64 * fc2_splits(e)) == 0 ? 1 : 0;
}
static int mode_splits(const hmm_encoder* e, int) { return fc2_splits(e); }

static WsPlan ws_plan(const hmm_encoder* e, int batch, int sk_mode) {
    WsPlan p{};
    p.n_img = batch * e->clips;
    p.R = p.n_img * e->T;
    const size_t D = e->D;
    size_t cur = 0;
    p.off_x = cur;   cur = align_up(cur + (size_t)p.R * D * 4, 256);
    p.off_a = cur;   cur = align_up(cur + (size_t)p.R * D * 2, 256);
    p.off_big = cur;
    const size_t big = (size_t)p.R * e->mlp * 2;
    p.off_im2col = p.off_big;
    const size_t im2col = align_up((size_t)p.n_img * e->n_patches * e->patch_k_pad * 2, 256);
    p.off_patch = p.off_big + im2col;
    const size_t pre = im2col + (size_t)p.n_img * e->n_patches * D * 4;
    cur = align_up(cur + (big > pre ? big : pre), 256);
    p.off_hl = cur;  cur = align_up(cur + (size_t)p.n_img * D * 2, 256);
    p.off_hv = cur;  cur = align_up(cur + (size_t)p.n_img * HMM_FEATURE_DIM * 4, 256);
    // cls-only last block: residual rows, two bf16 row buffers, MLP hidden
    p.off_xc = cur;  cur = align_up(cur + (size_t)p.n_img * D * 4, 256);
    p.off_ac = cur;  cur = align_up(cur + (size_t)p.n_img * D * 2, 256);
    p.off_qc = cur;  cur = align_up(cur + (size_t)p.n_img * D * 2, 256);
    p.off_hc = cur;  cur = align_up(cur + (size_t)p.n_img * e->mlp * 2, 256);
    p.off_sel = cur; cur = align_up(cur + (size_t)p.n_img * 4, 256);
    // split-K partial slabs [splits][R][D] fp32 (few-row forwards only)
    const int max_splits = mode_splits(e, sk_mode) > g_enc_splitk_out ? mode_splits(e, sk_mode) : g_enc_splitk_out;
    p.off_part = cur; cur = align_up(cur + (sk_mode ? (size_t)max_splits * p.R * D * 4 : 0), 256);
    p.total = cur + 256;
    return p;
}

}  // namespace hmm

extern "C" int hmm_encoder_create(hmm_encoder** out, int tower, int depth) {
    HMM_REQUIRE(out, HMM_E_INVALID, "encoder_create: null out");
    HMM_REQUIRE(tower == HMM_TOWER_VISION || tower == HMM_TOWER_AUDIO || tower == HMM_TOWER_TEXT, HMM_E_INVALID,
                "encoder_create: tower %d", tower);
    hmm_encoder* e = new hmm_encoder();
    e->tower = tower;
    if (tower == HMM_TOWER_VISION) {
        e->D = 1280; e->depth = 32; e->heads = 16; e->mlp = 5120; e->n_patches = 256;
        e->patch_k = 588; e->patch_k_pad = 640; e->clips = 1;
        e->pre_ln = true; e->stem_ln = false; e->bias_kv = false; e->scaled = false;
    } else if (tower == HMM_TOWER_AUDIO) {
        e->D = 768; e->depth = 12; e->heads = 12; e->mlp = 3072; e->n_patches = 228;
        e->patch_k = 256; e->patch_k_pad = 256; e->clips = 3;
        e->pre_ln = false; e->stem_ln = true; e->bias_kv = true; e->scaled = true;
    } else {                                           // CLIP-style text tower: 77 token positions, causal
        e->D = 1024; e->depth = 24; e->heads = 16; e->mlp = 4096; e->n_patches = 76;
        e->patch_k = 0; e->patch_k_pad = 64; e->clips = 1; e->vocab = 49408;
        e->pre_ln = false; e->stem_ln = false; e->bias_kv = false; e->scaled = true;
    }
    if (depth > 0) e->depth = depth;
    e->T = e->n_patches + 1;
    e->cls = e->pos = e->stem_g = e->stem_b = e->pre_g = e->pre_b = e->head_g = e->head_b = e->log_scale = nullptr;
    e->tok_emb = nullptr;
    e->patch_w = e->head_w = nullptr;
    std::vector<std::pair<void**, size_t>> fix;
    plan_params(e, fix);
    hipError_t err = hipGetDevice(&e->device);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&e->arena), e->arena_bytes);
    // the side stream and its fork/join events belong to the handle's device and exist before the first forward,
    // so that hmm_encoder_forward creates nothing (graph capture) and never lands them on another current device
    if (err == hipSuccess) err = g_enc_side_priority == 0 ? hipStreamCreateWithFlags(&e->side_stream, hipStreamNonBlocking)
                                                          : hipStreamCreateWithPriority(&e->side_stream, hipStreamNonBlocking, g_enc_side_priority);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming);
    for (int c = 0; c < 2 && err == hipSuccess; ++c) {
        err = hipStreamCreateWithFlags(&e->cls_stream[c], hipStreamNonBlocking);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_x[c], hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&e->ev_cls[c], hipEventDisableTiming);
    }
    if (err != hipSuccess) {
        set_error("encoder_create: device allocation (%zu B arena, side stream) failed: %s", e->arena_bytes,
                  hipGetErrorString(err));
        hmm_encoder_destroy(e);
        return HMM_E_HIP;
    }
    for (auto& f : fix) *f.first = e->arena + f.second;
    *out = e;
    return HMM_OK;
}

extern "C" void hmm_encoder_destroy(hmm_encoder* e) {
    if (!e) return;
    if (e->arena) (void)hipFree(e->arena);
    if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
    if (e->ev_join) (void)hipEventDestroy(e->ev_join);
    if (e->side_stream) (void)hipStreamDestroy(e->side_stream);
    for (int c = 0; c < 2; ++c) {
        if (e->ev_x[c]) (void)hipEventDestroy(e->ev_x[c]);
        if (e->ev_cls[c]) (void)hipEventDestroy(e->ev_cls[c]);
        if (e->cls_stream[c]) (void)hipStreamDestroy(e->cls_stream[c]);
    }
    delete e;
}

extern "C" int hmm_encoder_load_param(hmm_encoder* e, const char* key, const float* data_dev, int64_t numel,
                                      hmm_stream_t stream) {
    HMM_REQUIRE(e && key && data_dev, HMM_E_INVALID, "encoder_load_param: null argument");
    auto it = e->slots.find(key);
    HMM_REQUIRE(it != e->slots.end(), HMM_E_INVALID, "encoder_load_param: unexpected key '%s'", key);
    ParamSlot& s = it->second;
    HMM_REQUIRE(numel == s.numel_src, HMM_E_INVALID, "encoder_load_param: '%s' has %lld elements, expected %lld", key,
                (long long)numel, (long long)s.numel_src);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* dst = e->arena + s.offset;
    int rc = HMM_OK;
    switch (s.kind) {
        case PACK_F32:  rc = launch_copy_f32(data_dev, reinterpret_cast<float*>(dst), numel, st); break;
        case PACK_BF16: rc = launch_cast_bf16(data_dev, reinterpret_cast<bf16_t*>(dst), numel, st); break;
        case PACK_FOLD_CONV3D: rc = launch_fold_conv3d(data_dev, reinterpret_cast<bf16_t*>(dst), e->D, st); break;
    }
    if (rc != HMM_OK) return rc;
    s.loaded = true;
    return HMM_OK;
}

extern "C" int hmm_encoder_missing_params(hmm_encoder* e) {
    if (!e) return -1;
    int missing = 0;
    std::string names;
    for (auto& kv : e->slots)
        if (!kv.second.loaded) {
            if (missing < 8) names += (missing ? ", " : "") + kv.first;
            ++missing;
        }
    if (missing) set_error("%d parameter(s) not loaded: %s%s", missing, names.c_str(), missing > 8 ? ", ..." : "");
    return missing;
}

namespace hmm {
// Two half-batches on two streams: the HBM-bound kernels of one half (LayerNorm, K/V staging of the
// attention, residual read-modify-write epilogues) overlap the MFMA-bound GEMM tiles of the other on
// different CUs.  Measured -4.5 % on the ViT-H forward at batch 256; per-frame results are unchanged
// (every frame's rows go through the same kernels with the same K order).
static int split_point(const hmm_encoder* e, int batch) {
    const int split_min = e->tower == HMM_TOWER_AUDIO ? g_enc_split_min_audio : e->tower == HMM_TOWER_VISION ? g_enc_split_min : g_enc_split_min_text;
    if (e->streams < 2 || batch * e->clips < split_min) return 0;
    // audio, 5-7 segments (15-21 clips): the whole forward's fused in_proj + attention launch is ONE round of the chip (clips x 12
    // heads <= 256 workgroups), and one chain beats two half-rounds: 1.238 / 1.329 / 1.465 -> 1.200 / 1.251 / 1.285 ms
    // (profiles/r5_audio_chains_ab.json); at 4 segments two chains still win (1.143 vs 1.175), from 8 on the launch is two rounds
    if (e->tower == HMM_TOWER_AUDIO && g_enc_audio_one_round && batch * e->clips > split_min && batch * e->clips * e->heads <= kNumCU) return 0;
    const int b0 = (int)((long)batch * g_enc_split_num / 256);     // probe build: uneven halves (tile-round quantisation A/B)
    return b0 < 1 ? 1 : (b0 >= batch ? batch - 1 : b0);
}
}  // namespace hmm

static size_t workspace_bytes_exact(const hmm_encoder* e, int batch) {
    const int b0 = split_point(e, batch);
    const int sk = splitk_mode(e, batch);
    const size_t one = ws_plan(e, batch, sk).total;
    if (b0 == 0) return one;
    const size_t two = ws_plan(e, b0, sk).total + ws_plan(e, batch - b0, sk).total;
    return two > one ? two : one;            // a forward under stream capture runs as one chain (see hmm_encoder_forward)
}

// What a forward of `batch` samples needs, and never less than a smaller batch needs: the few-row regime adds fp32 split-K slabs
// (a text forward of 9 questions uses more than one of 10), so a caller that sizes its workspace once for its largest batch can
// run every smaller one.  The regime ends at a handful of samples; the loop below is a few iterations.
extern "C" size_t hmm_encoder_workspace_bytes(const hmm_encoder* e, int batch) {
    if (!e || batch < 1) return 0;
    size_t need = workspace_bytes_exact(e, batch);
    for (int b = 1; b < batch && splitk_mode(e, b); ++b) {
        const size_t s = workspace_bytes_exact(e, b);
        if (s > need) need = s;
    }
    return need;
}

extern "C" double hmm_encoder_flops(const hmm_encoder* e, int batch) {
    if (!e || batch < 1) return 0.0;
    const double D = e->D, T = e->T, H = e->mlp, Lk = e->T + (e->bias_kv ? 1 : 0);
    // un-folded patch projection, as the reference computes it (vision: both temporal taps)
    const double patch_k = e->tower == HMM_TOWER_VISION ? 2.0 * e->patch_k : e->patch_k;    // text: 0 (embedding lookup)
    double macs = e->n_patches * patch_k * D;                        // patch projection
    macs += e->depth * (T * D * (3 * D + D + 2 * H) + 2 * T * Lk * D);   // projections + MLP + QK^T + PV
    macs += D * HMM_FEATURE_DIM;                                     // head
    return 2.0 * macs * batch * e->clips;
}

extern "C" double hmm_encoder_flops_executed(const hmm_encoder* e, int batch) {
    if (!e || batch < 1) return 0.0;
    const double D = e->D, T = e->T, H = e->mlp, Lk = e->T + (e->bias_kv ? 1 : 0);
    const bool text = e->tower == HMM_TOWER_TEXT;
    double macs = text ? 0.0 : (double)e->n_patches * e->patch_k_pad * D;          // folded, K-padded patch projection
    const double full = T * D * (3 * D + D + 2 * H) + 2 * T * Lk * D;               // one whole block
    // last block of the vision / audio towers: K|V projection of every token, everything else for token 0 only
    const double last = T * D * 2 * D + D * (D + D + 2 * H) + 2 * Lk * D;
    macs += text ? e->depth * full : (e->depth - 1) * full + last;
    macs += D * HMM_FEATURE_DIM;                                                    // head
    return 2.0 * macs * batch * e->clips;
}

namespace hmm {
HMM_TUNABLE(int, g_enc_text_head_fused, 1) // text head: EOS select + row gather + LayerNorm as one launch (probe build: 0 = the three kernels, A/B)
HMM_TUNABLE(int, g_enc_cls_fork, 1)      // probe build: 0 = cls-row projection on the chain's own stream (A/B)
HMM_TUNABLE(int, g_enc_fused_min_vision, 48) // frames of a forward from which in_proj + attention run as the fused kernel (round 5, profiles/r5_fused_min_ab2.json: projection GEMM + attention kernel 32 / 36 / 40 / 44 frames -2.3 / -2.2 / -3.6 / -1.3 %, equal at 48, +1 ... +4.6 % from 56 on)
HMM_TUNABLE(int, g_enc_fused_min_audio, 9)   // clips (3 per segment) likewise: from three segments on (two segments 0.865 -> 0.839 ms unfused, three 0.936 vs 0.998; profiles/r5_fused_min_audio_ab.json)
HMM_TUNABLE(int, g_enc_mlp_chunk_rows, 0)  // probe build: > 0 = fc1 -> fc2 per chunk of this many token rows, the hidden activation of every chunk in the SAME buffer (round 5's energy experiment: keep it inside the 256-MB Infinity Cache; measured, not kept -- profiles/r5_mlp_chunk_probe.json)
HMM_TUNABLE(int, g_enc_sliver_rows, 16448) // token rows of a forward (batch x clips x tokens) up to which few-row GEMMs may use the sliver kernel

struct Chain {                 // one (half-)batch travelling through the tower on one stream
    const void* input; float* out; char* ws; WsPlan p; hipStream_t st; int batch;
    hipStream_t cls_st; hipEvent_t ev_x, ev_cls;      // fork for the cls-row projection of the fused attention path
    int tile;                  // GEMM dispatch of this forward: HMM_GEMM_TILE_AUTO, or AUTO_TILED when the forward is large
    bool fuse;                 // in_proj + attention as one kernel (large enough forwards only, see hmm_encoder_forward)
    int sk_mode;               // splitk_mode() of the forward: fc2 as a split-K launch reduced by the LayerNorm behind it
    mutable const float* pending_bias;   // fc2 of the previous block left partial slabs: the next LayerNorm adds them (+ this bias)
};
static float* chain_x(const Chain& c) { return reinterpret_cast<float*>(c.ws + c.p.off_x); }

#define HMM_TRY(call) do { int _rc = (call); if (_rc != HMM_OK) return _rc; } while (0)

static int chain_tokens(hmm_encoder* e, const Chain& c) {
    const WsPlan& p = c.p;
    float* x = reinterpret_cast<float*>(c.ws + p.off_x);
    bf16_t* im2col = reinterpret_cast<bf16_t*>(c.ws + p.off_im2col);
    float* patch = reinterpret_cast<float*>(c.ws + p.off_patch);
    const int D = e->D, n_img = p.n_img;
    if (e->tower == HMM_TOWER_TEXT) {                  // ids (B,77) int64 -> token embedding + positions; EOS rows
        const int64_t* ids = static_cast<const int64_t*>(c.input);
        HMM_TRY(launch_embed_tokens(ids, e->tok_emb, e->pos, x, n_img * e->T, e->T, e->vocab, c.st));
        if (!g_enc_text_head_fused) HMM_TRY(launch_select_eos(ids, reinterpret_cast<int32_t*>(c.ws + p.off_sel), n_img, e->T, c.st));
        return HMM_OK;
    }
    if (e->tower == HMM_TOWER_VISION) HMM_TRY(launch_im2col_vision(static_cast<const float*>(c.input), im2col, n_img, c.st));
    else                              HMM_TRY(launch_im2col_audio(static_cast<const float*>(c.input), im2col, n_img, c.st));
    HMM_TRY(gemm_bf16(im2col, e->patch_w, nullptr, patch, n_img * e->n_patches, D, e->patch_k_pad, HMM_EPI_F32, c.tile, c.st));
    HMM_TRY(launch_assemble_tokens(patch, e->cls, e->pos, e->stem_g, e->stem_b, 1e-5f, e->pre_g, e->pre_b, 1e-6f,
                                   x, n_img, e->T, D, c.st));
    return HMM_OK;
}

// fc1 + GELU then fc2 + residual on R rows.  Chunked (probe build only): the hidden activation of each chunk lives in the first
// `chunk` rows of `big`, so that fc2 reads what fc1 has just written from the Infinity Cache instead of HBM.
static int mlp_pair(hmm_encoder* e, const Chain& c, const BlockW& w, const bf16_t* a, bf16_t* big, float* x, int R) {
    const int D = e->D;
    const int chunk = g_enc_mlp_chunk_rows > 0 && g_enc_mlp_chunk_rows < R ? g_enc_mlp_chunk_rows : R;
    for (int r0 = 0; r0 < R; r0 += chunk) {
        const int rows = R - r0 < chunk ? R - r0 : chunk;
        int rc = gemm_bf16(a + (size_t)r0 * D, w.fc1_w, w.fc1_b, big, rows, e->mlp, D, HMM_EPI_BIAS_GELU_BF16, c.tile, c.st);
        if (rc != HMM_OK) return rc;
        rc = gemm_bf16(big, w.fc2_w, w.fc2_b, x + (size_t)r0 * D, rows, D, e->mlp, HMM_EPI_BIAS_RESID_F32, c.tile, c.st);
        if (rc != HMM_OK) return rc;
    }
    return HMM_OK;
}

static int chain_block(hmm_encoder* e, const Chain& c, int i) {
    const WsPlan& p = c.p;
    hipStream_t st = c.st;
    float* x = chain_x(c);
    bf16_t* a = reinterpret_cast<bf16_t*>(c.ws + p.off_a);
    bf16_t* big = reinterpret_cast<bf16_t*>(c.ws + p.off_big);
    float* xc = reinterpret_cast<float*>(c.ws + p.off_xc);
    bf16_t* ac = reinterpret_cast<bf16_t*>(c.ws + p.off_ac);
    bf16_t* qc = reinterpret_cast<bf16_t*>(c.ws + p.off_qc);
    bf16_t* hc = reinterpret_cast<bf16_t*>(c.ws + p.off_hc);
    const int D = e->D, T = e->T, R = p.R, n_img = p.n_img;
    const BlockW& w = e->blocks[i];
    const bool text = e->tower == HMM_TOWER_TEXT;     // the selected (EOS) row differs per sample: no cls-only shortcut
    const bool fused = e->fused_attention && c.fuse && e->tower == HMM_TOWER_VISION && D == 1280 && e->heads == 16 && T == 257;
    float* part = reinterpret_cast<float*>(c.ws + p.off_part);
    const size_t part_stride = (size_t)R * D;
    const int n_splits = mode_splits(e, c.sk_mode);
    const float* pending = c.pending_bias;            // fc2 of block i - 1 is still in its slabs: this block's norm_1 adds them
    c.pending_bias = nullptr;
    if (!(i + 1 < e->depth && fused)) {
        if (pending) HMM_TRY(launch_layernorm_reduce_bf16(x, part, part_stride, n_splits, pending, w.ln1_g, w.ln1_b, a, R, D, 1e-6f, st));
        else         HMM_TRY(launch_layernorm_bf16(x, (size_t)D, w.ln1_g, w.ln1_b, a, R, D, 1e-6f, st));
    }
    if (i + 1 < e->depth && fused) {
        // in_proj + attention in one kernel per (image, head): the packed qkv matrix never exists in HBM.  The cls rows
        // (one per image: they do not fit the kernel's 256-row tile) go through LayerNorm + a small GEMM of their own, on a
        // stream forked from the chain so that these two latency-bound launches run beside the big LayerNorm instead of in
        // front of the fused kernel (0.9 % of the forward); the attention output lands in `big` because every head of an
        // image still reads all of `a`.  Bitwise equal to the branch below.
        hipStream_t cst = g_enc_cls_fork ? c.cls_st : st;
        if (g_enc_cls_fork) {
            HMM_HIP_CHECK(hipEventRecord(c.ev_x, st));                       // x of this block is final on `st`
            HMM_HIP_CHECK(hipStreamWaitEvent(cst, c.ev_x, 0));
        }
        HMM_TRY(launch_layernorm_bf16(x, (size_t)T * D, w.ln1_g, w.ln1_b, ac, n_img, D, 1e-6f, cst));     // token 0 of every image
        HMM_TRY(gemm_bf16(ac, w.qkv_w, w.qkv_b, hc, n_img, 3 * D, D, HMM_EPI_BIAS_BF16, c.tile, cst));
        if (g_enc_cls_fork) HMM_HIP_CHECK(hipEventRecord(c.ev_cls, cst));
        HMM_TRY(launch_layernorm_bf16(x, (size_t)D, w.ln1_g, w.ln1_b, a, R, D, 1e-6f, st));
        if (g_enc_cls_fork) HMM_HIP_CHECK(hipStreamWaitEvent(st, c.ev_cls, 0));
        HMM_TRY(qkv_attention_bf16(a, w.qkv_w, w.qkv_b, hc, big, n_img, st));
        HMM_TRY(gemm_bf16(big, w.out_w, w.out_b, x, R, D, D, HMM_EPI_BIAS_RESID_F32, c.tile, st));
        HMM_TRY(launch_layernorm_bf16(x, (size_t)D, w.ln2_g, w.ln2_b, a, R, D, 1e-6f, st));
        HMM_TRY(mlp_pair(e, c, w, a, big, x, R));
    } else if (i + 1 < e->depth && e->fused_attention && c.fuse && e->tower == HMM_TOWER_AUDIO && D == 768 && e->heads == 12 && T == 229 &&
               e->bias_kv) {
        // audio: in_proj + attention in one kernel per (clip, head); every row of a clip fits the 256-row tile, so there is
        // no cls side path.  Bitwise equal to the branch below.
        HMM_TRY(qkv_attention_audio_bf16(a, w.qkv_w, w.qkv_b, w.bias_k, w.bias_v, big, n_img, st));
        HMM_TRY(gemm_bf16(big, w.out_w, w.out_b, x, R, D, D, HMM_EPI_BIAS_RESID_F32, c.tile, st));
        HMM_TRY(launch_layernorm_bf16(x, (size_t)D, w.ln2_g, w.ln2_b, a, R, D, 1e-6f, st));
        HMM_TRY(gemm_bf16(a, w.fc1_w, w.fc1_b, big, R, e->mlp, D, HMM_EPI_BIAS_GELU_BF16, c.tile, st));
        HMM_TRY(gemm_bf16(big, w.fc2_w, w.fc2_b, x, R, D, e->mlp, HMM_EPI_BIAS_RESID_F32, c.tile, st));
    } else if (i + 1 < e->depth || text) {
        HMM_TRY(gemm_bf16(a, w.qkv_w, w.qkv_b, big, R, 3 * D, D, HMM_EPI_BIAS_BF16, c.tile, st));
        HMM_TRY(attention_bf16(big, a, n_img, T, e->heads, D / e->heads, w.bias_k, w.bias_v, st, text));
        if (c.sk_mode == 1 && g_enc_splitk_out > 1) {
            HMM_TRY(gemm_bf16_splitk(a, w.out_w, part, R, D, D, g_enc_splitk_out, -1, st));
            HMM_TRY(launch_layernorm_reduce_bf16(x, part, part_stride, g_enc_splitk_out, w.out_b, w.ln2_g, w.ln2_b, a, R, D, 1e-6f, st));
        } else {
            HMM_TRY(gemm_bf16(a, w.out_w, w.out_b, x, R, D, D, HMM_EPI_BIAS_RESID_F32, c.tile, st));
            HMM_TRY(launch_layernorm_bf16(x, (size_t)D, w.ln2_g, w.ln2_b, a, R, D, 1e-6f, st));
        }
        HMM_TRY(gemm_bf16(a, w.fc1_w, w.fc1_b, big, R, e->mlp, D, HMM_EPI_BIAS_GELU_BF16, c.tile, st));
        if (c.sk_mode && n_splits > 1 && i + 1 < e->depth) {        // the next block's norm_1 reduces (text: the last block has none)
            HMM_TRY(gemm_bf16_splitk(big, w.fc2_w, part, R, D, e->mlp, n_splits, -1, st));
            c.pending_bias = w.fc2_b;
        } else {
            HMM_TRY(gemm_bf16(big, w.fc2_w, w.fc2_b, x, R, D, e->mlp, HMM_EPI_BIAS_RESID_F32, c.tile, st));
        }
    } else {
        // The head reads only token 0 (SelectElement(index=0)), so the LAST block needs K/V for every
        // token but Q, attention output, out-proj and the MLP for the cls row of each image only.
        // K|V projection of all rows (in_proj rows D..3D), Q projection of the cls rows (rows 0..D).
        HMM_TRY(gemm_bf16(a, w.qkv_w + (size_t)D * D, w.qkv_b + D, big, R, 2 * D, D, HMM_EPI_BIAS_BF16, c.tile, st));
        HMM_TRY(launch_gather_rows(a, (size_t)T * D * 2, ac, n_img, D * 2, st));
        HMM_TRY(gemm_bf16(ac, w.qkv_w, w.qkv_b, qc, n_img, D, D, HMM_EPI_BIAS_BF16, c.tile, st));
        HMM_TRY(attention_cls_bf16(qc, big, ac, n_img, T, e->heads, D / e->heads, w.bias_k, w.bias_v, st));
        HMM_TRY(launch_gather_rows(x, (size_t)T * D * 4, xc, n_img, D * 4, st));
        HMM_TRY(gemm_bf16(ac, w.out_w, w.out_b, xc, n_img, D, D, HMM_EPI_BIAS_RESID_F32, c.tile, st));
        HMM_TRY(launch_layernorm_bf16(xc, (size_t)D, w.ln2_g, w.ln2_b, ac, n_img, D, 1e-6f, st));
        HMM_TRY(gemm_bf16(ac, w.fc1_w, w.fc1_b, hc, n_img, e->mlp, D, HMM_EPI_BIAS_GELU_BF16, c.tile, st));
        HMM_TRY(gemm_bf16(hc, w.fc2_w, w.fc2_b, xc, n_img, D, e->mlp, HMM_EPI_BIAS_RESID_F32, c.tile, st));
    }
    return HMM_OK;
}

// head: LN on the cls rows -> Linear(D,1024) -> L2 normalise (x20, clip mean for audio)
static int chain_head(hmm_encoder* e, const Chain& c) {
    const WsPlan& p = c.p;
    float* xc = reinterpret_cast<float*>(c.ws + p.off_xc);
    bf16_t* hl = reinterpret_cast<bf16_t*>(c.ws + p.off_hl);
    float* hv = reinterpret_cast<float*>(c.ws + p.off_hv);
    if (e->tower == HMM_TOWER_TEXT && g_enc_text_head_fused) {
        // SelectEOSAndProject: the row at the EOS position of each sample, LayerNormed -- one launch instead of three
        HMM_TRY(launch_layernorm_eos_bf16(chain_x(c), static_cast<const int64_t*>(c.input), e->T, e->head_g, e->head_b, hl, p.n_img, e->D,
                                          1e-6f, c.st));
    } else {
        if (e->tower == HMM_TOWER_TEXT)
            HMM_TRY(launch_gather_selected_rows(chain_x(c), reinterpret_cast<const int32_t*>(c.ws + p.off_sel), e->T, xc,
                                                p.n_img, e->D * 4, c.st));
        HMM_TRY(launch_layernorm_bf16(xc, (size_t)e->D, e->head_g, e->head_b, hl, p.n_img, e->D, 1e-6f, c.st));
    }
    HMM_TRY(gemm_bf16(hl, e->head_w, nullptr, hv, p.n_img, HMM_FEATURE_DIM, e->D, HMM_EPI_F32, c.tile, c.st));
    HMM_TRY(launch_l2norm_rows(hv, c.out, c.batch, e->clips, e->log_scale, c.st));
    return HMM_OK;
}
#undef HMM_TRY

}  // namespace hmm

extern "C" int hmm_encoder_forward(hmm_encoder* e, const void* input_dev, int batch, float* out_dev,
                                   void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream) {
    HMM_REQUIRE(e && input_dev && out_dev && workspace_dev, HMM_E_INVALID, "encoder_forward: null argument");
    HMM_REQUIRE(batch >= 1, HMM_E_INVALID, "encoder_forward: batch=%d", batch);
    int cur_dev = -1;
    HMM_HIP_CHECK(hipGetDevice(&cur_dev));
    HMM_REQUIRE(cur_dev == e->device, HMM_E_STATE, "encoder_forward: handle lives on device %d, current device is %d",
                e->device, cur_dev);
    if (!e->ready) {
        HMM_REQUIRE(hmm_encoder_missing_params(e) == 0, HMM_E_STATE, "encoder_forward: %s", hmm_last_error());
        e->ready = true;
    }
    HMM_REQUIRE((int64_t)batch * e->clips * e->T < (1ll << 31) / 8, HMM_E_INVALID, "encoder_forward: batch too large");
    const size_t need = workspace_bytes_exact(e, batch);
    HMM_REQUIRE(workspace_bytes >= need, HMM_E_WORKSPACE, "encoder_forward: workspace %zu < required %zu",
                workspace_bytes, need);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char* ws = static_cast<char*>(workspace_dev);
    const size_t in_bytes_per_sample = e->tower == HMM_TOWER_VISION ? (size_t)3 * 224 * 224 * 4
                                       : e->tower == HMM_TOWER_AUDIO ? (size_t)3 * 128 * 204 * 4 : (size_t)e->T * 8;

    // Few-row GEMMs (cls rows, head, whole small batches) may go to the sliver kernel only when the forward is small: it is
    // the faster kernel on an otherwise idle chip (one question: 3.2 -> 1.3 ms), but next to the big GEMMs of a large batch
    // its unshared L2 reads cost them more than they save (batch 256: +1 %; tools/small_gemm_ab_probe.py).
    const int tile = (int64_t)batch * e->clips * e->T <= g_enc_sliver_rows ? HMM_GEMM_TILE_AUTO : HMM_GEMM_TILE_AUTO_TILED;
    // The fused in_proj + attention kernel walks one 256 x 256 x D tile per (sample, head): 16 (12) workgroups per frame (clip).
    // Below ~32 frames that leaves most CUs idle behind a long K walk, and the few-row projection GEMM (64 x 64 tiles behind the
    // ring) + the attention kernel are faster (1 frame: 3.85 -> 2.60 ms; tools/fused_small_probe.py).  Same bits either way.
    const bool fuse = e->tower != HMM_TOWER_TEXT &&
                      batch * e->clips >= (e->tower == HMM_TOWER_VISION ? g_enc_fused_min_vision : g_enc_fused_min_audio);
    const int sk = fuse ? 0 : splitk_mode(e, batch);  // (the few-row regime ends far below the fused one)
    Chain chains[2];
    int n_chains = 1;
    int b0 = split_point(e, batch);
    {
        // Under hipGraph stream capture the batch runs as ONE chain: capturing the two-chain fork (a forked stream that
        // forks again for its cls projection) crashes inside the ROCm 7.0 runtime.  Embeddings are bitwise the same.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        HMM_HIP_CHECK(hipStreamIsCapturing(st, &cap));
        if (cap != hipStreamCaptureStatusNone) b0 = 0;
    }
    if (b0 == 0) {
        chains[0] = Chain{input_dev, out_dev, ws, ws_plan(e, batch, sk), st, batch, e->cls_stream[0], e->ev_x[0], e->ev_cls[0], tile, fuse, sk, nullptr};
    } else {
        const WsPlan p0 = ws_plan(e, b0, sk);
        chains[0] = Chain{input_dev, out_dev, ws, p0, st, b0, e->cls_stream[0], e->ev_x[0], e->ev_cls[0], tile, fuse, sk, nullptr};
        chains[1] = Chain{static_cast<const char*>(input_dev) + (size_t)b0 * in_bytes_per_sample, out_dev + (size_t)b0 * HMM_FEATURE_DIM,
                          ws + p0.total, ws_plan(e, batch - b0, sk), e->side_stream, batch - b0,
                          e->cls_stream[1], e->ev_x[1], e->ev_cls[1], tile, fuse, sk, nullptr};
        n_chains = 2;
        HMM_HIP_CHECK(hipEventRecord(e->ev_fork, st));                       // fork
        HMM_HIP_CHECK(hipStreamWaitEvent(e->side_stream, e->ev_fork, 0));
    }
    // launches are interleaved block by block so that both streams always have work queued
    const int prev_small = gemm_set_small_tiles(n_chains == 2 ? g_enc_two_chain_small_tiles : 128);
    int rc = HMM_OK;
    for (int c = 0; c < n_chains && rc == HMM_OK; ++c) rc = chain_tokens(e, chains[c]);
    for (int i = 0; i < e->depth && rc == HMM_OK; ++i)
        for (int c = 0; c < n_chains && rc == HMM_OK; ++c) rc = chain_block(e, chains[c], i);
    for (int c = 0; c < n_chains && rc == HMM_OK; ++c) rc = chain_head(e, chains[c]);
    gemm_set_small_tiles(prev_small);
    if (rc != HMM_OK)                                                        // a failed chain may have left its cls fork un-joined
        for (int c = 0; c < 2; ++c) (void)hipStreamSynchronize(e->cls_stream[c]);
    if (n_chains == 2) {
        // join -- also when a launch failed mid-chain: whatever was queued on the side stream must be ordered before
        // the caller's stream continues (the caller may free or reuse the workspace right after an error return)
        const hipError_t j0 = hipEventRecord(e->ev_join, e->side_stream);
        const hipError_t j1 = j0 == hipSuccess ? hipStreamWaitEvent(st, e->ev_join, 0) : j0;
        if (j1 != hipSuccess) {
            (void)hipStreamSynchronize(e->side_stream);                      // last resort: block the host instead
            if (rc == HMM_OK) {
                set_error("encoder_forward: joining the side stream failed: %s", hipGetErrorString(j1));
                rc = HMM_E_HIP;
            }
        }
    }
    return rc;
}

extern "C" int hmm_encoder_set_fused_attention(hmm_encoder* e, int on) {
    HMM_REQUIRE(e, HMM_E_INVALID, "encoder_set_fused_attention: null handle");
    e->fused_attention = on != 0;
    return HMM_OK;
}

extern "C" int hmm_encoder_set_streams(hmm_encoder* e, int n_streams) {
    HMM_REQUIRE(e, HMM_E_INVALID, "encoder_set_streams: null handle");
    HMM_REQUIRE(n_streams == 1 || n_streams == 2, HMM_E_INVALID, "encoder_set_streams: %d (1 or 2)", n_streams);
    e->streams = n_streams;
    return HMM_OK;
}
