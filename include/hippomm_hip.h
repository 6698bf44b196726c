/*
 * hippomm_hip.h -- C ABI of libhippomm_hip.so, the MI355X (gfx950) implementation of
 * HippoMM's perceptual-encoding + similarity hot path.
 *
 * The reference (linyueqian/HippoMM) is pure Python and has no FFI; its boundary for this
 * path is three Python call sites.  Each entry point below names the reference interface
 * it replaces (file:line relative to the reference repo).  The Python shims that keep the
 * reference signatures live in hippomm_amd/ (vector_ops.py, consolidation.py, encoder.py)
 * and bind these symbols with ctypes; INTEGRATION.md shows the reference-side patch.
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is a DEVICE pointer owned by the caller
 *     (PyTorch-ROCm allocates).  The library never frees or reallocates caller memory and
 *     never allocates outputs.  Only hmm_encoder_create() allocates (its own packed weights).
 *   - every launch goes to the caller's stream (hipStream_t passed as void*; hmm_encoder_forward also uses streams
 *     owned by the handle, forked from and joined to the caller's stream with events); no call synchronises the
 *     device or allocates, so all calls can be captured into a HIP graph (tests/test_gpu_encoder_batch.py).  A forward
 *     issued under stream capture runs as a single chain whatever hmm_encoder_set_streams says.
 *   - return value: 0 = ok, negative = error (HMM_E_*); hmm_last_error() returns a
 *     thread-local message for the last failing call on this thread.
 *   - single-threaded use per handle, as in the reference (all call sites are on the main
 *     thread of one process).  Never call from a fork()ed child.
 */
#ifndef HIPPOMM_HIP_H
#define HIPPOMM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HMM_OK            0
#define HMM_E_INVALID    -1   /* bad argument (shape, null pointer, unsupported size) */
#define HMM_E_WORKSPACE  -2   /* workspace too small */
#define HMM_E_HIP        -3   /* a HIP runtime call failed */
#define HMM_E_STATE      -4   /* handle not ready (weights missing) */

#define HMM_FEATURE_DIM  1024 /* width of every embedding / store row (reference shape guard:
                                 hippocampal_memory.py:484, :829, :1190, :3135) */

typedef void* hmm_stream_t;   /* hipStream_t */

/* Alignment: stores, queries and workspaces (*_dev) must be 16-byte aligned -- anything hipMalloc or a torch allocation returns is
 * 256-byte aligned; a pointer into the middle of such a buffer must keep the 16 bytes.  Refused with HMM_E_INVALID otherwise. */

int         hmm_abi_version(void);
const char* hmm_last_error(void);
/* HMM_OK when the current HIP device is what this library is built for (gfx950 with 256 compute units: the launch
 * geometries and the code objects assume it); HMM_E_STATE with a message otherwise.  The Python package calls it once per
 * device before the first kernel. */
int         hmm_device_supported(void);

/* ------------------------------------------------------------------------------------------
 * feature_search scan.  Replaces top_k_cosine_similarity(a, b, k)
 * (hippomm/utils/vector_ops.py:151-188; callers hippocampal_memory.py:3153, :3304).
 *
 *   sims[i] = dot(store[i], q) / (||store[i]|| * ||q||)      fp32, one pass over the store
 *   result  = the k' = min(k, n_rows) rows with the largest sims, best first.
 *   Order on ties / NaN: UNSPECIFIED in the reference -- it takes argsort(sims)[-k:][::-1] (vector_ops.py:185) and numpy's
 *   default introsort is unstable, so which of several equal (or NaN) similarities survives the cut depends on the input
 *   length and layout.  THIS library's rule is a total order: NaN (zero-norm row or query) ranks above every number; among
 *   equal sims the HIGHER row index comes first; -0.0 == +0.0.  It reproduces what the reference returns on the golden
 *   duplicate-row and zero-row cases (tests/golden/scan_golden.json) but not always: on live case 31 the reference
 *   returns zero-norm row 12 of {12, 55} where this rule returns 55 (tests/test_gpu_live_golden.py checks values and
 *   membership there, not the order inside a tie group).
 *
 *   store_dev  (n_rows, 1024) fp32 row-major, resident in HBM      query_dev (1024) fp32
 *   idx_out_dev int64[k'], sim_out_dev fp32[k'], n_out_dev int32[1] (= k')
 * ---------------------------------------------------------------------------------------- */
size_t hmm_cosine_topk_workspace_bytes(int64_t n_rows, int k);
int    hmm_cosine_topk(const float* store_dev, int64_t n_rows, int dim,
                       const float* query_dev, int k,
                       int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                       void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* Sharded scan: same scan, but emits the local top-k as packed 64-bit order keys
 * ((ordered sim bits << 32) | local row) so that ranks can all-gather 8*k bytes and merge with
 * hmm_topk_merge_keys, which adds each shard's row offset and applies the same total order. */
int    hmm_cosine_topk_keys(const float* store_dev, int64_t n_rows, int dim,
                            const float* query_dev, int k, uint64_t* keys_out_dev /* [k], 0-padded */,
                            void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);
int    hmm_topk_merge_keys(const uint64_t* keys_dev /* [n_shards][k] */, int n_shards, int k,
                           const int64_t* shard_row_offset_dev /* [n_shards] */,
                           int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                           hmm_stream_t stream);

/* feature_search through a bf16 SHADOW of the store (SURVEY 8d's optional shadow store; reported separately against
 * 2048 B per row).  hmm_shadow_store_build writes shadow row r = bf16(store[r] / ||store[r]||) (2048 B per row, NaN for a
 * zero-norm or non-finite row); hmm_cosine_topk_prefilter streams the shadow for approximate similarities, keeps every row that
 * can be among the k best under a proven error bound (|s~ - s| < 0.004: bf16 has an 8-bit significand), re-scores those rows on
 * the fp32 store with the arithmetic of hmm_cosine_topk and returns the k best: the SAME indices and the SAME fp32 similarities,
 * bit for bit, as hmm_cosine_topk (vector_ops.py:178-186 semantics, same total order).  When the candidate set is not provably
 * complete (a store of thousands of near-ties) the exact scan runs instead, inside the same call, decided on the device.
 * k > 64 or fewer than 16384 rows: the call IS hmm_cosine_topk.  stats_out_dev (may be null) int32[2]: candidates re-scored and
 * saturated block lists of the last call (-1, -1 when the prefilter was not used); a non-zero second entry or more than 1024
 * candidates means the exact scan answered.
 * Precondition of the bit identity: every row's squared norm is finite and non-zero in fp32 (unit-scale embeddings are).  A row
 * whose norm overflows or underflows gets a NaN shadow row, which ranks first here, while hmm_cosine_topk gives it 0 or +-inf.
 * The shadow is a snapshot of the rows it was built from: rebuild it after the rows change. */
size_t hmm_shadow_store_bytes(int64_t n_rows);
int    hmm_shadow_store_build(const float* store_dev, int64_t n_rows, int dim, void* shadow_dev, size_t shadow_bytes,
                              hmm_stream_t stream);
size_t hmm_cosine_topk_prefilter_workspace_bytes(int64_t n_rows, int k);
int    hmm_cosine_topk_prefilter(const float* store_dev, const void* shadow_dev, int64_t n_rows, int dim,
                                 const float* query_dev, int k, int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                 int32_t* stats_out_dev, void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* hmm_cosine_topk_segmented through the shadow: one pass over 2048 B per row for approximate similarities, per event the rows
 * that can be among its k best (same bound) re-scored on the fp32 store.  Same outputs as hmm_cosine_topk_segmented, bit for
 * bit (hippocampal_memory.py:3143-3153 semantics).  k > 64, or events of fewer than 128 rows on average: the call IS
 * hmm_cosine_topk_segmented. */
size_t hmm_cosine_topk_segmented_prefilter_workspace_bytes(int64_t n_rows, int n_segments, int k);
int    hmm_cosine_topk_segmented_prefilter(const float* store_dev, const void* shadow_dev, int64_t n_rows, int dim,
                                           const float* query_dev, const int64_t* seg_offsets_dev, int n_segments, int k,
                                           int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                           void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* Batched feature_search: the top-k of n_queries queries against the same store in ONE pass over it
 * (16 queries per pass; the Q x rows similarity block runs on the fp32 matrix cores, so a row is still read once from
 * HBM).  The reference calls top_k_cosine_similarity once per question (hippomm/utils/vector_ops.py:151-188 via
 * hippocampal_memory.py:3153, :3304).  queries_dev (n_queries,1024) fp32 row-major.  Outputs are row-major with row
 * stride k: idx_out[q*k ..], sim_out[q*k ..] (the first min(k, n_rows) entries of a row are valid), n_out[q].
 * Same similarity, total order and tie rule as hmm_cosine_topk (results can differ from it only where two similarities
 * agree to fp32 rounding: the dot products are summed in a different order).  k > 64 falls back to one scan per query. */
size_t hmm_cosine_topk_multi_workspace_bytes(int64_t n_rows, int n_queries, int k);
int    hmm_cosine_topk_multi(const float* store_dev, int64_t n_rows, int dim, const float* queries_dev, int n_queries,
                             int k, int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                             void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* Per-event feature_search in one pass.  Replaces the Python loop over events that calls
 * top_k_cosine_similarity(query, event.features[...], k=5) once per event
 * (hippomm/core/hippocampal_memory.py:3143-3153, :3294-3304).  store_dev is the concatenation of the
 * events' (n_e,1024) matrices, seg_offsets_dev int64[n_segments+1] their row offsets (non-decreasing,
 * last = n_rows).  Per event e: n_out[e] = min(k, n_e); idx_out[e*k ..] rows WITHIN the event (best first,
 * -1 padded), sim_out[e*k ..] (0 padded).  Same similarity and total order as hmm_cosine_topk. k <= 1024. */
size_t hmm_cosine_topk_segmented_workspace_bytes(int64_t n_rows, int n_segments, int k);
int    hmm_cosine_topk_segmented(const float* store_dev, int64_t n_rows, int dim, const float* query_dev,
                                 const int64_t* seg_offsets_dev, int n_segments, int k,
                                 int64_t* idx_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                                 void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* The global ranking behind that loop (hippocampal_memory.py:3275-3277: the hits of every event in one list, sorted by similarity
 * descending -- Python's stable sort: equal similarities stay in event order -- and the best ones kept).  Inputs: the three outputs of
 * hmm_cosine_topk_segmented[_prefilter].  Outputs: the best keep' = min(keep, number of hits) hits, best first: event index, row
 * within the event, similarity (-1 / -1 / 0 padded to `keep`), *n_out = keep'.  NaN similarities rank first.  keep <= 64. */
int    hmm_rank_segment_hits(const int64_t* idx_dev, const float* sims_dev, const int32_t* counts_dev, int n_segments, int k,
                             int keep, int64_t* event_out_dev, int64_t* row_out_dev, float* sim_out_dev, int32_t* n_out_dev,
                             hmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Consolidation similarity.  Replaces HippocampalMemory._select_key_frames(features, times,
 * similarity_threshold=0.9) (hippomm/core/hippocampal_memory.py:944-967; caller :855).
 *
 *   Fn = F / ||F||_rows (fp32), S = Fn Fn^T with every dot accumulated in fp64 (f64 MFMA) and
 *   rounded to fp32 once, keep 0, then keep i iff all S[i, kept] < (float)threshold.
 *   A NaN similarity blocks (as `nan < thr` is False in the reference).  n <= 2 keeps all.
 *
 *   features_dev (n, 1024) fp32 row-major   kept_out_dev int64[n] (first *n_kept valid)
 * ---------------------------------------------------------------------------------------- */
size_t hmm_gram_select_workspace_bytes(int n);
int    hmm_gram_select(const float* features_dev, int n, int dim, float threshold,
                       int64_t* kept_out_dev, int32_t* n_kept_out_dev,
                       void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Perceptual encoder.  Replaces ImageBind._load_model / ImageBind.forward
 * (hippomm/models/foundation_models.py:31-35, :116-133), i.e. upstream
 * imagebind_huge's vision and audio towers (un-vendored third-party code; see
 * oracle/imagebind_oracle.py for the restated architecture).
 *
 * One handle per tower.  bf16 MFMA GEMMs with fp32 accumulation, fp32 residual stream,
 * fp32 LayerNorm / softmax statistics.
 * ---------------------------------------------------------------------------------------- */
typedef struct hmm_encoder hmm_encoder;

#define HMM_TOWER_VISION 0   /* (B,3,224,224) fp32 -> (B,1024) unit rows                   */
#define HMM_TOWER_AUDIO  1   /* (B,3,1,128,204) fp32 -> (B,1024) = mean_3clips(20 * unit)  */
#define HMM_TOWER_TEXT   2   /* (B,77) int64 CLIP-BPE token ids -> (B,1024) = exp(log_logit_scale) * unit
                                (the query side of feature_search, hippocampal_memory.py:2173-2176) */

/* depth <= 0 selects the imagebind_huge depth (32 vision / 12 audio); a smaller depth builds
 * the same tower with fewer blocks (used by CI-sized parity fixtures). */
int  hmm_encoder_create(hmm_encoder** out, int tower, int depth);
void hmm_encoder_destroy(hmm_encoder* enc);

/* Upload one parameter by its UPSTREAM state-dict key (e.g.
 * "modality_trunks.vision.blocks.7.attn.in_proj_weight").  data_dev is a DEVICE fp32 tensor
 * with the upstream shape, contiguous; the library packs it (bf16 cast, Conv3d temporal-tap
 * fold, ...) into its own storage on `stream`.  Unknown keys / wrong sizes fail. */
int  hmm_encoder_load_param(hmm_encoder* enc, const char* upstream_key,
                            const float* data_dev, int64_t numel, hmm_stream_t stream);
/* Number of parameters still missing (0 = ready); names via hmm_last_error() when > 0. */
int  hmm_encoder_missing_params(hmm_encoder* enc);

/* Workspace of a forward of `batch` samples.  Non-decreasing in `batch`: a workspace sized for the largest batch a caller will
 * ever pass serves every smaller one (the few-sample forwards keep fp32 split-K slabs that slightly larger batches do not). */
size_t hmm_encoder_workspace_bytes(const hmm_encoder* enc, int batch);
/* input_dev: vision (batch,3,224,224) fp32 | audio (batch,3,1,128,204) fp32 | text (batch,77) int64
 * out_dev:   (batch,1024) fp32
 * The kernels are chosen by the size of the call (the reference calls with one question, one audio segment, the frames of a
 * segment or a 32-frame buffer: hippocampal_memory.py:1180, :1222, :1328, :2173).  A sample's embedding does not depend on the
 * batch it arrives in WITHIN a regime, bit for bit; there are two: few-row forwards (batch x clips x tokens <= 300 rows for the
 * vision tower = one frame, <= 700 for audio / text = one segment, up to nine questions), whose fc2 is a deterministic split-K
 * launch reduced inside the next LayerNorm, and everything larger.  Across the boundary the embeddings agree to 1 - cos ~ 1e-5
 * (stated tolerance 5e-5). */
int  hmm_encoder_forward(hmm_encoder* enc, const void* input_dev, int batch, float* out_dev,
                         void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);
/* FLOPs (2 x MAC) of one forward of `batch` samples as the REFERENCE computes it (un-folded patch convolution, every
 * block on every token: SURVEY 8d's 334.98 GFLOP per vision frame) -- the figure roofline fractions are quoted on. */
double hmm_encoder_flops(const hmm_encoder* enc, int batch);
/* FLOPs this build actually executes for the same forward: temporal taps of the Conv3d folded into one K-padded matrix,
 * and the last block computed for the selected row only (vision / audio: K,V for every token, the rest for token 0). */
double hmm_encoder_flops_executed(const hmm_encoder* enc, int batch);
/* n_streams = 2 (default): from 13 frames / 4 audio segments (except 5-7, whose fused attention launch is one round of the chip) / 54 questions on, a forward runs as two half-batches, the second on a stream owned
 * by the handle (forked from / joined to the caller's stream with events).  n_streams = 1: one chain on the caller's
 * stream only.  Embeddings are bitwise identical either way (tests/test_gpu_encoder_batch.py). */
int  hmm_encoder_set_streams(hmm_encoder* enc, int n_streams);
/* Vision and audio towers: on (default) = in_proj and the attention core of a block run as ONE kernel per (sample, head)
 * and the packed qkv matrix never goes through HBM; off = a QKV GEMM followed by the attention kernel.  Embeddings are
 * bitwise identical either way.  No effect on the text tower (77 tokens would fill 30 % of the kernel's 256-row tile).
 * Forwards below 48 frames / 9 clips (three segments) use the two-kernel path regardless: the fused kernel is 16 (12) workgroups
 * per sample. */
int  hmm_encoder_set_fused_attention(hmm_encoder* enc, int on);

/* ------------------------------------------------------------------------------------------
 * Device-side vision preprocessing.  Replaces, for already decoded frames, the transform chain of
 * imagebind.data.load_and_transform_vision_data (torchvision Resize(224, BICUBIC) -> CenterCrop(224) ->
 * ToTensor -> Normalize(CLIP mean/std) [upstream, recalled]) used at hippomm/models/foundation_models.py:87-90.
 * Bit-identical to Pillow's resize: two passes, 8-bit intermediate, 22-bit fixed-point coefficients computed by
 * the host (hippomm_amd/preprocess.py) for the 224 centre-cropped columns (kh/bh) and rows (kv/bv):
 *   k*_dev int32[224][ksize], b*_dev int32[224][2] = (first tap, tap count); rows [row_first,row_last) of the
 *   input are the ones the vertical taps touch.  frames_dev uint8 (batch,in_h,in_w,3) RGB; out (batch,3,224,224).
 * ---------------------------------------------------------------------------------------- */
size_t hmm_preprocess_vision_workspace_bytes(int batch, int rows_needed);
int    hmm_preprocess_vision_u8(const uint8_t* frames_dev, int batch, int in_h, int in_w,
                                const int32_t* kh_dev, const int32_t* bh_dev, int ksize_h,
                                const int32_t* kv_dev, const int32_t* bv_dev, int ksize_v,
                                int row_first, int row_last, float* out_dev,
                                void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* Host-side hand-over of one frame decoded by Pillow to the (pinned) upload buffer of hmm_preprocess_vision_u8 -- the step
 * between Image.open(path).convert("RGB") and the transform chain in imagebind.data.load_and_transform_vision_data [upstream,
 * recalled] (hippomm/models/foundation_models.py:87-90).  No GPU call.  Pillow stores RGB as 4 bytes per pixel (R,G,B,pad);
 * src_rgbx: n_pixels * 4 bytes; dst_rgb: n_pixels * 3 bytes, nothing beyond them is written.  The _arrow_ form takes the
 * `struct ArrowArray*` inside the "arrow_array" capsule of Image.__arrow_c_array__() (Arrow C data interface:
 * fixed_size_list<uint8>[4] with width * height entries; HMM_E_INVALID on any other layout) and copies the window
 * [y0, y0 + roi_h) x [x0, x0 + roi_w) as a dense (roi_h, roi_w, 3) block: only the pixels the centre crop's resampling taps
 * touch have to cross PCIe (57 % of a 1280x720 frame).  Called through ctypes these run without the interpreter lock, which
 * is what lets the decode threads of hippomm_amd/preprocess.py scale. */
int hmm_host_rgbx_to_rgb(const uint8_t* src_rgbx, size_t n_pixels, uint8_t* dst_rgb);
int hmm_host_arrow_rgbx_to_rgb(const void* arrow_array, int width, int height, int x0, int y0, int roi_w, int roi_h,
                               uint8_t* dst_rgb);

/* ------------------------------------------------------------------------------------------
 * Device-side audio front end.  Replaces waveform2melspec + Normalize inside
 * imagebind.data.load_and_transform_audio_data [upstream, recalled] as called at
 * hippomm/models/foundation_models.py:106-109: per clip, `waveform -= waveform.mean()`, then
 * torchaudio.compliance.kaldi.fbank(htk_compat=True, sample_frequency=16000, use_energy=False, window_type="hanning",
 * num_mel_bins=128, dither=0.0, frame_length=25, frame_shift=10), transposed to (128, frames), zero-padded / cut to
 * 204 frames, then (x - mean) / std.  clips_dev: n_clips mono fp32 clips at 16 kHz, clip c at clips_dev + c*clip_stride,
 * clip_len samples each (2 s = 32000 in the reference).  out_dev: (n_clips, 128, 204) fp32, i.e. the (B,3,1,128,204)
 * tensor the audio tower takes when n_clips = 3*B.  Clip selection (3 clips spread over the segment) is host logic.
 * window_dev (400 floats: torch.hann_window(400, periodic=False)) and mel_banks_dev (128 x 257 floats: kaldi
 * get_mel_banks(128, 512, 16000, 20, 0) plus one zero column) may be null, in which case the library generates them on
 * the device from the same formulas; pass tables computed on the host to reproduce torchaudio's weights to the bit.
 * ---------------------------------------------------------------------------------------- */
size_t hmm_audio_fbank_workspace_bytes(int n_clips);
int    hmm_audio_fbank(const float* clips_dev, int n_clips, int clip_len, int64_t clip_stride,
                       const float* window_dev, const float* mel_banks_dev,
                       float norm_mean, float norm_std, float* out_dev,
                       void* workspace_dev, size_t workspace_bytes, hmm_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Building blocks of the encoder, exported so that each kernel is parity-tested on its own
 * against a torch fp32 reference of the same op (tests/test_gpu_ops.py) and timed on its own
 * (bench.py roofline).  bf16 tensors are raw uint16 bit patterns in device memory.
 * ---------------------------------------------------------------------------------------- */
#define HMM_EPI_BIAS_BF16        0  /* C_bf16 = A W^T + bias                                  */
#define HMM_EPI_BIAS_GELU_BF16   1  /* C_bf16 = gelu_erf(A W^T + bias)                        */
#define HMM_EPI_BIAS_RESID_F32   2  /* C_f32 += A W^T + bias  (in-place residual)             */
#define HMM_EPI_F32              3  /* C_f32 = A W^T (+ bias if non-null)                     */

/* C[M,N] = A[M,K] (bf16 row-major, lda=K) x W[N,K]^T (bf16 row-major) ; K % 64 == 0, N % 128 == 0.
 * Rows of A beyond M are never read; rows of C beyond M are never written. */
int hmm_op_gemm_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                     void* c_dev, int M, int N, int K, int epilogue, hmm_stream_t stream);
/* The same product on one named tile geometry (hmm_op_gemm_bf16 picks per shape), so that every kernel the dispatcher
 * can choose is parity-tested on every shape.  Every geometry adds the K products of an output element in the same
 * order, so results are bitwise equal across geometries. */
#define HMM_GEMM_TILE_AUTO       -1  /* by shape; few rows may go to the sliver kernel                              */
#define HMM_GEMM_TILE_AUTO_TILED -2  /* by shape among the tiled kernels only (what a large-batch forward uses)       */
#define HMM_GEMM_TILE_128x128     0  /* 4 waves, double-buffered LDS-DMA                                  */
#define HMM_GEMM_TILE_256x128     1
#define HMM_GEMM_TILE_256x256     2  /* 8 waves, same loop                                                */
#define HMM_GEMM_TILE_256x256_PP  3  /* 8 waves, 4-phase ping-pong, counted vmcnt; needs N%256==0, K%128==0 */
#define HMM_GEMM_TILE_PP_PEELED   4  /* what AUTO uses for large shapes: PP on whole rounds + 128x128 tail */
#define HMM_GEMM_TILE_SLIVER      5  /* few rows: one wave per 16..64 x 16 sliver, operands from L2 straight into fragments; epilogues 0..3 */
#define HMM_GEMM_TILE_128x128_RING 6 /* 128x128 tiles behind a 4-deep LDS-DMA ring (counted vmcnt): launches of few tiles, peeled tails */
#define HMM_GEMM_TILE_64x64_RING  7  /* 64x64 tiles, 4 waves, behind the same ring: more workgroups for mid-size M with a long K */
#define HMM_GEMM_TILE_32x32_RING  8  /* 32x32 tiles, one 16x16 block per wave, same ring: a few dozen rows x a long K */
#define HMM_GEMM_TILE_32x32_RING_K2 9   /* 32x32 tiles, deep K: 2 K-tiles per ring stage, 4 stages (64 KiB: two workgroups per CU); K % 128 == 0 */
#define HMM_GEMM_TILE_64x64_RING_K2 11  /* 64x64 tiles, 2 K-tiles per stage, 3 stages (96 KiB, one workgroup per CU): launches of at most 256 such tiles; K % 128 == 0 */
#define HMM_GEMM_TILE_128x64_RING 12  /* 128 rows x 64 columns, 4 waves of 64 x 32, same ring (96 KiB: one workgroup per CU): few-row launches whose 64x64 tiles would exceed the CUs */
#define HMM_GEMM_TILE_64x128_RING 13  /* 64 rows x 128 columns, 4 waves of 32 x 64, same ring */
#define HMM_GEMM_TILE_128x128_RING8 14 /* 128x128 tiles behind the ring with EIGHT waves (64 x 32 each, two per SIMD) */
#define HMM_GEMM_TILE_128x64_RING8  15 /* 128x64 tiles, eight waves of 32 x 32 */
#define HMM_GEMM_TILE_64x128_RING8  16 /* 64x128 tiles, eight waves of 32 x 32 */
#define HMM_GEMM_TILE_32x32_RING_K4 10  /* 32x32 tiles, 4 K-tiles per stage, 4 stages (128 KiB): few rows x a long K (one question's fc2); K % 256 == 0 */
int hmm_op_gemm_bf16_tile(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                          void* c_dev, int M, int N, int K, int epilogue, int tile, hmm_stream_t stream);
/* Deterministic split-K (few-row forwards: one frame, one question, one audio segment; hippocampal_memory.py:1180, :1222,
 * :2173 call the encoder with such batches): part_dev[s][M][N] fp32 = A[:, s K/splits : (s+1) K/splits] W[:, same]^T for
 * s < splits, no bias; K % (64 splits) == 0; tile = one of the *_RING geometries or < 0 for a choice by shape.  An element's
 * bits depend on (K, splits) only -- every geometry walks a split's K range in the same order. */
int hmm_op_gemm_bf16_splitk(const uint16_t* a_dev, const uint16_t* w_dev, float* part_dev, int M, int N, int K,
                            int splits, int tile, hmm_stream_t stream);
/* The consumer of those slabs: x_f32[rows, D] = ((part[0] + part[1] + ... + part[splits-1]) + bias) + x (in place, the
 * slabs in split order) and y_bf16 = LayerNorm(x) * gamma + beta -- the residual epilogue of the GEMM and the LayerNorm
 * behind it in one launch.  part_dev [splits][rows][D] fp32; D in {768, 1024, 1280}. */
int hmm_op_layernorm_reduce_bf16(float* x_dev, const float* part_dev, int splits, const float* bias_dev,
                                 const float* gamma_dev, const float* beta_dev, uint16_t* y_dev, int rows, int D,
                                 float eps, hmm_stream_t stream);
/* y_bf16[rows, D] = LayerNorm(x_f32[rows, D]) * gamma + beta ; D in {768, 1024, 1280} */
int hmm_op_layernorm_bf16(const float* x_dev, const float* gamma_dev, const float* beta_dev,
                          uint16_t* y_dev, int rows, int D, float eps, hmm_stream_t stream);
/* Multi-head self-attention core on packed qkv (rows = batch*tokens, 3*D columns ordered
 * [q | k | v], heads contiguous inside each): out[rows, D] bf16.  bias_k/bias_v (D fp32, may be
 * null) append one extra key/value position (nn.MultiheadAttention add_bias_kv). */
int hmm_op_attention_bf16(const uint16_t* qkv_dev, uint16_t* out_dev, int batch, int tokens,
                          int heads, int head_dim, const float* bias_k_dev, const float* bias_v_dev,
                          hmm_stream_t stream);
/* Fused in_proj + attention of the vision tower (D = 1280, 16 heads of 80, 257 tokens per image): a_dev [n_img*257][1280]
 * bf16 (LayerNorm output), w_dev [3840][1280] bf16 + bias_dev [3840] (packed in_proj), qkv_cls_dev [n_img][3840] bf16 =
 * the projection of each image's token 0 (hmm_op_gemm_bf16 on the gathered cls rows), out_dev [n_img*257][1280] bf16.
 * Bitwise equal to hmm_op_gemm_bf16(HMM_EPI_BIAS_BF16) + hmm_op_attention_bf16. */
int hmm_op_qkv_attention_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                              const uint16_t* qkv_cls_dev, uint16_t* out_dev, int n_img, hmm_stream_t stream);
/* The same fusion for the audio tower (D = 768, 12 heads of 64, 229 tokens per clip, nn.MultiheadAttention add_bias_kv):
 * a_dev [n_clips*229][768] bf16, w_dev [2304][768] bf16 + bias_dev [2304], bias_k_dev / bias_v_dev [768] fp32 (the learned
 * extra key / value position), out_dev [n_clips*229][768] bf16.  All 229 rows of a clip sit inside the kernel's 256-row
 * tile, so there is no separate cls projection.  Bitwise equal to hmm_op_gemm_bf16(HMM_EPI_BIAS_BF16) + hmm_op_attention_bf16
 * with the same bias_k / bias_v. */
int hmm_op_qkv_attention_audio_bf16(const uint16_t* a_dev, const uint16_t* w_dev, const float* bias_dev,
                                    const float* bias_k_dev, const float* bias_v_dev, uint16_t* out_dev, int n_clips,
                                    hmm_stream_t stream);
/* Causal variant (text tower): key j is visible to query i iff j <= i; no bias_kv. */
int hmm_op_attention_causal_bf16(const uint16_t* qkv_dev, uint16_t* out_dev, int batch, int tokens,
                                 int heads, int head_dim, hmm_stream_t stream);

/* Timing / test hooks of the scan (bench.py roofline, tests): the streaming kernel of hmm_cosine_topk alone --
 * scan_topk_kernel writing its per-block candidate keys (cand_keys_dev: 2048*k uint64) -- and the plain similarity
 * pass (sims_dev: n_rows fp32) that k > 128 uses. */
int hmm_op_scan_topk_only(const float* store_dev, int64_t n_rows, const float* query_dev, int k,
                          uint64_t* cand_keys_dev, hmm_stream_t stream);
int hmm_op_scan_sims(const float* store_dev, int64_t n_rows, const float* query_dev, float* sims_dev,
                     hmm_stream_t stream);

/* ---- memory_store event files: host-side parsing of the feature matrices (no GPU call) ----
 * Replaces, for the 2-D arrays of numbers only, the json.load + np.array(list) of load_theta_event
 * (hippomm/core/hippocampal_memory.py:369-395): hmm_json_find_matrices reports every `[[numbers], ...]` with equally long rows and
 * at least min_values numbers in text[0, len) outside string literals (byte span of the outer brackets + shape; at most `cap`
 * entries written, *n_found counts all); hmm_json_parse_matrix_f32 converts one reported span into rows x cols fp32,
 * out[r][c] = (float)(double)literal with correctly rounded, locale-independent decimal -> double conversion -- the value
 * np.array(json.load(f)[...]).astype(float32) has.  NaN / Infinity / -Infinity (Python's json spelling) are numbers here.  A
 * literal outside the double range fails with HMM_E_INVALID: the caller keeps json.load for that file. */
typedef struct hmm_json_matrix { size_t begin, end, rows, cols; } hmm_json_matrix;
int hmm_json_find_matrices(const char* text, size_t len, size_t min_values, hmm_json_matrix* out, int cap, int* n_found);
int hmm_json_parse_matrix_f32(const char* text, size_t begin, size_t end, size_t rows, size_t cols, float* out_host);
/* The writer's side (save_theta_event, hippocampal_memory.py:331-335: json.dump(event.to_dict(), f, indent=2)): rows x cols host
 * doubles as the text json.dumps(m.tolist(), indent=2) has for that list when its closing bracket sits at close_indent spaces
 * (rows at +2, values at +4) -- float.__repr__ digits and notation, NaN / Infinity / -Infinity as json spells them; byte for byte.
 * out_text needs hmm_json_matrix_text_bound(rows, cols, close_indent) bytes; *written = the length (no terminator). */
size_t hmm_json_matrix_text_bound(size_t rows, size_t cols, int close_indent);
int hmm_json_write_matrix_f64(const double* m_host, size_t rows, size_t cols, int close_indent, char* out_text, size_t cap, size_t* written);

#ifdef __cplusplus
}
#endif
#endif /* HIPPOMM_HIP_H */
