// TEMPORARY: replaced by encoder.hip in the next commit.
#include "hmm_common.h"
extern "C" int hmm_encoder_create(hmm_encoder**, int, int) { hmm::set_error("encoder not built yet"); return HMM_E_STATE; }
extern "C" void hmm_encoder_destroy(hmm_encoder*) {}
extern "C" int hmm_encoder_load_param(hmm_encoder*, const char*, const float*, int64_t, hmm_stream_t) { return HMM_E_STATE; }
extern "C" int hmm_encoder_missing_params(hmm_encoder*) { return -1; }
extern "C" size_t hmm_encoder_workspace_bytes(const hmm_encoder*, int) { return 0; }
extern "C" int hmm_encoder_forward(hmm_encoder*, const float*, int, float*, void*, size_t, hmm_stream_t) { return HMM_E_STATE; }
extern "C" double hmm_encoder_flops(const hmm_encoder*, int) { return 0; }
extern "C" int hmm_op_gemm_bf16(const uint16_t*, const uint16_t*, const float*, void*, int, int, int, int, hmm_stream_t) { return HMM_E_STATE; }
extern "C" int hmm_op_layernorm_bf16(const float*, const float*, const float*, uint16_t*, int, int, float, hmm_stream_t) { return HMM_E_STATE; }
extern "C" int hmm_op_attention_bf16(const uint16_t*, uint16_t*, int, int, int, int, const float*, const float*, hmm_stream_t) { return HMM_E_STATE; }
