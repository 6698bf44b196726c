// Attention compute on K / V images resident in LDS, shared by the stand-alone attention kernel (attention.hip: K and V
// staged from the packed qkv matrix in HBM) and the fused QKV-projection + attention kernel (qkv_attention.hip: Q, K
// and V written into LDS straight from the projection's accumulators).  Same code, same bits.
//
// One workgroup (8 waves, two per SIMD) per (sample, head).  K and V are ROW-major in LDS: K rows padded to an odd
// multiple of 16 B (conflict-free ds_read_b128 of the A-operand), V rows at a 192-B stride (the four rows a
// ds_read_b64_tr_b16 block touches fall in disjoint bank windows).  Each wave owns 32-query tiles.
//
// The score tile is computed TRANSPOSED, S^T = K Q^T (v_mfma_f32_32x32x16_bf16, K as the A-operand),
// so a lane holds one query column with its keys in registers: max / sum are register-local plus
// one exchange with lane^32, and the exponentiated accumulator, converted to bf16 in place, IS the
// B-operand of the second product O^T = V^T P^T (no LDS round trip for P).  V^T fragments come
// straight from the row-major V image through the transposing LDS read.  Keys are processed in
// chunks of 3 tiles (96 keys) with an online softmax (running max / sum, O rescaled per chunk),
// which keeps a wave under 256 VGPRs so that two waves share each SIMD: while one is in its
// exp / max / sum VALU section, its partner issues MFMAs.
// fp32 scores, statistics and output accumulation; P and the output are rounded to bf16 once.
#pragma once
#include "hmm_common.h"

namespace hmm {

typedef __bf16 __attribute__((address_space(3))) * lds_bf16_ptr;

template <int DH, int NKT>
struct AttnCfg {
    static constexpr int KS = DH / 16;              // k-steps of QK^T
    static constexpr int DT = (DH + 31) / 32;       // 32-row d tiles of O^T
    static constexpr int NKEY = NKT * 32;
    static constexpr int KROW = DH * 2 + 16;        // K row stride, bytes
    static constexpr int VROW = 192;                // V row stride, bytes (>= 2*32*DT)
    static constexpr int K_BYTES = NKEY * KROW;
    static constexpr int V_BYTES = NKEY * VROW;
    static constexpr int X_FLOATS = DH + 2;         // cooperative extra-query partial: O[DH], m, l
    static constexpr int X_BYTES = (NKT * X_FLOATS * 4 + 15) / 16 * 16;
    static constexpr int LDS = K_BYTES + V_BYTES + X_BYTES;
    static constexpr int CH = 3;                    // key tiles per online-softmax chunk
    static constexpr int NCH = (NKT + CH - 1) / CH;
    static_assert(DT * 64 <= VROW, "V row must cover every d tile a tr-read touches");
};

constexpr int kAttnWaves = 8;

// probe build: in-kernel stamps of the attention phase (tools/fused_stamp_probe.py); nothing in the product library
#ifdef HMM_PROBE
#define HMM_ATTN_PROBE_PARAM , unsigned long long* attn_stamps = nullptr
#define HMM_ATTN_STAMP(slot) \
    if (attn_stamps && threadIdx.x == 0) attn_stamps[(slot)] = __builtin_amdgcn_s_memrealtime();
#else
#define HMM_ATTN_PROBE_PARAM
#define HMM_ATTN_STAMP(slot)
#endif

// k_lds / v_lds: the images (rows >= Lk zero-filled); part: C::X_BYTES of LDS scratch; load_q(row, ks) -> the bf16x8
// fragment d = 16 ks + 8 hh .. + 7 of query `row` (row < T) for this lane's hh = lane >> 5; qf: the fragments of query
// tile `first_qt`, preloaded by the caller; out_head = out + sample * T * D + head * DH.  Ends with the results stored.
// Query split (few samples: one workgroup per (sample, head) leaves most CUs idle): this workgroup is part q_part of q_parts
// and takes the 32-query tiles q_part + q_parts * (wave + 8 i); the single extra query (T = 8 * 32 + 1) goes with part 0.
// A query tile is computed by one wave on its own whichever workgroup holds it: the same bits for every q_parts.
template <int DH, int NKT, class QLoad>
__device__ __forceinline__ void attention_core(const char* k_lds, const char* v_lds, float* part, QLoad load_q,
                                               bf16x8 (&qf)[AttnCfg<DH, NKT>::KS], bf16_t* __restrict__ out_head,
                                               int T, int Lk, int D, float scale_log2e, bool causal, int q_part = 0,
                                               int q_parts = 1 HMM_ATTN_PROBE_PARAM) {
    using C = AttnCfg<DH, NKT>;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    const int hh4 = 4 * hh;
    const float neg_inf = -INFINITY;
    const int nqt = (T + 31) >> 5;
    // per-lane LDS bases
    const char* k_base = k_lds + r * C::KROW + hh * 16;
    // tr-read: 16-lane group g = lane>>4 reads a 4-key x 16-d block; lane (4q+p) of the group supplies
    // row q, columns 4p..4p+3, and receives column (lane&15) of the 4 rows.
    const int g = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
    const char* v_base = v_lds + (4 * (g >> 1) + q4) * C::VROW + (16 * (g & 1) + 4 * p4) * 2;

    // 257 = 8*32 + 1: a ninth query tile with ONE valid row would make wave 0 run two tiles back to
    // back (the critical path of the workgroup).  When the tile past the 8 main ones holds exactly one
    // query, that query is instead processed cooperatively below: one key tile per wave.
    const bool coop = (nqt == kAttnWaves + 1) && (T - kAttnWaves * 32 == 1) && !causal;
    const int nqt_main = coop ? kAttnWaves : nqt;
    const int first_qt = q_part + q_parts * wave;
    for (int qt = first_qt; qt < nqt_main; qt += kAttnWaves * q_parts) {
        const int qrow = qt * 32 + r;
        if (qt != first_qt) {                               // later tiles reload their queries
            const int qr = qrow < T ? qrow : T - 1;
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) qf[ks] = load_q(qr, ks);
        }
        f32x16 o[C::DT];
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[dt] = f32x16{};
        float m_run = -INFINITY, l_run = 0.f;

#pragma unroll
        for (int ch = 0; ch < C::NCH; ++ch) {
            constexpr int CHT = C::CH;
            f32x16 s[CHT];
            // S^T tiles of this chunk
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
                    f32x16 acc = {};
                    bf16x8 kf[C::KS];
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
                        kf[ks] = *reinterpret_cast<const bf16x8*>(k_base + kt * 32 * C::KROW + ks * 32);
#pragma unroll
                    for (int ks = 0; ks < C::KS; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], acc, 0, 0, 0);
                    s[i] = acc;
                    __builtin_amdgcn_sched_barrier(0);     // one tile's K fragments live at a time
                }
            }
            // mask keys >= Lk.  Only the tile(s) straddling Lk need it (wave-uniform branch); the compare
            // and select are opaque asm so that the compiler neither hoists 144 lane masks into SGPRs
            // nor if-converts the branch.
            float mc = -INFINITY;
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
                    if ((kt + 1) * 32 > Lk || causal) {
                        const int lim = causal ? (qrow + 1 < Lk ? qrow + 1 : Lk) : Lk;   // first invisible key
                        const int rel = lim - kt * 32 - hh4;         // key masked iff (reg&3)+8*(reg>>2) >= rel
#define HMM_MASK1(REG, KC)                                                                         \
    {                                                                                              \
        float v = s[i][REG];                                                                       \
        asm volatile("v_cmp_ge_i32 vcc, " #KC ", %1\n\tv_cndmask_b32 %0, %0, %2, vcc"              \
                     : "+v"(v) : "v"(rel), "v"(neg_inf) : "vcc");                                  \
        s[i][REG] = v;                                                                             \
    }
                        HMM_MASK1(0, 0) HMM_MASK1(1, 1) HMM_MASK1(2, 2) HMM_MASK1(3, 3)
                        HMM_MASK1(4, 8) HMM_MASK1(5, 9) HMM_MASK1(6, 10) HMM_MASK1(7, 11)
                        HMM_MASK1(8, 16) HMM_MASK1(9, 17) HMM_MASK1(10, 18) HMM_MASK1(11, 19)
                        HMM_MASK1(12, 24) HMM_MASK1(13, 25) HMM_MASK1(14, 26) HMM_MASK1(15, 27)
#undef HMM_MASK1
                    }
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) mc = fmaxf(mc, s[i][reg]);
                }
            }
            mc = fmaxf(mc, __shfl_xor(mc, 32, 64));
            const float m_new = fmaxf(m_run, mc);
            // a chunk whose keys are all masked (mc = -inf while m_run = -inf) cannot occur: chunk 0
            // always holds key 0.  Later all-masked chunks keep m_new = m_run (finite).
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * scale_log2e);
            const float neg_m = -m_new * scale_log2e;
            m_run = m_new;
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) o[dt][reg] *= alpha;
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
#pragma unroll
                    for (int reg = 0; reg < 16; ++reg) {
                        const float p = __builtin_amdgcn_exp2f(fmaf(s[i][reg], scale_log2e, neg_m));
                        s[i][reg] = p;
                        l_run += p;
                    }
                }
            }
            // O^T += V^T P^T for the chunk: accumulator registers 8*st..8*st+7 of tile kt are the B
            // fragment of k-step st (slot j of lane half hh = key 16*st + 8*(j>>2) + 4*hh + (j&3)).
#pragma unroll
            for (int i = 0; i < CHT; ++i) {
                const int kt = ch * CHT + i;
                if (kt < NKT) {
#pragma unroll
                    for (int st = 0; st < 2; ++st) {
                        bf16x8 pf;
#pragma unroll
                        for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)s[i][8 * st + j];
#pragma unroll
                        for (int dt = 0; dt < C::DT; ++dt) {
                            const char* vp = v_base + (kt * 32 + 16 * st) * C::VROW + dt * 64;
                            const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
                            const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8 * C::VROW));
                            const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                            o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }

        // A lane holds 4 consecutive d (8 B) per accumulator group and its partner lane^32 the next 4;
        // v_permlane32_swap pairs two groups so that every lane stores 16 B (half the store instructions, 32-B pieces).
        const float l = l_run + __shfl_xor(l_run, 32, 64);
        const float inv_l = 1.0f / l;
        bf16_t* op = out_head + (size_t)(qrow < T ? qrow : T - 1) * D + 8 * hh;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (dt * 32 + 16 * m < DH) {                          // compile-time: 5 (dh 80) / 4 (dh 64) stores
                    const int ra = 8 * m, rb = 8 * m + 4;             // accumulator groups gq = 2m and 2m+1
                    bf16x4 a4 = {(bf16_t)(o[dt][ra + 0] * inv_l), (bf16_t)(o[dt][ra + 1] * inv_l),
                                 (bf16_t)(o[dt][ra + 2] * inv_l), (bf16_t)(o[dt][ra + 3] * inv_l)};
                    bf16x4 b4 = {(bf16_t)(o[dt][rb + 0] * inv_l), (bf16_t)(o[dt][rb + 1] * inv_l),
                                 (bf16_t)(o[dt][rb + 2] * inv_l), (bf16_t)(o[dt][rb + 3] * inv_l)};
                    const uint2 pa = __builtin_bit_cast(uint2, a4), pb = __builtin_bit_cast(uint2, b4);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pa.x, pb.x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pa.y, pb.y, false, false);
                    if (qrow < T) *reinterpret_cast<uint4*>(op + dt * 32 + 16 * m) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
                }
            }
    }

    HMM_ATTN_STAMP(3)
    if (coop && q_part == 0) {                              // workgroup-uniform
        const int xq = T - 1;                                  // the extra query row: every column of the B operand = this query
#pragma unroll
        for (int ks = 0; ks < C::KS; ++ks) qf[ks] = load_q(xq, ks);
        for (int kt = wave; kt < NKT; kt += kAttnWaves) {      // wave 0 also takes the last key tile
            f32x16 sx = {};
#pragma unroll
            for (int ks = 0; ks < C::KS; ++ks) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(k_base + kt * 32 * C::KROW + ks * 32);
                sx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sx, 0, 0, 0);
            }
            float mw = -INFINITY;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int key = kt * 32 + (reg & 3) + 8 * (reg >> 2) + hh4;
                const float v = key < Lk ? sx[reg] : -INFINITY;
                sx[reg] = v;
                mw = fmaxf(mw, v);
            }
            mw = fmaxf(mw, __shfl_xor(mw, 32, 64));            // finite: every key tile holds a key < Lk
            const float neg_m = -mw * scale_log2e;
            float lw = 0.f;
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(sx[reg], scale_log2e, neg_m));
                sx[reg] = pv;
                lw += pv;
            }
            lw += __shfl_xor(lw, 32, 64);
            f32x16 ox[C::DT];
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) ox[dt] = f32x16{};
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                bf16x8 pf;
#pragma unroll
                for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)sx[8 * st + j];
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt) {
                    const char* vp = v_base + (kt * 32 + 16 * st) * C::VROW + dt * 64;
                    const bf16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp));
                    const bf16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(vp + 8 * C::VROW));
                    const bf16x8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    ox[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, ox[dt], 0, 0, 0);
                }
            }
            if (r == 0) {                                       // column 0: lanes 0 (hh=0) and 32 (hh=1)
                float* dst = part + kt * C::X_FLOATS;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int d0 = dt * 32 + 8 * gq + hh4;
                        if (d0 < DH) {
                            dst[d0 + 0] = ox[dt][4 * gq + 0]; dst[d0 + 1] = ox[dt][4 * gq + 1];
                            dst[d0 + 2] = ox[dt][4 * gq + 2]; dst[d0 + 3] = ox[dt][4 * gq + 3];
                        }
                    }
                if (hh == 0) { dst[DH] = mw; dst[DH + 1] = lw; }
            }
        }
        __syncthreads();
        if (tid < DH) {                                         // combine the NKT partials for output column tid
            float m = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) m = fmaxf(m, part[kt * C::X_FLOATS + DH]);
            float acc = 0.f, l = 0.f;
#pragma unroll
            for (int kt = 0; kt < NKT; ++kt) {
                const float w = __builtin_amdgcn_exp2f((part[kt * C::X_FLOATS + DH] - m) * scale_log2e);
                acc = fmaf(part[kt * C::X_FLOATS + tid], w, acc);
                l = fmaf(part[kt * C::X_FLOATS + DH + 1], w, l);
            }
            out_head[(size_t)xq * D + tid] = (bf16_t)(acc / l);
        }
    }
}

}  // namespace hmm
