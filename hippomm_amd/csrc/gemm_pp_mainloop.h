// Main loop of the 256x256x64 ping-pong bf16 GEMM tile (gemm_bf16.hip) as a device function, so that the fused
// QKV-projection + attention kernel (qkv_attention.hip) runs exactly the same K loop -- same staging, same MFMA order, same
// bits -- as the stand-alone GEMM.
//
// 8 waves as 2 (M) x 4 (N), wave tile 128x64 = 4 quadrants of 64x32 (16 MFMAs each for the full K = 64).
// A K-tile lives in LDS as four 16-KiB half-tiles chosen so that each one dies early:
//   A_lo = the first 64 rows of every wave-row, A_hi = the second 64; B_lo / B_hi likewise in N.
//   phase A (quadrants lo,lo and lo,hi) reads A_lo, B_lo, B_hi into registers -> all three are dead after its reads
//   phase B (quadrants hi,hi and hi,lo) reads A_hi, reuses the B registers     -> A_hi is dead after its reads
// so phase B re-stages (t+2).A_lo, B_lo, B_hi and the next phase A re-stages (t+2).A_hi (2 x global_load_lds_dwordx4 per
// wave and half-tile): the LDS-DMA stream runs 1-2 K-tiles ahead of the MFMAs and is never drained in the loop; the only
// wait is a counted vmcnt(8) at the end of each read section, which retires the half-tiles the NEXT phase reads (staged two
// phases earlier) while the 8 youngest pieces stay in flight.  Waves 4-7 run one barrier behind waves 0-3, so on every
// SIMD one wave issues its 32-MFMA burst while its partner does its LDS reads and DMA issue.
// Hazards: a half-tile is re-staged only in a phase after the one whose reads were retired by
// lgkmcnt(0) *before* that phase's first barrier (WAR); staged data is read only in a phase after
// the barrier that follows every wave's counted vmcnt (RAW).
// The LDS image of a half-tile is lane-linear for the DMA (1 KiB = 8 rows x 128 B per wave-instruction); the bank
// swizzle chunk' = chunk ^ ((row >> 1) & 7) is applied to the per-lane SOURCE address by the caller (PPSources) and on
// the ds_read_b128 side here.
// Ablations, tile-walk and pipeline variants that were measured and rejected are recorded in profiles/LABNOTES.md 4.3 / 4.5.
#pragma once
#include "hmm_common.h"

namespace hmm {

#define HMM_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define HMM_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Per-lane staging sources: 32-bit ELEMENT offsets from A / W of the two 1-KiB pieces (j = 0, 1) a wave stages per
// half-tile, for K-tile 0 (the loop adds kt * 64).  Piece j of wave w holds half-tile rows lr = (w + 8 j) * 8 + (lane >> 3),
// and lane (lane & 7) must point at 16-B chunk (lane & 7) ^ ((lr >> 1) & 7) of that row.
//   A_lo / A_hi: half-tile row lr <-> tile row (lr & 63) + (lr >> 6) * 128 (+ 64 for A_hi)
//   B_lo / B_hi: half-tile row lr <-> tile column (lr >> 5) * 64 + (lr & 31) (+ 32 for B_hi)
struct PPSources { int alo[2], ahi[2], blo[2], bhi[2]; };

constexpr int kPPHalf = 16384, kPPTile = 4 * kPPHalf;     // LDS: 2 K-tiles x 4 half-tiles = 128 KiB

// acc[mi][ni]: rows wm*128 + 16 mi + (lane & 15), columns wn*64 + 16 ni + 4 (lane >> 4) .. + 3 of the tile.
// KT = K / 64 must be even and >= 2.  On return every wave is past its last LDS read and every DMA has landed.
// The quadrants are issued in pairs, 32 MFMAs per burst: a K-tile has 4 barrier intervals, and the partner group's whole
// read section (16 / 8 fragment reads + its DMA issue) fits under one 512-cycle burst.  (Round 1 ran one quadrant per
// phase, 8 intervals per K-tile: same bits, 1-3 % slower in the tower.)
// TAIL (fused in_proj + attention kernels): 1 = tile columns 240..255 do not exist -- the waves of the last wave-column
// (wn == 3) skip the MFMAs of their last 16-column block (acc[..][3] stays zero) instead of multiplying padding;
// 2 = tile columns 192..255 do not exist (heads of 64: q | k | v fill three wave-columns) -- the waves of the last
// wave-column skip their fragment reads and MFMAs altogether and only keep staging and the barriers.
template <int TAIL = 0>
__device__ __forceinline__ void pp_mainloop(const bf16_t* __restrict__ A, const bf16_t* __restrict__ W,
                                            const PPSources& src, int KT, char* smem, int lane, int wave,
                                            f32x4 (&acc)[8][4]) {
    constexpr int HALF = kPPHalf, TILE = kPPTile;
    constexpr int H_ALO = 0, H_AHI = HALF, H_BLO = 2 * HALF, H_BHI = 3 * HALF;
    const int wm = wave >> 2, wn = wave & 3;
    const bool skip_tail = TAIL != 0 && wn == 3;                  // wave-uniform
    const bool idle = TAIL == 2 && skip_tail;                     // this wave computes nothing
#define HMM_STAGE(base, s, kt, buf, half)                                                                   \
    do {                                                                                                    \
        __builtin_amdgcn_global_load_lds(HMM_GLB_PTR((base) + (ptrdiff_t)(s[0] + (kt) * 64)),               \
                                         HMM_LDS_PTR(smem + (buf) * TILE + (half) + wave * 1024), 16, 0, 0); \
        __builtin_amdgcn_global_load_lds(HMM_GLB_PTR((base) + (ptrdiff_t)(s[1] + (kt) * 64)),               \
                                         HMM_LDS_PTR(smem + (buf) * TILE + (half) + (wave + 8) * 1024), 16, 0, 0); \
    } while (0)
    const int fsw = (lane & 15) >> 1;
    const int ck0 = ((lane >> 4) ^ fsw) * 16, ck1 = ((4 + (lane >> 4)) ^ fsw) * 16;
    const char* a_rd = smem + (wm * 64 + (lane & 15)) * 128;
    const char* b_rd = smem + (wn * 32 + (lane & 15)) * 128;
    bf16x8 af[4][2], blo[2][2], bhi[2][2];
#define HMM_READ_A(buf, half)                                                                         \
    if (!idle)                                                                                        \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
        af[mi][kh] = *reinterpret_cast<const bf16x8*>(a_rd + (buf) * TILE + (half) + mi * 2048 + (kh ? ck1 : ck0));
#define HMM_READ_B(dst, buf, half)                                                                    \
    if (!idle)                                                                                        \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                  \
        dst[ni][kh] = *reinterpret_cast<const bf16x8*>(b_rd + (buf) * TILE + (half) + ni * 2048 + (kh ? ck1 : ck0));
#define HMM_MFMA_QUAD(mo, no, bsrc)                                                                   \
    if (!idle)                                                                                        \
    _Pragma("unroll") for (int kh = 0; kh < 2; ++kh)                                                  \
    _Pragma("unroll") for (int mi = 0; mi < 4; ++mi)                                                  \
    _Pragma("unroll") for (int ni = 0; ni < 2; ++ni)                                                  \
        if (!(TAIL == 1 && (no) + ni == 3 && skip_tail))                                              \
            acc[(mo) + mi][(no) + ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bsrc[ni][kh], af[mi][kh], \
                                                                                acc[(mo) + mi][(no) + ni], 0, 0, 0);
#define HMM_BAR()                              \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);
#define HMM_LGKM0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define HMM_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")

    // prologue: K-tile 0 complete, K-tile 1 complete; tile 0 must have landed
    HMM_STAGE(A, src.alo, 0, 0, H_ALO); HMM_STAGE(W, src.blo, 0, 0, H_BLO);
    HMM_STAGE(W, src.bhi, 0, 0, H_BHI); HMM_STAGE(A, src.ahi, 0, 0, H_AHI);
    HMM_STAGE(A, src.alo, 1, 1, H_ALO); HMM_STAGE(W, src.blo, 1, 1, H_BLO);
    HMM_STAGE(W, src.bhi, 1, 1, H_BHI); HMM_STAGE(A, src.ahi, 1, 1, H_AHI);
    HMM_VM8();
    HMM_BAR();
    if (wm == 1) { HMM_BAR(); }                               // waves 4-7 run one barrier behind

#define HMM_KTILE2(t, buf)                                                                \
    {                                                                                     \
        /* phase A: (lo,lo) + (lo,hi).  (t).A_hi of the OTHER buffer's previous tile died in phase B(t-1): re-stage it */ \
        HMM_READ_A(buf, H_ALO) HMM_READ_B(blo, buf, H_BLO) HMM_READ_B(bhi, buf, H_BHI)    \
        if ((t) >= 1 && (t) + 1 < KT) HMM_STAGE(A, src.ahi, (t) + 1, (buf) ^ 1, H_AHI);   \
        if ((t) + 1 < KT) { HMM_VM8(); } else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } \
        HMM_LGKM0(); HMM_BAR()                                                            \
        __builtin_amdgcn_s_setprio(1);                                                    \
        HMM_MFMA_QUAD(0, 0, blo) HMM_MFMA_QUAD(0, 2, bhi)                                 \
        __builtin_amdgcn_s_setprio(0);                                                    \
        HMM_BAR()                                                                         \
        /* phase B: (hi,hi) + (hi,lo); A_lo, B_lo, B_hi of this buffer are dead: stage tile t+2 into them */ \
        HMM_READ_A(buf, H_AHI)                                                            \
        if ((t) + 2 < KT) {                                                               \
            HMM_STAGE(A, src.alo, (t) + 2, buf, H_ALO); HMM_STAGE(W, src.blo, (t) + 2, buf, H_BLO); \
            HMM_STAGE(W, src.bhi, (t) + 2, buf, H_BHI);                                   \
            HMM_VM8();                                                                    \
        } else {                                                                          \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                              \
        }                                                                                 \
        HMM_LGKM0(); HMM_BAR()                                                            \
        __builtin_amdgcn_s_setprio(1);                                                    \
        HMM_MFMA_QUAD(4, 2, bhi) HMM_MFMA_QUAD(4, 0, blo)                                 \
        __builtin_amdgcn_s_setprio(0);                                                    \
        HMM_BAR()                                                                         \
    }

    for (int t = 0; t < KT; t += 2) {
        HMM_KTILE2(t, 0)
        HMM_KTILE2(t + 1, 1)
    }
    if (wm == 0) { HMM_BAR(); }                               // re-align the two wave groups
#undef HMM_KTILE2
#undef HMM_STAGE
#undef HMM_READ_A
#undef HMM_READ_B
#undef HMM_MFMA_QUAD
#undef HMM_BAR
#undef HMM_LGKM0
#undef HMM_VM8
}

}  // namespace hmm
