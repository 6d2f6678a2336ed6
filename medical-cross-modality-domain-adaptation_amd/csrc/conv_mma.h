// conv_mma.h — the register-pipelined fragment set and the slice schedule macros of the fp32 MFMA main loops
// (conv_igemm.hip: convolutions; conv_wino.hip: the batched GEMM of the Winograd path).  Moved here unchanged from conv_igemm.hip.
#pragma once
#include "conv_common.h"

namespace pnpconv {

// ---- register-pipelined fragments: one 8-k slice (kq) of a stage ---------------------------------
// The main loops keep two Frag sets: while the 4*TM*TN MFMAs of slice kq run, the ds_reads of slice kq+1 are in
// flight, the global loads of the next stage are issued (slice 0) and stored to the other LDS buffer (slice 3).
// The only LDS latency a wave exposes per stage is the first slice's reads right after the barrier.
template <int TM, int TN, bool A_MMAJOR, int LDA, int LDB>
struct Frag {
    f32x4 a[TM];
    float b[4][TN];
    __device__ __forceinline__ void load(const float* __restrict__ As, const float* __restrict__ Bs, int kq, int wm0, int wn0,
                                         int lane) {
        const int l31 = lane & 31;
        const int kb = kq * 8 + 4 * (lane >> 5);
        if constexpr (A_MMAJOR) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(As + (wm0 + tm * 32 + l31) * LDA + kb);
        } else {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int j = 0; j < 4; ++j) a[tm][j] = As[(kb + j) * LDA + wm0 + tm * 32 + l31];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[j][tn] = Bs[(kb + j) * LDB + wn0 + tn * 32 + l31];
    }
    __device__ __forceinline__ void mma(Acc<TM, TN>& acc) const {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][j], b[j][tn], acc.v[tm][tn], 0, 0, 0);
    }
};
}  // namespace pnpconv

#define PNP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// Schedule of slices 1..3 of a stage (two Frag sets: the fragment reads of the NEXT slice and the MFMAs of THIS slice are independent).
// PNP_CONV_ILV selects where the next stage's LDS stores go (compile-time; measured A/B/A/B on the 512-channel layers at B=16):
//   0  after the last slice's MFMAs (round 1)
//   1  as 0, with the fragment reads of slices 2 / 3 interleaved behind single MFMAs instead of in front of them: no change
//   2  behind the second half of the last slice's MFMAs: 512->512 forward +2.4 %, g10 forward +3.2 % / wgrad +4.6 %, segmenter step
//      422 -> 434 slices/s, joint GAN step 147.4 -> 151.1 — the stores (and the vmcnt wait in front of them) left the exposed chain
//      [last MFMA -> stores -> barrier -> first fragment reads -> first MFMA] that the co-resident workgroup has to cover
//   3  behind the first half of the last slice;  4  behind the second half of slice 2 (loads are issued under slice 0)
#ifndef PNP_CONV_ILV
#define PNP_CONV_ILV 2
#endif
#define PNP_STORE_BEHIND(MMA, STORE, NMFMA, LEAD)                           \
    MMA;                                                                    \
    STORE;                                                                  \
    if ((LEAD) > 0) __builtin_amdgcn_sched_group_barrier(0x008, (LEAD), 0); \
    _Pragma("unroll") for (int i_ = 0; i_ < (NMFMA) / 2; ++i_) {            \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                  \
    }                                                                       \
    PNP_SCHED_FENCE();
// last slice (+ the next stage's LDS stores unless they already went behind slice 2)
#define PNP_LAST_SLICE(MMA, STORE, NMFMA)                                   \
    if constexpr (PNP_CONV_ILV == 2) {                                      \
        PNP_STORE_BEHIND(MMA, STORE, NMFMA, (NMFMA) / 2)                    \
    } else if constexpr (PNP_CONV_ILV == 3) {                               \
        PNP_STORE_BEHIND(MMA, STORE, NMFMA, 0)                              \
    } else if constexpr (PNP_CONV_ILV == 4) {                               \
        MMA;                                                                \
        PNP_SCHED_FENCE();                                                  \
    } else {                                                                \
        MMA;                                                                \
        PNP_SCHED_FENCE();                                                  \
        STORE;                                                              \
        PNP_SCHED_FENCE();                                                  \
    }
// slice 2 (variant 4 carries the stores here)
#define PNP_SLICE2(LOAD, MMA, STORE, NMFMA, NDS)                            \
    if constexpr (PNP_CONV_ILV == 4) {                                      \
        LOAD;                                                               \
        PNP_SCHED_FENCE();                                                  \
        PNP_STORE_BEHIND(MMA, STORE, NMFMA, (NMFMA) / 2)                    \
    } else {                                                                \
        PNP_SLICE(LOAD, MMA, NMFMA, NDS)                                    \
    }
#define PNP_SLICE(LOAD, MMA, NMFMA, NDS)                                   \
    if constexpr (PNP_CONV_ILV == 1) {                                      \
        LOAD;                                                               \
        MMA;                                                                \
        _Pragma("unroll") for (int i_ = 0; i_ < (NMFMA); ++i_) {            \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);              \
            __builtin_amdgcn_sched_group_barrier(0x100, (NDS), 0);          \
        }                                                                   \
        PNP_SCHED_FENCE();                                                  \
    } else {                                                                \
        LOAD;                                                               \
        PNP_SCHED_FENCE();                                                  \
        MMA;                                                                \
        PNP_SCHED_FENCE();                                                  \
    }
