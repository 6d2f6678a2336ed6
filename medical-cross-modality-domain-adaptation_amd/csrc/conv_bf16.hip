// conv_bf16.hip — bf16-operand MFMA convolutions (BASELINE.json configs[4]: bf16 mixed precision).
//
// Same implicit GEMMs as conv_igemm.hip, with the MFMA operands rounded to bfloat16 (round-to-nearest-even, v_cvt_pk_bf16_f32) while a
// tile is staged into LDS and contracted by v_mfma_f32_32x32x16_bf16 (fp32 accumulation, 16x the fp32-MFMA rate).  Tensors in HBM stay
// fp32 — activations, filters (= the fp32 master weights), gradients — so every other kernel of the step is untouched and the
// arithmetic is exactly "both operands of every convolution rounded to bf16, products accumulated in fp32": what
// oracle/tf_ops.py (round_bf16) restates and tests/test_bf16_budget.py budgets.
//
//   conv_taps_bf16_kernel   forward (any stride) / stride-1 data gradient / stride-phase sub-filters: zero padding, C % 32 == 0, taps
//                           unrolled (the layers conv_taps_kernel serves)
//   conv_wgrad_bf16_kernel  filter gradient, stride 1, zero padding, C % 4 == 0, OW >= 32 (the layers conv_wgrad_kernel<.., 3, ..> serves)
// Everything else (C in {3,5,16,40}, K <= 16, strided filter gradients, in-kernel SYMMETRIC) stays on the fp32 kernels: those layers
// are HBM- or launch-bound, not MFMA-bound.  So do the filter gradients of 16 / 32-channel inputs with 32 / 64 filters on maps of >= 8192 pixels
// (conv_small.hip's 16x16x4 fp32 tiles: 0.064 / 0.113 / 0.053 ms against 0.109 / 0.146 / 0.096 ms on the 128-row bf16 tiles at B = 16).
// PNP_DTYPE_BF16 in a geometry PERMITS bf16 operands; a layer that stays fp32 is exact, never looser.
//
// LDS tiles hold bf16 with the REDUCTION index contiguous for BOTH operands (the 32x32x16 MFMA takes 8 consecutive k per lane from
// each): rows of 32 k (64 B) + 16 B pad = 80 B, so a ds_read_b128 lane group (16 rows) lands on 16 distinct 16-byte slots
// (5 r mod 16 is a bijection).  Operands whose reduction index is NOT contiguous in memory — filters [k][n] for fwd / dgrad, x and dy
// ([pixel][channel]) for the filter gradient — are transposed on the way in: a thread fetches a 4 x 4 block (four 16-byte loads from
// four consecutive reduction rows), transposes it in registers and writes four 8-byte rows.  Thread -> block map: reduction block =
// t & 7, row block = t >> 3, so that 16 consecutive lanes write 2 x 64 contiguous bytes 80*4 B apart (all 32 banks, conflict-free).
#include "conv_common.h"

using namespace pnpconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int LDH = BK + 8;     // bf16 elements per LDS row

__device__ __forceinline__ bf16x4 cvt4(float a, float b, float c, float d) {
    bf16x4 h;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    return h;
}

template <int TM, int TN>
struct FragH {
    bf16x8 a[TM][2], b[TN][2];
    __device__ __forceinline__ void load(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int wm0, int wn0, int lane) {
        const int l31 = lane & 31, kh = (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[tm][ks] = *reinterpret_cast<const bf16x8*>(As + (wm0 + tm * 32 + l31) * LDH + ks * 16 + kh);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[tn][ks] = *reinterpret_cast<const bf16x8*>(Bs + (wn0 + tn * 32 + l31) * LDH + ks * 16 + kh);
        }
    }
    __device__ __forceinline__ void mma(Acc<TM, TN>& acc) const {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm][ks], b[tn][ks], acc.v[tm][tn], 0, 0, 0);
    }
};

// a [32 k][ROWS n] block of a row-major fp32 matrix -> LDS rows [n][k]: 4 x 4 register transposes (see the file header)
template <int ROWS>
struct TransposedTile {
    static constexpr int NB = ROWS / 4;                  // 4-row blocks of the tile
    static constexpr int PASSES = (NB + 31) / 32;        // 32 row blocks per pass of the 256 threads
    f32x4 reg[PASSES][4];
    int kb, nb;
    __device__ __forceinline__ void init(int t) { kb = t & 7; nb = t >> 3; }
    __device__ __forceinline__ bool active(int p) const { return nb + 32 * p < NB; }
    __device__ __forceinline__ void store(__bf16* lds) const {
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
            if (!active(p)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<bf16x4*>(lds + (4 * (nb + 32 * p) + e) * LDH + 4 * kb) = cvt4(reg[p][0][e], reg[p][1][e], reg[p][2][e], reg[p][3][e]);
        }
    }
};

// filter-gradient tile -> dW rows (or a split partial)
// ======================= forward / stride-1 data gradient / stride-phase sub-filters (taps unrolled) ============================
constexpr int DEPTH = 3;      // global-load stages in flight per workgroup (register ring)

template <int BM, int BN, int WM, int WN, int KIND, int R, int S>
__global__ void __launch_bounds__(NTHREADS, 2) conv_taps_bf16_kernel(ConvArgs a) {
    constexpr int NTAP = R * S;
    static_assert(NTAP <= 32, "one validity bit per tap");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ASZ = BM * LDH, BSZ = BN * LDH;
    constexpr int NR = BM / 32;
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (ASZ + BSZ)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = a.nblk_m * a.nblk_n;
    const int z = blockIdx.x / nblk;
    int bid = blockIdx.x - z * nblk;
    if (a.xcd_swizzle) bid = xcd_remap(bid, nblk);
    int mt, nt;
    tile_coords(bid, a.nblk_m, a.nblk_n, a.gn, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    // A rows (output pixels): byte offset of the pixel shifted by -pad, one validity bit per tap
    const int kg = t & 7, mrow = t >> 3;
    int abase[NR];
    unsigned amask[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int m = m0 + mrow + 32 * i;
        const bool ok = m < a.M;
        if (!ok) m = 0;
        const int n = m / a.OHW;
        const int rem = m - n * a.OHW;
        const int oh = rem / a.OW;
        const int ow = rem - oh * a.OW;
        const int vh0 = oh * a.stride - a.pad_t, vw0 = ow * a.stride - a.pad_l;
        abase[i] = (((n * a.H + vh0) * a.W + vw0) * a.C + 4 * kg) * 4;
        unsigned mk = 0;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int ih = vh0 + (tap / S) * a.dil, iw = vw0 + (tap % S) * a.dil;
            const bool v = ok & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            mk |= (v ? 1u : 0u) << tap;
        }
        amask[i] = mk;
    }
    // B: filter rows [k][n] -> LDS [n][k]
    TransposedTile<BN> tb;
    tb.init(t);
    unsigned boff[TransposedTile<BN>::PASSES][4];
#pragma unroll
    for (int p = 0; p < TransposedTile<BN>::PASSES; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * (tb.nb + 32 * p);
            boff[p][j] = (tb.active(p) && n < a.K) ? (unsigned)(((4 * tb.kb + j) * a.K + n) * 4) : OOB2;
        }

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    // Register ring of DEPTH stages: a 32-k stage is only 8 MFMAs (256 matrix-pipe cycles) per wave, a tenth of the global-load
    // latency under load, so the loads of stage j + DEPTH are issued while stage j is contracted (first version: one stage ahead,
    // stage time == load latency: 512->512 ran at 342 TF/s).  The ring is indexed with compile-time constants only (loop unrolled
    // DEPTH times) so that it lives in VGPRs.  No stage is conditional inside the loop (a uniform "past the end?" select around a load
    // makes hipcc branch around every load and wait vmcnt(0) in between — seen in the first attempt): fetches past the last stage
    // simply re-read the last channel group (valid addresses, never stored), and the nst % DEPTH leftover stages run after the loop.
    struct Stage {
        f32x4 a[NR];
        f32x4 b[TransposedTile<BN>::PASSES][4];
    };
    Stage ring[DEPTH];
    const int ncc_total = a.C / BK;
    const int cc_begin = z * (a.chunks_per_split / NTAP);
    int cc_end = cc_begin + a.chunks_per_split / NTAP;
    if (cc_end > ncc_total) cc_end = ncc_total;
    const int nst = (cc_end - cc_begin) * NTAP;
    int l_cc = cc_begin, l_tap = 0;                        // (channel group, tap) of the next stage to fetch: uniform, scalar registers
    auto gload = [&](Stage& st) {
        const int tr = l_tap / S, ts = l_tap - tr * S;
        const int tshift = ((tr * a.dil * a.W + ts * a.dil) * a.C) * 4;
        const int sa = l_cc * (BK * 4);
        const int sb = ((l_tap * a.C + l_cc * BK) * a.K) * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned vo = ((amask[i] >> l_tap) & 1u) ? (unsigned)(abase[i] + tshift) : OOB2;
            st.a[i] = bload4s(rx, vo, sa);
        }
#pragma unroll
        for (int p = 0; p < TransposedTile<BN>::PASSES; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) st.b[p][j] = bload4s(rw, boff[p][j], sb);
        const int wrap = (l_tap + 1 == NTAP) ? 1 : 0;
        l_tap = wrap ? 0 : l_tap + 1;
        l_cc = min(l_cc + wrap, cc_end - 1);
    };
    auto lstore = [&](const Stage& st, __bf16* An, __bf16* Bn) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            *reinterpret_cast<bf16x4*>(An + (mrow + 32 * i) * LDH + 4 * kg) = cvt4(st.a[i][0], st.a[i][1], st.a[i][2], st.a[i][3]);
#pragma unroll
        for (int p = 0; p < TransposedTile<BN>::PASSES; ++p) {
            if (!tb.active(p)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<bf16x4*>(Bn + (4 * (tb.nb + 32 * p) + e) * LDH + 4 * tb.kb) = cvt4(st.b[p][0][e], st.b[p][1][e], st.b[p][2][e], st.b[p][3][e]);
        }
    };

    Acc<TM, TN> acc;
    acc.zero();
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) gload(ring[d]);
    lstore(ring[0], lds, lds + 2 * ASZ);
    __syncthreads();
    FragH<TM, TN> f;
    // stage j: LDS buffer j & 1; ring[j % DEPTH] is free (its stage was stored during stage j - 1) and takes stage j + DEPTH;
    // ring[(j + 1) % DEPTH] holds stage j + 1, fetched DEPTH - 1 stages ago
    auto stage = [&](int j, Stage& fetch_into, const Stage& store_from, bool fetch) {
        const int cur = j & 1;
        const __bf16* As = lds + cur * ASZ;
        const __bf16* Bs = lds + 2 * ASZ + cur * BSZ;
        f.load(As, Bs, wm0, wn0, lane);
        if (fetch) gload(fetch_into);
        f.mma(acc);
        lstore(store_from, lds + (cur ^ 1) * ASZ, lds + 2 * ASZ + (cur ^ 1) * BSZ);
        __syncthreads();
    };
    const int nmain = (nst / DEPTH) * DEPTH;
    for (int j0 = 0; j0 < nmain; j0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) stage(j0 + d, ring[d], ring[(d + 1) % DEPTH], true);
    }
    static_assert(DEPTH == 3, "tail below is written for a ring of three");
    if (nst - nmain >= 1) stage(nmain, ring[0], ring[1], false);
    if (nst - nmain >= 2) stage(nmain + 1, ring[1], ring[2], false);
    conv_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, m0, n0, wm0, wn0, lane, mt * WM + wave / WN, z == 0);
}

// ===================================== filter gradient ============================================
// dW[m' = (tap, c)][k] = sum_p x[p shifted by tap][c] * dy[p][k]; a.x = x, a.w = dy ([P][K]); reduction over output pixels p, split over
// workgroups (grid.x = tiles * nsplit, partials summed by conv_igemm.hip's splitk_reduce_kernel).  Both operands are pixel-major in
// memory, so both tiles are transposed on the way into LDS.
template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(NTHREADS, 2) conv_wgrad_bf16_kernel(ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ASZ = BM * LDH, BSZ = BN * LDH;
    static_assert(BM == 128, "one 4-row block of the A tile per thread");
    __shared__ __attribute__((aligned(16))) __bf16 lds[2 * (ASZ + BSZ)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = a.nblk_m * a.nblk_n;
    const int lid = a.xcd_swizzle ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int z = lid / nblk;
    const int bid = lid - z * nblk;
    int mt, nt;
    tile_coords(bid, a.nblk_m, a.nblk_n, a.gn, mt, nt);
    const int mm0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    const int P = a.M;
    const int nchunks_total = (P + BK - 1) / BK;
    const int c_begin = z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;
    const int nchunks = c_end - c_begin;

    // A: x rows (pixels) p = stage*32 + 4*kb + j, channels of this thread's 4 consecutive m' (one tap: C % 4 == 0)
    TransposedTile<BM> ta;
    TransposedTile<BN> tb;
    ta.init(t);
    tb.init(t);
    const int mm = mm0 + 4 * ta.nb;
    const bool mok = mm < a.Kred;
    const int m_ = mok ? mm : 0;
    const int rs_u = m_ / a.C, c_u = m_ - rs_u * a.C;
    const int r_u = rs_u / a.S, s_u = rs_u - r_u * a.S;
    const int l_dh = r_u * a.dil - a.pad_t, l_dw = s_u * a.dil - a.pad_l;
    int l_ow[4], l_oh[4], l_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int p = c_begin * BK + 4 * ta.kb + j;
        const int n = p / a.OHW;
        const int rem = p - n * a.OHW;
        l_oh[j] = rem / a.OW;
        l_ow[j] = rem - l_oh[j] * a.OW;
        l_off[j] = (((n * a.H + l_oh[j] + l_dh) * a.W + l_ow[j] + l_dw) * a.C + c_u) * 4;
    }
    unsigned boff[TransposedTile<BN>::PASSES][4];
#pragma unroll
    for (int p = 0; p < TransposedTile<BN>::PASSES; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + 4 * (tb.nb + 32 * p);
            boff[p][j] = (tb.active(p) && n < a.K) ? (unsigned)(((4 * tb.kb + j) * a.K + n) * 4) : OOB2;
        }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    // rows past the end of this split's pixel range: splits end on stage boundaries, and rows past the LAST pixel are past the end of
    // dy (hardware zero) — same argument as conv_wgrad_kernel's MODE 3.  Fetches past this split's last stage (the ring runs DEPTH ahead)
    // read the next split's rows, or zeros past the end of the tensors; those stages are never stored or contracted.
    struct Stage {
        f32x4 a[4];
        f32x4 b[TransposedTile<BN>::PASSES][4];
    };
    Stage ring[DEPTH];
    int l_chunk = c_begin;
    auto gload = [&](Stage& st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = mok & ((unsigned)(l_oh[j] + l_dh) < (unsigned)a.H) & ((unsigned)(l_ow[j] + l_dw) < (unsigned)a.W);
            st.a[j] = bload4(rx, ok ? (unsigned)l_off[j] : OOB);
            l_ow[j] += BK;
            l_off[j] += BK * a.C * 4;
            const bool ww = l_ow[j] >= a.OW;                    // at most one wrap per step because OW >= 32
            l_ow[j] -= ww ? a.OW : 0;
            l_oh[j] += ww ? 1 : 0;
            l_off[j] += ww ? (a.W - a.OW) * a.C * 4 : 0;
            const bool hw = l_oh[j] >= a.OH;
            l_oh[j] -= hw ? a.OH : 0;
            l_off[j] += hw ? (a.H - a.OH) * a.W * a.C * 4 : 0;
        }
        const int soff = l_chunk * BK * a.K * 4;
#pragma unroll
        for (int p = 0; p < TransposedTile<BN>::PASSES; ++p)
#pragma unroll
            for (int j = 0; j < 4; ++j) st.b[p][j] = bload4s(rw, boff[p][j], soff);
        l_chunk = min(l_chunk + 1, nchunks_total);          // rows past the end of dy read as zeros
    };
    auto lstore = [&](const Stage& st, __bf16* An, __bf16* Bn) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            *reinterpret_cast<bf16x4*>(An + (4 * ta.nb + e) * LDH + 4 * ta.kb) = cvt4(st.a[0][e], st.a[1][e], st.a[2][e], st.a[3][e]);
#pragma unroll
        for (int p = 0; p < TransposedTile<BN>::PASSES; ++p) {
            if (!tb.active(p)) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                *reinterpret_cast<bf16x4*>(Bn + (4 * (tb.nb + 32 * p) + e) * LDH + 4 * tb.kb) = cvt4(st.b[p][0][e], st.b[p][1][e], st.b[p][2][e], st.b[p][3][e]);
        }
    };

    Acc<TM, TN> acc;
    acc.zero();
    if (nchunks > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) gload(ring[d]);
        lstore(ring[0], lds, lds + 2 * ASZ);
        __syncthreads();
        FragH<TM, TN> f;
        auto stage = [&](int j, Stage& fetch_into, const Stage& store_from, bool fetch) {
            const int cur = j & 1;
            const __bf16* As = lds + cur * ASZ;
            const __bf16* Bs = lds + 2 * ASZ + cur * BSZ;
            f.load(As, Bs, wm0, wn0, lane);
            if (fetch) gload(fetch_into);
            f.mma(acc);
            lstore(store_from, lds + (cur ^ 1) * ASZ, lds + 2 * ASZ + (cur ^ 1) * BSZ);
            __syncthreads();
        };
        const int nmain = (nchunks / DEPTH) * DEPTH;
        for (int j0 = 0; j0 < nmain; j0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) stage(j0 + d, ring[d], ring[(d + 1) % DEPTH], true);
        }
        if (nchunks - nmain >= 1) stage(nmain, ring[0], ring[1], false);
        if (nchunks - nmain >= 2) stage(nmain + 1, ring[1], ring[2], false);
    }
    wgrad_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, mm0, n0, wm0, wn0, lane);
}

template <int BM, int BN, int WM, int WN, int KIND>
bool launch_taps_tile(const ConvArgs& a, dim3 grid, hipStream_t st) {
#define PNP_TAPS(RR, SS)                                                                                                          \
    if (a.R == RR && a.S == SS) {                                                                                                 \
        PnpProfScope ps(prof_class(KIND), st, conv_flops(a), conv_bytes(a), "conv_taps_bf16_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, \
                        WM, WN, KIND, RR, SS);                                                                                     \
        hipLaunchKernelGGL((conv_taps_bf16_kernel<BM, BN, WM, WN, KIND, RR, SS>), grid, dim3(NTHREADS), 0, st, a);                 \
        return true;                                                                                                              \
    }
    PNP_TAPS(3, 3)
    if constexpr (KIND == 0) { PNP_TAPS(5, 5) }
    if constexpr (KIND == 1) { PNP_TAPS(1, 1) PNP_TAPS(1, 2) PNP_TAPS(2, 1) PNP_TAPS(2, 2) PNP_TAPS(2, 3) PNP_TAPS(3, 2) }
#undef PNP_TAPS
    return false;
}

template <int KIND>
bool launch_taps_kind(const ConvArgs& a, int tile, dim3 grid, hipStream_t st) {
    if (tile == 0) return launch_taps_tile<128, 128, 2, 2, KIND>(a, grid, st);
    if (tile == 1) return launch_taps_tile<128, 64, 2, 2, KIND>(a, grid, st);
    if (tile == 2) return launch_taps_tile<128, 32, 4, 1, KIND>(a, grid, st);
    return false;
}

}  // namespace

namespace pnpconv {

bool launch_taps_bf16(const ConvArgs& a, int tile, int kind, dim3 grid, hipStream_t st) {
    if (kind == 0) return launch_taps_kind<0>(a, tile, grid, st);
    if (kind == 1) return launch_taps_kind<1>(a, tile, grid, st);
    return false;
}

bool launch_wgrad_bf16(const ConvArgs& a, int tile, dim3 grid, hipStream_t st) {
#define PNP_WG(BN_, WM_, WN_)                                                                                                     \
    {                                                                                                                             \
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), conv_bytes(a), "conv_wgrad_bf16_kernel<128, %d, %d, %d>", BN_, WM_, WN_); \
        hipLaunchKernelGGL((conv_wgrad_bf16_kernel<128, BN_, WM_, WN_>), grid, dim3(NTHREADS), 0, st, a);                          \
        return true;                                                                                                              \
    }
    if (tile == 0) PNP_WG(128, 2, 2)
    if (tile == 1) PNP_WG(64, 2, 2)
    if (tile == 2) PNP_WG(32, 4, 1)
#undef PNP_WG
    return false;
}

}  // namespace pnpconv
