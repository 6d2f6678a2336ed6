// conv_igemm.hip — NHWC x HWIO implicit-GEMM convolution on gfx950 fp32 matrix cores.
//
// Replaces tf.nn.conv2d / tf.nn.atrous_conv2d (+ tf.pad SYMMETRIC, + tf.nn.dropout) at
// /root/reference/layers.py:18,24,67,73,86,92 and their TF-autodiff gradients.
//
// GEMM view (no im2col buffer is ever materialised):
//   fwd  : Y[M=N*OH*OW][K]   = A[M][R*S*C] * Wm[R*S*C][K]       A gathered from x on the fly
//   dgrad: the same kernel on dy with flipped/transposed filters; stride > 1: one stride-1 correlation per stride phase (plan_phases)
//   wgrad: dW[R*S*C][K]      = At[R*S*C][P=N*OH*OW] * dY[P][K]  split over P across workgroups
//
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32 fma chain, 64 FLOP/clk/SIMD = the 157 TF fp32 peak).
// Workgroup = 4 waves (256 threads); BK = 32 reduction elements per LDS stage; LDS double-buffered; global loads go through
// buffer descriptors (out-of-range offset = hardware zero: no branches for padding / ragged edges); a stage is a 4-slice register
// pipeline (Frag): ds_reads of slice q+1, the next stage's global loads and its LDS stores all fly under MFMAs; 1 barrier / stage.
//
// Kernels:
//   conv_taps_kernel        zero/VALID padding, C % 32 == 0, filter shapes 1x1..3x3 / 5x5, any stride (forward, stride-1 data gradient,
//                           stride-phase sub-filters): taps unrolled, scalar offsets; optional fused inference-BN epilogue
//   conv_fwd_kernel         everything else on the matrix cores (any C, in-kernel SYMMETRIC mirror, filters smaller than the stride)
//   conv_wgrad_kernel       filter gradient; reduction over pixels split across workgroups + splitk_reduce_kernel
//   conv_fwd_narrow_kernel  K <= 16 outputs at stride 1 on the vector ALUs (LDS patch, filters through the scalar cache, v_pk_fma_f32)
//   wgrad_direct(4)_kernel  filter gradient for K <= 16 on the vector ALUs (dy through the scalar cache) + splitk_reduce_many_kernel
//   flip_transpose(_phase)_kernel, splitk_reduce(_scatter/_drop)_kernel, sympad_fwd/bwd_kernel, bn_fold_kernel, naive_conv_kernel
// Host planners: choose_tile / choose_split (dispatch-round filling), plan_phases (strided data gradients), wgrad_plan_split, wgd_plan,
// narrow_fwd_ok — their decisions are visible through the pnp_conv2d_*_workspace_bytes queries (tests/test_abi.py).
//
// LDS layouts (dwords):
//   fwd  A tile  [BM][36]     row = output pixel, 32 k's contiguous (+4 pad). A fragments are read with
//                             ONE ds_read_b128 per 4 MFMAs: lane l takes k = 8q+4*(l>>5)+{0..3}; MFMA j of the
//                             group then contracts k-pair {8q+j, 8q+4+j} (a k-permutation, legal because the
//                             B fragment uses the same pairing).  Stride 36 dwords => 16-B slot = 9*row mod 16,
//                             a bijection over each b128 lane group => conflict-free (SQ_LDS_BANK_CONFLICT = 0 measured).
//   B tile       [32][BN+4]   row = k, n contiguous; fragment = ds_read_b32, lanes 0-31 consecutive => conflict-free.
//   wgrad A tile [32][BM+4]   row = pixel (reduction index), m' contiguous; ds_read_b32 like B.
#include "conv_common.h"
#include "conv_mma.h"

using namespace pnpconv;

// Timing-experiment builds only (make variant NAME=trace EXTRA=-DPNP_TRACE=1): wave 0 of the first 256 workgroups of the two tap-unrolled
// forward kernels writes the shader clock at kernel entry, after the tile set-up, at the start of the main loop, after every stage's
// barrier, and around the epilogue; tools/experiments/stage_trace.py reads it back (pnp_debug_trace_read).  Never in the shipped library.
#ifdef PNP_TRACE
constexpr int TRACE_WG = 512, TRACE_N = 192;
__device__ unsigned long long g_trace[TRACE_WG][TRACE_N];
#define PNP_TRACE_MARK(slot)                                                                             \
    do {                                                                                                 \
        const unsigned tr_ = blockIdx.x - (gridDim.x > 4352u ? 4096u : 0u);      /* big launches: workgroups of the steady state */  \
        if (threadIdx.x == 0 && tr_ < (unsigned)TRACE_WG && (slot) < TRACE_N) g_trace[tr_][(slot)] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int pnp_debug_trace_read(void* host, size_t bytes) {
    if (bytes > sizeof(unsigned long long) * TRACE_WG * TRACE_N) bytes = sizeof(unsigned long long) * TRACE_WG * TRACE_N;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_trace), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define PNP_TRACE_MARK(slot) do { } while (0)
#endif

namespace {

// ---- B tile (weights for fwd, dy for wgrad): row-major [rows][ncols] global matrix ------------
template <int BN, bool VEC>
struct BLoader {
    static constexpr int C4 = BN / 4;             // float4 columns per row
    static constexpr int RP = NTHREADS / C4;      // rows per pass
    static constexpr int NP = BK / RP;            // passes
    static constexpr int LD = BN + 4;
    f32x4 reg[NP];
    int bcol, brow;
    __device__ __forceinline__ void init(int t) {
        bcol = t % C4;
        brow = t / C4;
    }
    // rows [k0, k0+32) of a [nrows][ncols] matrix, columns [n0, n0+BN)
    __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rb, int k0, int nrows, int ncols, int n0) {
        const int n = n0 + 4 * bcol;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int k = k0 + brow + RP * i;
            const bool rok = k < nrows;
            const unsigned base = (unsigned)(k * ncols + n) * 4u;
            if constexpr (VEC) {      // ncols % 4 == 0
                reg[i] = bload4(rb, (rok && n < ncols) ? base : OOB);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) reg[i][e] = bload1(rb, (rok && n + e < ncols) ? base + 4u * e : OOB);
            }
        }
    }
    // Scalar-offset variant (wgrad, dy < 2 GiB): the thread's byte offset inside a 32-row stage never changes, the stage itself is a
    // scalar offset; rows past the end of the matrix are past the buffer -> 0 from the hardware.  No VALU per stage.
    unsigned boff[NP];
    __device__ __forceinline__ void init_s(int ncols, int n0) {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            boff[i] = (n0 + 4 * bcol < ncols) ? (unsigned)(((brow + RP * i) * ncols + n0 + 4 * bcol) * 4) : 0x80000000u;
    }
    __device__ __forceinline__ void load_s(__amdgpu_buffer_rsrc_t rb, int k0, int ncols) {
        const int soff = k0 * ncols * 4;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            reg[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rb, boff[i], soff, 0));
    }
    __device__ __forceinline__ void store(float* lds) const {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            *reinterpret_cast<f32x4*>(lds + (brow + RP * i) * LD + 4 * bcol) = reg[i];
    }
};

// ---- fwd A tile: [BM pixels][32 k] gathered from x ---------------------------------------------
// MODE 0: C % 32 == 0 (whole stage shares one filter tap), 1: C % 4 == 0, 2: any C (scalar)
template <int BM, int MODE, bool UPS>
struct FwdALoader {
    static constexpr int NR = BM / 32;   // rows per thread
    static constexpr int LD = BK + 4;
    f32x4 reg[NR];
    int kg, mrow;
    int pixbase[NR], vh0[NR], vw0[NR];
    unsigned valid;
    __device__ __forceinline__ void init(int t, int m0, const ConvArgs& a) {
        kg = t & 7;
        mrow = t >> 3;
        valid = 0;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            int m = m0 + mrow + 32 * i;
            bool ok = m < a.M;
            if (!ok) m = 0;
            int n, oh, ow;
            split_row(a, m, n, oh, ow);
            pixbase[i] = n * a.H * a.W;
            vh0[i] = oh * a.stride - a.pad_t;
            vw0[i] = ow * a.stride - a.pad_l;
            if (ok) valid |= 1u << i;
        }
    }
    // byte offset of x[pixel(i, tap rs)][c], or OOB when the tap reads padding / the row is past M
    __device__ __forceinline__ unsigned offset(const ConvArgs& a, int i, int rs, int c, bool kok) const {
        int r = rs / a.S;
        int s = rs - r * a.S;
        int ih = 0, iw = 0;
        bool ok = kok && ((valid >> i) & 1u);
        ok = map_coord<UPS>(vh0[i] + r * a.dil, a.H, a.ups, a.pad_mode, ih) & ok;
        ok = map_coord<UPS>(vw0[i] + s * a.dil, a.W, a.ups, a.pad_mode, iw) & ok;
        return ok ? (unsigned)((pixbase[i] + ih * a.W + iw) * a.C + c) * 4u : OOB;
    }
    // MODE 4 (C % 4 == 0, C >= 16, zero padding, no upsampling): (tap row, tap col, channel) of this thread's 4 k's are carried
    // from stage to stage (k advances by 32: at most two channel wraps because C >= 16) instead of being re-derived by two integer
    // divisions, and the zero-padding test is two unsigned compares — ~40 VALU per stage instead of ~150.  The C = 16 layers at
    // 256^2 and the 40 -> 5 output convolution are VALU-bound on this arithmetic, not on the MFMAs.
    int i_k, i_c, i_s, i_r;
    int rowoff[NR];
    __device__ __forceinline__ void init4(const ConvArgs& a, int k_first) {
        i_k = k_first + 4 * kg;
        const int rs = i_k / a.C;
        i_c = i_k - rs * a.C;
        i_r = rs / a.S;
        i_s = rs - i_r * a.S;
#pragma unroll
        for (int i = 0; i < NR; ++i) rowoff[i] = (pixbase[i] + vh0[i] * a.W + vw0[i]) * a.C;
    }
    __device__ __forceinline__ void load4(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx) {
        const bool kok = i_k < a.Kred;
        const int dh = i_r * a.dil, dw = i_s * a.dil;
        const int tsh = (dh * a.W + dw) * a.C + i_c;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const bool ok = kok & (((valid >> i) & 1u) != 0) & ((unsigned)(vh0[i] + dh) < (unsigned)a.H) &
                            ((unsigned)(vw0[i] + dw) < (unsigned)a.W);
            reg[i] = bload4(rx, ok ? (unsigned)((rowoff[i] + tsh) * 4) : OOB);
        }
        i_k += BK;
        i_c += BK;
#pragma unroll
        for (int w = 0; w < 2; ++w) {
            const bool c1 = i_c >= a.C;
            i_c -= c1 ? a.C : 0;
            i_s += c1 ? 1 : 0;
            const bool c2 = i_s >= a.S;
            i_s -= c2 ? a.S : 0;
            i_r += c2 ? 1 : 0;
        }
    }
    __device__ __forceinline__ void load(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx, int k0) {
        if constexpr (MODE == 4) {          // stages are visited in order: the carried state IS k0
            load4(a, rx);
            return;
        }
        if constexpr (MODE == 0) {
            int rs = k0 / a.C;                 // block-uniform
            int c = k0 - rs * a.C + 4 * kg;
            const bool kok = k0 < a.Kred;
#pragma unroll
            for (int i = 0; i < NR; ++i) reg[i] = bload4(rx, offset(a, i, rs, c, kok));
        } else if constexpr (MODE == 1) {
            int kf = k0 + 4 * kg;
            int rs = kf / a.C;
            int c = kf - rs * a.C;
            const bool kok = kf < a.Kred;      // C%4==0 => Kred%4==0 => whole float4 in range
#pragma unroll
            for (int i = 0; i < NR; ++i) reg[i] = bload4(rx, offset(a, i, rs, c, kok));
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int kf = k0 + 4 * kg + e;
                    int rs = kf / a.C;
                    reg[i][e] = bload1(rx, offset(a, i, rs, kf - rs * a.C, kf < a.Kred));
                }
        }
    }
    __device__ __forceinline__ void store(float* lds) const {
#pragma unroll
        for (int i = 0; i < NR; ++i)
            *reinterpret_cast<f32x4*>(lds + (mrow + 32 * i) * LD + 4 * kg) = reg[i];
    }
};

// ---- wgrad A tile: [32 pixels][BM m'] gathered from x, m' = (r*S+s)*C + c ----------------------
// MODE 1: C % 4 == 0 (a thread's 4 consecutive m' share the tap), 2: any C
template <int BM, int MODE>
struct WgradALoader {
    static constexpr int C4 = BM / 4;
    static constexpr int RP = NTHREADS / C4;
    static constexpr int NP = BK / RP;
    static constexpr int LD = BM + 4;
    f32x4 reg[NP];
    int acol, arow;
    int rs_u, c_u;          // MODE 1: this thread's tap and channel (fixed for the whole kernel)
    bool mok;               // MODE 1: m' in range
    int mm;                 // first m' of this thread
    // MODE 3 (stride 1, zero padding, OW >= 32, C % 4 == 0): the reduction index p (output pixel) advances by 32 per stage, so the
    // pixel coordinates and the input byte offset are carried incrementally — no integer division and ~12 VALU per row per stage
    // instead of ~70 (three divisions) in the general path.
    int l_ow[NP], l_oh[NP], l_off[NP];
    int l_dh, l_dw;         // tap displacement: r*dil - pad_t, s*dil - pad_l
    __device__ __forceinline__ void init(int t, int mm0, const ConvArgs& a, int p_first = 0) {
        acol = t % C4;
        arow = t / C4;
        mm = mm0 + 4 * acol;
        mok = mm < a.Kred;
        int m_ = mok ? mm : 0;
        rs_u = m_ / a.C;
        c_u = m_ - rs_u * a.C;
        if constexpr (MODE == 3) {
            const int r = rs_u / a.S, s = rs_u - r * a.S;
            l_dh = r * a.dil - a.pad_t;
            l_dw = s * a.dil - a.pad_l;
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int p = p_first + arow + RP * i;
                const int n = p / a.OHW;
                const int rem = p - n * a.OHW;
                l_oh[i] = rem / a.OW;
                l_ow[i] = rem - l_oh[i] * a.OW;
                l_off[i] = (((n * a.H + l_oh[i] + l_dh) * a.W + l_ow[i] + l_dw) * a.C + c_u) * 4;
            }
        }
    }
    // MODE 3: fetch the rows at the current position, then advance every row by 32 output pixels
    __device__ __forceinline__ void load_advance(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx, int pend) {
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            // (no test against the end of the pixel range: splits end on stage boundaries, and rows past the LAST pixel either land
            // past the end of x -> 0 from the hardware, or meet a dy row past the end of dy -> 0)
            const bool ok = mok & ((unsigned)(l_oh[i] + l_dh) < (unsigned)a.H) & ((unsigned)(l_ow[i] + l_dw) < (unsigned)a.W);
            reg[i] = bload4(rx, ok ? (unsigned)l_off[i] : OOB);
            l_ow[i] += BK;
            l_off[i] += BK * a.C * 4;
            const bool ww = l_ow[i] >= a.OW;                    // at most one wrap per step because OW >= 32
            l_ow[i] -= ww ? a.OW : 0;
            l_oh[i] += ww ? 1 : 0;
            l_off[i] += ww ? (a.W - a.OW) * a.C * 4 : 0;
            const bool hw = l_oh[i] >= a.OH;
            l_oh[i] -= hw ? a.OH : 0;
            l_off[i] += hw ? (a.H - a.OH) * a.W * a.C * 4 : 0;
        }
    }
    __device__ __forceinline__ unsigned offset(const ConvArgs& a, int p, int rs, int c, bool ok) const {
        int n = p / a.OHW;
        int rem = p - n * a.OHW;
        int oh = rem / a.OW;
        int ow = rem - oh * a.OW;
        int r = rs / a.S;
        int s = rs - r * a.S;
        int ih = 0, iw = 0;
        ok = map_coord<false>(oh * a.stride - a.pad_t + r * a.dil, a.H, 1, a.pad_mode, ih) & ok;
        ok = map_coord<false>(ow * a.stride - a.pad_l + s * a.dil, a.W, 1, a.pad_mode, iw) & ok;
        return ok ? (unsigned)(((n * a.H + ih) * a.W + iw) * a.C + c) * 4u : OOB;
    }
    __device__ __forceinline__ void load(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx, int p0, int pend) {
        if constexpr (MODE == 3) {          // stages are visited in order: the carried state IS p0
            load_advance(a, rx, pend);
            return;
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = p0 + arow + RP * i;
            const bool pok = p < pend;
            const int pp = pok ? p : 0;
            if constexpr (MODE <= 1 || MODE == 3) {
                reg[i] = bload4(rx, offset(a, pp, rs_u, c_u, pok && mok));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m_ = mm + e;
                    const int rs = m_ / a.C;
                    reg[i][e] = bload1(rx, offset(a, pp, rs, m_ - rs * a.C, pok && m_ < a.Kred));
                }
            }
        }
    }
    __device__ __forceinline__ void store(float* lds) const {
#pragma unroll
        for (int i = 0; i < NP; ++i)
            *reinterpret_cast<f32x4*>(lds + (arow + RP * i) * LD + 4 * acol) = reg[i];
    }
};

// Split accumulation (narrow tiles): a wave with TM*TN <= 2 accumulator tiles rotates through only two dependent MFMA chains; with a
// second accumulator set for the odd k of every 8-group it rotates through four, like the 2x2 tile (summed once in front of the epilogue).
#ifndef PNP_SPLIT_ACC
#define PNP_SPLIT_ACC 0
#endif
template <int TM, int TN, bool A_MMAJOR, int LDA, int LDB>
__device__ __forceinline__ void frag_mma2(const Frag<TM, TN, A_MMAJOR, LDA, LDB>& f, Acc<TM, TN>& acc0, Acc<TM, TN>& acc1) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (j & 1) acc1.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][j], f.b[j][tn], acc1.v[tm][tn], 0, 0, 0);
                else acc0.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[tm][j], f.b[j][tn], acc0.v[tm][tn], 0, 0, 0);
            }
}
template <int TM, int TN>
__device__ __forceinline__ void acc_add(Acc<TM, TN>& a, const Acc<TM, TN>& b) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) a.v[i][j][e] += b.v[i][j][e];
}
// ================================ forward / dgrad kernel ========================================
// KIND 0 = forward, 1 = data gradient of a stride-1 convolution, 2 = data gradient of a strided convolution (input = dy
// zero-upsampled by `ups`).  0 and 1 run the same code; the distinct symbol lets rocprof separate forward from backward
// launches.  Only KIND 2 carries the integer division of the upsampling test.
template <int BM, int BN, int WM, int WN, int MODE, int KIND, bool VECB>
__device__ __forceinline__ void conv_fwd_body(ConvArgs a, const int blk) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BK + 4, LDB = BN + 4;
    constexpr int ASZ = BM * LDA, BSZ = BK * LDB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

    if constexpr (KIND != 2) a.ups = 1;
    // Two workgroups share a CU (one wave each per SIMD).  Their stage loops run at the same rate, so whatever phase offset they
    // start with persists; dispatched together they stay aligned and the matrix pipe idles whenever both are in the non-MFMA part of
    // a stage (barrier, first fragment reads, LDS stores).  Starting every second dispatch wave late by about that part's length
    // de-phases them for the whole kernel.  (Speed only: nothing depends on which workgroups share a CU.)
    if (a.stagger > 0 && ((blk >> 8) & 1)) {
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(16);
    }
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = a.nblk_m * a.nblk_n;
    const int z = blk / nblk;                      // reduction split (data gradients only; a.nsplit == 1 otherwise)
    int bid = blk - z * nblk;
    if (a.xcd_swizzle) bid = xcd_remap(bid, nblk);
    int mt, nt;
    tile_coords(bid, a.nblk_m, a.nblk_n, a.gn, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    FwdALoader<BM, MODE, KIND == 2> la;
    BLoader<BN, VECB> lb;
    la.init(t, m0, a);
    if constexpr (MODE == 4) la.init4(a, z * a.chunks_per_split * BK);
    lb.init(t);

    Acc<TM, TN> acc;
    acc.zero();

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    const int nchunks_total = (a.Kred + BK - 1) / BK;
    const int c_begin = z * a.chunks_per_split;
    const int c_end = (c_begin + a.chunks_per_split < nchunks_total) ? c_begin + a.chunks_per_split : nchunks_total;
    const int nchunks = c_end - c_begin;
    // Reduction order.  MODE 0 (C % 32 == 0): channel-group major, filter tap minor — the R*S stages that re-read the same input
    // pixels (shifted by one tap) run back to back, so the re-reads hit the XCD's L2.  The sum is order-independent up to fp32
    // rounding; the filter rows are visited in the matching order.  Stage index past c_end: every offset out of range -> zeros.
    auto stage_k0 = [&](int ci) {
        int k0 = ci * BK;
        if constexpr (MODE == 0) {
            const int rs_n = a.R * a.S;
            const int cc = ci / rs_n;
            k0 = (ci - cc * rs_n) * a.C + cc * BK;
        }
        return (ci < c_end) ? k0 : a.Kred;
    };
    {
        const int k0 = stage_k0(c_begin);
        la.load(a, rx, k0);
        lb.load(rw, k0, a.Kred, a.K, n0);
    }
    la.store(lds);
    lb.store(lds + 2 * ASZ);
    __syncthreads();
    Frag<TM, TN, true, LDA, LDB> f0, f1;
    f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
    // Main loop — ONE basic block per stage (no branches: the prefetch past the last stage reads out of range = zeros
    // and lands in the LDS buffer nobody reads).  Per stage and wave: 64 MFMAs in 4 slices of 16; under slice q the
    // ds_reads of slice q+1 are in flight; the next stage's global loads are issued under slice 0 and written to the
    // other LDS buffer under slice 3; only the 6 reads of the next stage's slice 0 are exposed after the barrier.
    // PNP_CONV_ABLATE (compile-time, timing experiments only — results are wrong when set):
    //   1 = no global loads after the first stage, 2 = no MFMAs, 4 = no LDS stores, 8 = no barrier
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        const float* As = lds + cur * ASZ;
        const float* Bs = lds + 2 * ASZ + cur * BSZ;
        float* An = lds + (cur ^ 1) * ASZ;
        float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
        const int k0 = stage_k0(c_begin + c + 1);
        // ---- slice 0
        f1.load(As, Bs, 1, wm0, wn0, lane);
        if constexpr (!(kAblate & 1)) {
            la.load(a, rx, k0);
            lb.load(rw, k0, a.Kred, a.K, n0);
        }
        if constexpr (!(kAblate & 2)) f0.mma(acc);
        // interleave: the 6 ds_reads first, then per MFMA a handful of the address VALU/SALU and, when its address is
        // ready, one buffer_load — the ~130 address instructions of the next stage ride in the shadow of the 16 MFMAs
        // (64 cycles each) instead of in front of them.
        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
        for (int i = 0; i < 4 * TM * TN; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x006, 10, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        PNP_SCHED_FENCE();
        // ---- slice 1
        f0.load(As, Bs, 2, wm0, wn0, lane);
        PNP_SCHED_FENCE();
        if constexpr (!(kAblate & 2)) f1.mma(acc);
        PNP_SCHED_FENCE();
        // ---- slice 2
        f1.load(As, Bs, 3, wm0, wn0, lane);
        PNP_SCHED_FENCE();
        if constexpr (!(kAblate & 2)) f0.mma(acc);
        PNP_SCHED_FENCE();
        // ---- slice 3: MFMAs first, THEN the LDS stores of the next stage.  The stores need the global loads issued under slice 0;
        // measured load-to-use latency under this traffic is ~3000 cycles (ablation: load->LDS->barrier chain alone 2.3 us per
        // stage), so waiting for them before the slice-3 MFMAs (2100 cycles after issue) stalled the matrix pipe ~20 %.
        if constexpr (!(kAblate & 2)) f1.mma(acc);
        PNP_SCHED_FENCE();
        if constexpr (!(kAblate & 4)) {
            la.store(An);
            lb.store(Bn);
        }
        PNP_SCHED_FENCE();
        if constexpr (!(kAblate & 8)) __syncthreads();
        f0.load(An, Bn, 0, wm0, wn0, lane);
    }

    conv_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, m0, n0, wm0, wn0, lane, mt * WM + wave / WN, z == 0);
}

template <int BM, int BN, int WM, int WN, int MODE, int KIND, bool VECB>
__global__ void __launch_bounds__(NTHREADS, 2) conv_fwd_kernel(ConvArgs a) {
    conv_fwd_body<BM, BN, WM, WN, MODE, KIND, VECB>(a, (int)blockIdx.x);
}

// ---- all stride phases of a strided data gradient in ONE launch (small feature maps) -------------------------------------------------
// A k5 s4 critic layer on a 16^2 -> 4^2 map has 16 phases of 256 GEMM rows each: one launch per phase (16 workgroups, reduction split
// 32 ways, + a scatter-reduce) cost 16 x (17 + 4.5) us for 3.4 GFLOP.  Here the workgroups of all phases share one grid; a workgroup
// looks up its phase (sub-filter extent / padding / output lattice / filter block) in the kernel arguments and runs the general
// C % 32 == 0 stage loop, writing its rows straight to the phase's pixels (no reduction split, no second kernel).
struct PhaseDesc {
    int w_off;              // float offset of the phase's flipped sub-filter [T][U][K][C]
    int R, S, OH, OW, pad_t, pad_l, o_h0, o_w0;
    int first_blk, nblk_m;
};
struct GroupArgs {
    ConvArgs base;          // everything the phases share (x = dy, y = dx, H/W = dy's extent, C, K, output lattice o_s/o_H/o_W, tiles in N)
    PhaseDesc ph[16];
    int nph;
};

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(NTHREADS, 2) conv_dgrad_phases_kernel(GroupArgs g) {
    int p = 0;
    for (int i = 1; i < g.nph; ++i) p = ((int)blockIdx.x >= g.ph[i].first_blk) ? i : p;
    const PhaseDesc d = g.ph[p];
    ConvArgs a = g.base;
    a.w = g.base.w + d.w_off;
    a.R = d.R; a.S = d.S; a.OH = d.OH; a.OW = d.OW; a.pad_t = d.pad_t; a.pad_l = d.pad_l;
    a.o_h0 = d.o_h0; a.o_w0 = d.o_w0;
    a.OHW = d.OH * d.OW;
    a.ow_sh = a.ohw_sh = -1;                  // per-phase extents: divide
    a.M = a.N * a.OHW;
    a.Kred = d.R * d.S * a.C;
    a.w_bytes = (unsigned)(a.Kred * a.K) * 4u;
    a.nblk_m = d.nblk_m;
    a.nsplit = 1;
    a.chunks_per_split = (a.Kred + BK - 1) / BK;
    conv_fwd_body<BM, BN, WM, WN, 0, 1, true>(a, (int)blockIdx.x - d.first_blk);
}

// ============ forward (any stride) / stride-1 data gradient, zero padding, C % 32 == 0, RxS filter: taps unrolled ============
// The common case on the reference path (every 3x3 SAME conv from group_2 on and their dgrads, the strided 3x3 / 5x5 critic convs,
// and the stride-phase sub-filters 1x1 ... 3x3 of their data gradients).  With the filter taps unrolled at compile time the per-stage
// address work collapses:
//   A:  voffset = pixel_base(row) + tap_shift   (1 v_add + 1 v_cndmask per row; validity = one bit per (row, tap), set up once)
//       soffset = channel-group * 128 bytes      (scalar)
//   B:  voffset = constant per thread, soffset = (tap*C + channel-group*32) * K * 4   (scalar)
// i.e. ~16 VALU per stage instead of ~130 (the general kernel re-derives tap, mirror/zero test and pixel offset per row per stage).

// Waves per SIMD the register allocation must leave room for.  The 128x64 / 128x32 tiles take 54 / 46 KB of LDS — three workgroups fit a
// CU's 160 KB — but at 175 VGPRs only two waves fit a SIMD; asking for three (<= 168 VGPRs) gives every SIMD a third wave to issue MFMAs
// while the other two sit in the per-stage barrier / LDS-store part (a 128x64 stage carries half the MFMAs of a 128x128 one for the same
// fixed part).  PNP_TAPS_NARROW_WAVES = 2 restores the round-2 allocation (A/B: tools/experiments/).
#ifndef PNP_TAPS_NARROW_WAVES
#define PNP_TAPS_NARROW_WAVES 2
#endif
template <int BM, int BN, int WM, int WN, int KIND, int R, int S>
__global__ void __launch_bounds__(NTHREADS, ((BN <= 64 && R * S <= 9) ? PNP_TAPS_NARROW_WAVES : 2)) conv_taps_kernel(ConvArgs a) {
    constexpr int NTAP = R * S;
    static_assert(NTAP <= 32, "one validity bit per tap");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BK + 4, LDB = BN + 4;
    constexpr int ASZ = BM * LDA, BSZ = BK * LDB;
    constexpr int NR = BM / 32;
    constexpr int C4 = BN / 4, RPB = NTHREADS / C4, NPB = BK / RPB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

    PNP_TRACE_MARK(0);
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

    // ---- per-thread A rows: byte offset of (pixel shifted by -pad) and the validity bit of every tap -------------------
    const int kg = t & 7, mrow = t >> 3;
    int abase[NR];
    unsigned amask[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int m = m0 + mrow + 32 * i;
        const bool ok = m < a.M;
        if (!ok) m = 0;
        int n, oh, ow;
        split_row(a, m, n, oh, ow);
        const int vh0 = oh * a.stride - a.pad_t, vw0 = ow * a.stride - a.pad_l;
        abase[i] = (((n * a.H + vh0) * a.W + vw0) * a.C + 4 * kg) * 4;
        // tap (r, s) is valid iff its row AND its column are inside the image: R + S tests and R shifted ORs instead of R*S full tests
        unsigned wm = 0, mk = 0;
#pragma unroll
        for (int sx = 0; sx < S; ++sx) wm |= ((unsigned)(vw0 + sx * a.dil) < (unsigned)a.W ? 1u : 0u) << sx;
#pragma unroll
        for (int ry = 0; ry < R; ++ry) mk |= ((unsigned)(vh0 + ry * a.dil) < (unsigned)a.H ? wm : 0u) << (ry * S);
        amask[i] = ok ? mk : 0u;
    }
    // ---- per-thread B rows ------------------------------------------------------------------------------------------
    const int bcol = t % C4, brow = t / C4;
    unsigned boff[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
        boff[i] = (n0 + 4 * bcol < a.K) ? (unsigned)(((brow + RPB * i) * a.K + n0 + 4 * bcol) * 4) : OOB2;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    f32x4 areg[NR], breg[NPB];
    auto gload = [&](int cc, int tap) {     // tap is a compile-time constant at every call site (unrolled)
        const int tshift = (((tap / S) * a.dil * a.W + (tap % S) * a.dil) * a.C) * 4;
        const int sa = cc * (BK * 4);
        const int sb = ((tap * a.C + cc * BK) * a.K) * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned vo = ((amask[i] >> tap) & 1u) ? (unsigned)(abase[i] + tshift) : OOB2;
            areg[i] = bload4s(rx, vo, sa);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) breg[i] = bload4s(rw, boff[i], sb);
    };
    auto lstore = [&](float* An, float* Bn) {
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4*>(An + (mrow + 32 * i) * LDA + 4 * kg) = areg[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) *reinterpret_cast<f32x4*>(Bn + (brow + RPB * i) * LDB + 4 * bcol) = breg[i];
    };

    Acc<TM, TN> acc;
    acc.zero();

    // reduction range of this workgroup in channel groups (a.chunks_per_split is a multiple of NTAP on this path)
    const int ncc_total = a.C / BK;
    const int cc_begin = z * (a.chunks_per_split / NTAP);
    int cc_end = cc_begin + a.chunks_per_split / NTAP;
    if (cc_end > ncc_total) cc_end = ncc_total;

    PNP_TRACE_MARK(1);
    gload(cc_begin, 0);
    lstore(lds, lds + 2 * ASZ);
    __syncthreads();
    PNP_TRACE_MARK(2);
    Frag<TM, TN, true, LDA, LDB> f0, f1;
    f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
    int sc = 0;   // stage counter (LDS buffer parity)
    for (int cc = cc_begin; cc < cc_end; ++cc) {
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int cur = sc & 1;
            ++sc;
            const float* As = lds + cur * ASZ;
            const float* Bs = lds + 2 * ASZ + cur * BSZ;
            float* An = lds + (cur ^ 1) * ASZ;
            float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
            // next stage: next tap of this channel group, or tap 0 of the next group; past the end the last group is simply
            // fetched again (valid addresses, lands in the LDS buffer nobody reads)
            const int ntap = (tap + 1 < NTAP) ? tap + 1 : 0;
            int ncc = (tap + 1 < NTAP) ? cc : cc + 1;
            ncc = (ncc < cc_end) ? ncc : cc;
            // ---- slice 0 (same 4-slice register pipeline as conv_fwd_kernel)
            f1.load(As, Bs, 1, wm0, wn0, lane);
            gload(ncc, ntap);
            f0.mma(acc);
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int i = 0; i < 4 * TM * TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            PNP_SCHED_FENCE();
            PNP_SLICE(f0.load(As, Bs, 2, wm0, wn0, lane), f1.mma(acc), 4 * TM * TN, (TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            PNP_SLICE2(f1.load(As, Bs, 3, wm0, wn0, lane), f0.mma(acc), lstore(An, Bn), 4 * TM * TN, (TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            PNP_LAST_SLICE(f1.mma(acc), lstore(An, Bn), 4 * TM * TN)
            __syncthreads();
            PNP_TRACE_MARK(3 + sc);
            f0.load(An, Bn, 0, wm0, wn0, lane);
        }
    }
    PNP_TRACE_MARK(4 + sc);

    conv_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, m0, n0, wm0, wn0, lane, mt * WM + wave / WN, z == 0);
    PNP_TRACE_MARK(5 + sc);
}

// ---- the same tap-unrolled convolution with THREE LDS stages and a register ring of two global-load stages (narrow tiles) ------------
// A 128x64 stage carries 32 MFMAs per wave — half of a 128x128 stage — for the same fixed part: barrier, the fragment reads of the next
// stage's first slice that can only start behind it, and a global-load-to-LDS-store distance of three slices (24 MFMAs of cover instead of
// 48).  PMC on the two-stage kernel: SQ_WAIT_INST_LDS per MFMA 1.05 against 0.50 on the wide tile, matrix-pipe utilisation 0.71 against
// 0.92.  With a third LDS stage (3 x 27 KB = 81 408 B: two workgroups still fit a CU's 160 KB, to the byte) the barrier of stage s
// publishes the data of stage s+2, so everything stage s+1 needs is already visible while stage s runs:
//   * the first-slice fragment reads of stage s+1 are issued under the LAST slice of stage s, before the barrier — nothing is exposed
//     behind it;
//   * global loads run TWO stages ahead of their LDS store (ring of two register sets: one whole stage more cover).
// Buffer rotation: stage s computes from LDS stage s % 3, stores the registers that hold stage s+2 into (s+2) % 3 (last read in stage
// s-1: every wave is past that stage's barrier), and requests stage s+3 from memory.  The ring index must be a compile-time constant, so
// the channel-group loop is unrolled by two where the tap count is odd; the host only picks this kernel when the workgroup's channel
// groups come in pairs then (C % 64 == 0 and an un-split or evenly split reduction: every layer of the model that runs narrow tiles).
template <int BM, int BN, int WM, int WN, int KIND, int R, int S>
__global__ void __launch_bounds__(NTHREADS, 2) conv_taps3_kernel(ConvArgs a) {
    constexpr int NTAP = R * S;
    static_assert(NTAP <= 32, "one validity bit per tap");
    constexpr int CCU = (NTAP & 1) ? 2 : 1;          // channel groups per loop trip: an even number of stages
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BK + 4, LDB = BN + 4;
    constexpr int ASZ = BM * LDA, BSZ = BK * LDB, STG = ASZ + BSZ;
    constexpr int NR = BM / 32;
    constexpr int C4 = BN / 4, RPB = NTHREADS / C4, NPB = BK / RPB;
    static_assert(3 * STG * 4 <= 81920, "three stages must leave room for two workgroups per CU");
    __shared__ __attribute__((aligned(16))) float lds[3 * STG];

    PNP_TRACE_MARK(0);
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

    const int kg = t & 7, mrow = t >> 3;
    int abase[NR];
    unsigned amask[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        int m = m0 + mrow + 32 * i;
        const bool ok = m < a.M;
        if (!ok) m = 0;
        int n, oh, ow;
        split_row(a, m, n, oh, ow);
        const int vh0 = oh * a.stride - a.pad_t, vw0 = ow * a.stride - a.pad_l;
        abase[i] = (((n * a.H + vh0) * a.W + vw0) * a.C + 4 * kg) * 4;
        // tap (r, s) is valid iff its row AND its column are inside the image: R + S tests and R shifted ORs instead of R*S full tests
        unsigned wm = 0, mk = 0;
#pragma unroll
        for (int sx = 0; sx < S; ++sx) wm |= ((unsigned)(vw0 + sx * a.dil) < (unsigned)a.W ? 1u : 0u) << sx;
#pragma unroll
        for (int ry = 0; ry < R; ++ry) mk |= ((unsigned)(vh0 + ry * a.dil) < (unsigned)a.H ? wm : 0u) << (ry * S);
        amask[i] = ok ? mk : 0u;
    }
    const int bcol = t % C4, brow = t / C4;
    unsigned boff[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
        boff[i] = (n0 + 4 * bcol < a.K) ? (unsigned)(((brow + RPB * i) * a.K + n0 + 4 * bcol) * 4) : OOB2;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    struct Regs {
        f32x4 a[NR];
        f32x4 b[NPB];
    };
    Regs ring[2];
    auto gload = [&](Regs& r, int cc, int tap) {     // tap is a compile-time constant at every call site (unrolled)
        const int tshift = (((tap / S) * a.dil * a.W + (tap % S) * a.dil) * a.C) * 4;
        const int sa = cc * (BK * 4);
        const int sb = ((tap * a.C + cc * BK) * a.K) * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned vo = ((amask[i] >> tap) & 1u) ? (unsigned)(abase[i] + tshift) : OOB2;
            r.a[i] = bload4s(rx, vo, sa);
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i) r.b[i] = bload4s(rw, boff[i], sb);
    };
    auto lstore = [&](const Regs& r, float* An, float* Bn) {
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4*>(An + (mrow + 32 * i) * LDA + 4 * kg) = r.a[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) *reinterpret_cast<f32x4*>(Bn + (brow + RPB * i) * LDB + 4 * bcol) = r.b[i];
    };

    constexpr bool SPLIT = PNP_SPLIT_ACC != 0 && TM * TN <= 2;
    Acc<TM, TN> acc, accb;
    acc.zero();
    if constexpr (SPLIT) accb.zero();
#define PNP_T3_MMA(F) do { if constexpr (SPLIT) frag_mma2(F, acc, accb); else F.mma(acc); } while (0)

    const int ncc_total = a.C / BK;
    const int cc_begin = z * (a.chunks_per_split / NTAP);
    int cc_end = cc_begin + a.chunks_per_split / NTAP;
    if (cc_end > ncc_total) cc_end = ncc_total;
    const int cc_last = cc_end - 1;
    // (stage index -> (channel group, tap)) d stages after (cc, tap); past the end the last group is fetched again (never contracted)
#define PNP_T3_CC(cc, tap, d) (((cc) + ((tap) + (d)) / NTAP) < cc_end ? ((cc) + ((tap) + (d)) / NTAP) : cc_last)
#define PNP_T3_TAP(tap, d) (((tap) + (d)) % NTAP)

    // prologue: stages 0 and 1 into LDS, stage 2 in flight in ring[0]
    PNP_TRACE_MARK(1);
    int tsc = 0;
    (void)tsc;
    gload(ring[0], cc_begin, 0);
    gload(ring[1], PNP_T3_CC(cc_begin, 0, 1), PNP_T3_TAP(0, 1));
    lstore(ring[0], lds, lds + ASZ);
    gload(ring[0], PNP_T3_CC(cc_begin, 0, 2), PNP_T3_TAP(0, 2));
    lstore(ring[1], lds + STG, lds + STG + ASZ);
    __syncthreads();
    // Fragment ring of four 8-k slices: slice q of a stage is contracted while slice q+2 is being read — two slices (16 MFMAs of a 2x1
    // wave tile) of cover for every LDS read instead of one, and the lookahead runs across the stage boundary (the next stage is visible)
    Frag<TM, TN, true, LDA, LDB> fr[4];
    fr[0].load(lds, lds + ASZ, 0, wm0, wn0, lane);
    fr[1].load(lds, lds + ASZ, 1, wm0, wn0, lane);
    PNP_TRACE_MARK(2);
    int o_cur = 0, o_nxt = STG, o_st = 2 * STG;        // float offsets of the LDS stages holding s, s+1 and receiving s+2
    constexpr int NMF = 4 * TM * TN;
    constexpr int NDS = (TM + 4 * TN + NMF - 1) / NMF;
    for (int cc0 = cc_begin; cc0 < cc_end; cc0 += CCU) {
#pragma unroll
        for (int u = 0; u < CCU; ++u) {
            const int cc = cc0 + u;
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int par = (u * NTAP + tap) & 1;            // compile-time after unrolling: stage parity inside the trip
                const float* As = lds + o_cur;
                const float* Bs = As + ASZ;
                const float* An = lds + o_nxt;
                const float* Bn = An + ASZ;
                float* Ast = lds + o_st;
                float* Bst = Ast + ASZ;
                // ---- slice 0: fragments of slice 2, the global loads of stage s+3 into the ring slot stored LAST stage
                fr[2].load(As, Bs, 2, wm0, wn0, lane);
                gload(ring[par ^ 1], PNP_T3_CC(cc, tap, 3), PNP_T3_TAP(tap, 3));
                PNP_T3_MMA(fr[0]);
                __builtin_amdgcn_sched_group_barrier(0x100, TM + 4 * TN, 0);
#pragma unroll
                for (int i = 0; i < NMF; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x006, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                PNP_SCHED_FENCE();
                PNP_SLICE(fr[3].load(As, Bs, 3, wm0, wn0, lane), PNP_T3_MMA(fr[1]), NMF, NDS)
                // ---- slices 2, 3: the NEXT stage's slices 0, 1 (visible since the previous barrier); the LDS stores of stage s+2 ride
                // behind the last MFMAs; the barrier then has nothing to wait for but the slowest wave
                PNP_SLICE(fr[0].load(An, Bn, 0, wm0, wn0, lane), PNP_T3_MMA(fr[2]), NMF, NDS)
                fr[1].load(An, Bn, 1, wm0, wn0, lane);
                PNP_SCHED_FENCE();
                PNP_T3_MMA(fr[3]);
                lstore(ring[par], Ast, Bst);
                __builtin_amdgcn_sched_group_barrier(0x008, NMF / 2, 0);
#pragma unroll
                for (int i = 0; i < NMF / 2; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                PNP_SCHED_FENCE();
                __syncthreads();
                ++tsc;
                PNP_TRACE_MARK(2 + tsc);
                const int o_t = o_cur;
                o_cur = o_nxt;
                o_nxt = o_st;
                o_st = o_t;
            }
        }
    }
#undef PNP_T3_CC
#undef PNP_T3_TAP
#undef PNP_T3_MMA
    if constexpr (SPLIT) acc_add(acc, accb);

    PNP_TRACE_MARK(3 + tsc);
    conv_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, m0, n0, wm0, wn0, lane, mt * WM + wave / WN, z == 0);
    PNP_TRACE_MARK(4 + tsc);
}

// ===================================== wgrad kernel ============================================
// a.x = x, a.w = dy ([P][K]), a.y = dW or the split workspace.  grid.x = nblk_m*nblk_n*nsplit
template <int BM, int BN, int WM, int WN, int MODE, bool VECB>
__global__ void __launch_bounds__(NTHREADS, 2) conv_wgrad_kernel(ConvArgs a) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = a.nblk_m * a.nblk_n;
    // XCD-aware order: all (tap, channel, filter) tiles of one pixel range z re-read the same x and dy rows, so they are given
    // consecutive logical ids = one XCD's L2 (the 40->5 output conv's wgrad fetched 2 GB per launch for 190 MB of operands when its 8
    // tiles per range were sprayed over the 8 XCDs).
    const int lid = a.xcd_swizzle ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int z = lid / nblk;
    int bid = lid - z * nblk;
    int mt, nt;
    tile_coords(bid, a.nblk_m, a.nblk_n, a.gn, mt, nt);
    const int mm0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    WgradALoader<BM, MODE> la;
    BLoader<BN, VECB> lb;
    la.init(t, mm0, a, z * a.chunks_per_split * BK);
    lb.init(t);
    constexpr bool BS = (MODE == 3) && VECB;       // host: dy < 2 GiB on this path
    if constexpr (BS) lb.init_s(a.K, n0);

    Acc<TM, TN> acc;
    acc.zero();

    const int P = a.M;
    const int nchunks_total = (P + BK - 1) / BK;
    const int c_begin = z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;
    const int nchunks = c_end - c_begin;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    if (nchunks > 0) {
        const int pend = (c_end * BK < P) ? c_end * BK : P;     // this split's pixel range ends here
        la.load(a, rx, c_begin * BK, pend);
        if constexpr (BS) lb.load_s(rw, c_begin * BK, a.K);
        else lb.load(rw, c_begin * BK, pend, a.K, n0);
        la.store(lds);
        lb.store(lds + 2 * ASZ);
        __syncthreads();
        Frag<TM, TN, false, LDA, LDB> f0, f1;
        f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
        for (int c = 0; c < nchunks; ++c) {      // same branch-free 4-slice pipeline as conv_fwd_kernel
            const int cur = c & 1;
            const float* As = lds + cur * ASZ;
            const float* Bs = lds + 2 * ASZ + cur * BSZ;
            float* An = lds + (cur ^ 1) * ASZ;
            float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
            const int p0 = (c_begin + c + 1) * BK;                 // >= pend on the last stage -> zeros
            f1.load(As, Bs, 1, wm0, wn0, lane);
            la.load(a, rx, p0, pend);
            if constexpr (BS) lb.load_s(rw, p0, a.K);
            else lb.load(rw, p0, pend, a.K, n0);
            f0.mma(acc);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);      // slice-1 fragment reads first
#pragma unroll
            for (int i = 0; i < 4 * TM * TN; ++i) {                     // then address arithmetic + loads under the MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            PNP_SCHED_FENCE();
            PNP_SLICE(f0.load(As, Bs, 2, wm0, wn0, lane), f1.mma(acc), 4 * TM * TN, (4 * TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            PNP_SLICE2(f1.load(As, Bs, 3, wm0, wn0, lane), f0.mma(acc), (la.store(An), lb.store(Bn)), 4 * TM * TN, (4 * TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            // stores after the last MFMAs: gives the global loads a whole stage of latency cover (see conv_fwd_kernel)
            PNP_LAST_SLICE(f1.mma(acc), (la.store(An), lb.store(Bn)), 4 * TM * TN)
            __syncthreads();
            f0.load(An, Bn, 0, wm0, wn0, lane);
        }
    }

    wgrad_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, mm0, n0, wm0, wn0, lane);
}

// ---- filter gradient, linear pixel walk (any stride, zero padding, OW >= 32, C % 4 == 0, K % 4 == 0), with a register
// ring of DEPTH global-load stages.  Unlike the forward kernel (whose A rows are re-read for 9 taps and whose filters sit in L2), every
// stage of the filter gradient fetches pixels nobody has touched before — all (tap, channel, filter) tiles of a pixel range walk it in
// step, so the first toucher misses to HBM and the others wait on the same miss: the load latency of EVERY stage is an HBM round trip,
// longer than the one stage (~64 MFMAs per wave) of cover conv_wgrad_kernel gives it.  Here stage j + DEPTH is requested while stage j
// is contracted; the ring is indexed with compile-time constants only (loop unrolled by DEPTH), the tail is peeled.
// PNP_WGRAD_ABLATE (compile-time, timing experiments only — results are wrong when set): 1 = no global loads after the prologue,
// 2 = x loads at a fixed address (no address arithmetic), 4 = no LDS stores, 8 = no barrier
#ifndef PNP_WGRAD_ABLATE
#define PNP_WGRAD_ABLATE 0
#endif
// UNI (opt-in, see launch_wgrad_tile): C % BM == 0, so a 128-row tile of the filter gradient holds ONE filter tap and the coordinates /
// validity / offset of a loader row depend on the pixel only — they are wave-uniform and are carried in SGPRs by the scalar unit (a wave's
// two half-waves load two consecutive pixels: two scalar rows per load, selected per lane).  The per-lane walk costs 1.45 VALU
// instructions per MFMA (PMC), this form about 0.15.
#ifndef PNP_WGRAD_UNIFORM_ROWS
#define PNP_WGRAD_UNIFORM_ROWS 1      // round 3: measured +2 % (512->512: 0.613 -> 0.601 ms, g10 3.07 -> 2.98 ms); PNP_WGRAD_UNI=0 at run time: off
#endif
// (UNI is folded into the depth parameter — DEPTH_ = 10 + depth — so that the measured kernels keep their symbol names.)
#ifndef PNP_WGRAD_NARROW_WAVES
#define PNP_WGRAD_NARROW_WAVES 2
#endif
template <int BM, int BN, int WM, int WN, int DEPTH_>
__global__ void __launch_bounds__(NTHREADS, (BN <= 64 ? PNP_WGRAD_NARROW_WAVES : 2)) conv_wgrad_ring_kernel(ConvArgs a) {
    constexpr bool UNI = DEPTH_ >= 10;
    constexpr int DEPTH = UNI ? DEPTH_ - 10 : DEPTH_;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    constexpr int AC4 = BM / 4, ARP = NTHREADS / AC4, ANP = BK / ARP;
    constexpr int BC4 = BN / 4, BRP = NTHREADS / BC4, BNP = BK / BRP;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

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

    const int nchunks_total = (a.M + BK - 1) / BK;
    const int c_begin = z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;
    const int nchunks = c_end - c_begin;

    // A rows: output pixel p = stage*32 + arow + ARP*i, this thread's 4 consecutive m' (one tap, 4 channels), carried incrementally
    const int acol = t % AC4, arow = t / AC4;
    const int mm = mm0 + 4 * acol;
    const bool mok = mm < a.Kred;
    const int m_ = mok ? mm : 0;
    const int rs_u = m_ / a.C, c_u = m_ - rs_u * a.C;
    const int r_u = rs_u / a.S, s_u = rs_u - r_u * a.S;
    // input coordinates of (output pixel, this thread's tap): ih = oh*stride + l_dh, iw = ow*stride + l_dw; the row / image wraps of the
    // OUTPUT walk are the compares ih >= ih_lim / iw >= iw_lim
    const int l_dh = r_u * a.dil - a.pad_t, l_dw = s_u * a.dil - a.pad_l;
    const int iw_lim = a.OW * a.stride + l_dw, ih_lim = a.OH * a.stride + l_dh;
    // one stage = BK consecutive output pixels = adv_h whole rows + adv_w pixels (rows of >= BK pixels: 0 rows + BK pixels; the 16^2 / 8^2
    // maps of the critics' last blocks: 2 / 4 rows + 0 pixels); at most one row wrap and one image wrap per step either way (host-checked)
    const int adv_h = BK / a.OW, adv_w = BK - adv_h * a.OW;
    const int step_w = adv_w * a.stride, step_h = adv_h * a.stride, step_off = (step_h * a.W + step_w) * a.C * 4;
    const int wrap_w = a.OW * a.stride, wrap_w_off = a.stride * (a.W - a.OW) * a.C * 4;
    const int wrap_h = a.OH * a.stride, wrap_h_off = (a.H - a.OH * a.stride) * a.W * a.C * 4;
    int l_iw[ANP], l_ih[ANP], l_off[ANP];
#pragma unroll
    for (int i = 0; i < ANP; ++i) {
        const int p = c_begin * BK + arow + ARP * i;
        const int n = p / a.OHW;
        const int rem = p - n * a.OHW;
        const int oh = rem / a.OW, ow = rem - oh * a.OW;
        l_ih[i] = oh * a.stride + l_dh;
        l_iw[i] = ow * a.stride + l_dw;
        l_off[i] = (((n * a.H + l_ih[i]) * a.W + l_iw[i]) * a.C + c_u) * 4;
    }
    // UNI: the same walk per (pass i, half-wave h) in wave-uniform registers; the tile's tap and first channel are uniform too
    static_assert(!UNI || (AC4 == 32 && ARP == 8), "UNI: a half-wave = one pixel row of 128 channels");
    const int t_rs = mm0 / a.C, t_c0 = mm0 - t_rs * a.C;
    const int t_r = t_rs / a.S, t_s = t_rs - t_r * a.S;
    const int u_dh = t_r * a.dil - a.pad_t, u_dw = t_s * a.dil - a.pad_l;
    const int u_iw_lim = a.OW * a.stride + u_dw, u_ih_lim = a.OH * a.stride + u_dh;
    const unsigned lane_c = (unsigned)((t_c0 + 4 * acol) * 4);
    int u_iw[ANP][2], u_ih[ANP][2], u_off[ANP][2];
    if constexpr (UNI) {
#pragma unroll
        for (int i = 0; i < ANP; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int p = c_begin * BK + 2 * wave + h + ARP * i;
                const int n = p / a.OHW;
                const int rem = p - n * a.OHW;
                const int oh = rem / a.OW, ow = rem - oh * a.OW;
                u_ih[i][h] = oh * a.stride + u_dh;
                u_iw[i][h] = ow * a.stride + u_dw;
                u_off[i][h] = (((n * a.H + u_ih[i][h]) * a.W + u_iw[i][h]) * a.C) * 4;
            }
    }
    // B rows (dy): constant per-thread offset inside a stage, the stage is the scalar offset
    const int bcol = t % BC4, brow = t / BC4;
    unsigned boff[BNP];
#pragma unroll
    for (int i = 0; i < BNP; ++i) boff[i] = (n0 + 4 * bcol < a.K) ? (unsigned)(((brow + BRP * i) * a.K + n0 + 4 * bcol) * 4) : OOB2;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);

    // Fetches past this split's last stage (the ring runs DEPTH ahead) read the next split's rows, or zeros past the end of the tensors;
    // those stages are never contracted.  Rows past the LAST pixel are zeros from the hardware (same argument as conv_wgrad_kernel MODE 3).
    struct Stage {
        f32x4 a[ANP];
        f32x4 b[BNP];
    };
    Stage ring[DEPTH];
    int l_chunk = c_begin;
    auto gload = [&](Stage& st) {
#pragma unroll
        for (int i = 0; i < ANP; ++i) {
            if constexpr (UNI) {
                unsigned so[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const bool ok = ((unsigned)u_ih[i][h] < (unsigned)a.H) & ((unsigned)u_iw[i][h] < (unsigned)a.W);
                    so[h] = ok ? (unsigned)u_off[i][h] : OOB2;          // + lane_c (< 2 KiB) stays out of range: x < 2 GiB on this path
                    u_iw[i][h] += step_w;
                    u_ih[i][h] += step_h;
                    u_off[i][h] += step_off;
                    const bool ww = u_iw[i][h] >= u_iw_lim;
                    u_iw[i][h] -= ww ? wrap_w : 0;
                    u_ih[i][h] += ww ? a.stride : 0;
                    u_off[i][h] += ww ? wrap_w_off : 0;
                    const bool hw = u_ih[i][h] >= u_ih_lim;
                    u_ih[i][h] -= hw ? wrap_h : 0;
                    u_off[i][h] += hw ? wrap_h_off : 0;
                }
                st.a[i] = bload4(rx, ((lane & 32) ? so[1] : so[0]) + lane_c);
                continue;
            }
            if constexpr (PNP_WGRAD_ABLATE & 2) {          // timing experiment: the loads without their address arithmetic
                st.a[i] = bload4(rx, (unsigned)l_off[i] & 0xFFFFFu);
                continue;
            }
            const bool ok = mok & ((unsigned)l_ih[i] < (unsigned)a.H) & ((unsigned)l_iw[i] < (unsigned)a.W);
            st.a[i] = bload4(rx, ok ? (unsigned)l_off[i] : OOB);
            l_iw[i] += step_w;
            l_ih[i] += step_h;
            l_off[i] += step_off;
            const bool ww = l_iw[i] >= iw_lim;                  // at most one wrap per step (adv_w < OW)
            l_iw[i] -= ww ? wrap_w : 0;
            l_ih[i] += ww ? a.stride : 0;
            l_off[i] += ww ? wrap_w_off : 0;
            const bool hw = l_ih[i] >= ih_lim;
            l_ih[i] -= hw ? wrap_h : 0;
            l_off[i] += hw ? wrap_h_off : 0;
        }
        // (readfirstlane: the stage counter is uniform, but the compiler did not prove it and wrapped every one of these loads in a
        // waterfall loop — 16 extra basic blocks per two stages, which also fenced the MFMA / load interleave)
        const int soff = __builtin_amdgcn_readfirstlane(l_chunk * BK * a.K * 4);
#pragma unroll
        for (int i = 0; i < BNP; ++i) st.b[i] = bload4s(rw, boff[i], soff);
        l_chunk = (l_chunk + 1 < nchunks_total) ? l_chunk + 1 : nchunks_total;      // rows past the end of dy read as zeros
    };
    auto lstore = [&](const Stage& st, float* An, float* Bn) {
#pragma unroll
        for (int i = 0; i < ANP; ++i) *reinterpret_cast<f32x4*>(An + (arow + ARP * i) * LDA + 4 * acol) = st.a[i];
#pragma unroll
        for (int i = 0; i < BNP; ++i) *reinterpret_cast<f32x4*>(Bn + (brow + BRP * i) * LDB + 4 * bcol) = st.b[i];
    };

    Acc<TM, TN> acc;
    acc.zero();
    if (nchunks > 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) gload(ring[d]);
        lstore(ring[0], lds, lds + 2 * ASZ);
        __syncthreads();
        Frag<TM, TN, false, LDA, LDB> f0, f1;
        f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
        constexpr int NDS = (4 * TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN);
        auto stage = [&](int j, Stage& fetch_into, const Stage& store_from, bool fetch) {
            const int cur = j & 1;
            const float* As = lds + cur * ASZ;
            const float* Bs = lds + 2 * ASZ + cur * BSZ;
            float* An = lds + (cur ^ 1) * ASZ;
            float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
            f1.load(As, Bs, 1, wm0, wn0, lane);
            if constexpr (!(PNP_WGRAD_ABLATE & 1)) {
                if (fetch) gload(fetch_into);
            }
            f0.mma(acc);
            __builtin_amdgcn_sched_group_barrier(0x100, 4 * TM + 4 * TN, 0);      // slice-1 fragment reads first
#pragma unroll
            for (int i = 0; i < 4 * TM * TN; ++i) {                                // then address arithmetic + loads under the MFMAs
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 12, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            PNP_SCHED_FENCE();
            PNP_SLICE(f0.load(As, Bs, 2, wm0, wn0, lane), f1.mma(acc), 4 * TM * TN, NDS)
            PNP_SLICE2(f1.load(As, Bs, 3, wm0, wn0, lane), f0.mma(acc), lstore(store_from, An, Bn), 4 * TM * TN, NDS)
            if constexpr (PNP_WGRAD_ABLATE & 4) {
                f1.mma(acc);
            } else {
                PNP_LAST_SLICE(f1.mma(acc), lstore(store_from, An, Bn), 4 * TM * TN)
            }
            if constexpr (!(PNP_WGRAD_ABLATE & 8)) __syncthreads();
            f0.load(An, Bn, 0, wm0, wn0, lane);
        };
        const int nmain = (nchunks / DEPTH) * DEPTH;
        for (int j0 = 0; j0 < nmain; j0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) stage(j0 + d, ring[d], ring[(d + 1) % DEPTH], true);
        }
        static_assert(DEPTH >= 1 && DEPTH <= 3, "tail below is written for rings of up to three");
        if (DEPTH >= 2 && nchunks - nmain >= 1) stage(nmain, ring[0], ring[1 % DEPTH], false);
        if (DEPTH >= 3 && nchunks - nmain >= 2) stage(nmain + 1, ring[1 % DEPTH], ring[2 % DEPTH], false);
    }

    wgrad_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, mm0, n0, wm0, wn0, lane);
}

__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int nsplit,
                                     size_t stride, int accumulate) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += gs) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * stride + i];
        out[i] = accumulate ? out[i] + s : s;
    }
}
// The same sum for MANY partials of a SMALL gradient (cls1's 64->64 @256^2 filter gradient at N = 32: 300+ partials of 36 864 values —
// one thread per value walking all of them serially ran at 0.3 TB/s, 247 us; r06 trace), in two deterministic levels and 16 bytes per lane:
// level 1 (blockIdx.y = group g): the partials [g zg, (g + 1) zg) summed in order into the group's FIRST partial (in place: that row is read
// by this thread only); level 2: the group rows summed in order into out.  n % 4 == 0, 16-byte aligned.
__global__ void __launch_bounds__(256) splitk_reduce_group_kernel(float* __restrict__ part, size_t n4, int nsplit, size_t stride, int zg) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int z0 = blockIdx.y * zg, z1 = min(z0 + zg, nsplit);
    f32x4 s = *reinterpret_cast<const f32x4*>(part + (size_t)z0 * stride + i * 4);
    for (int z = z0 + 1; z < z1; ++z) s += *reinterpret_cast<const f32x4*>(part + (size_t)z * stride + i * 4);
    *reinterpret_cast<f32x4*>(part + (size_t)z0 * stride + i * 4) = s;
}
static int launch_splitk_reduce(float* ws, float* out, size_t n, int nsplit, int accumulate, hipStream_t st) {
    // OFF by default: the filter gradients run on the side stream next to the data gradients, so the 0.2 ms per step this saves in kernel
    // time does not show in the step (within-run A/B: 269.65 vs 270.65 slices/s), and it changes the summation ORDER of every split
    // direct filter gradient — one more realisation of the chaotic fp32-vs-fp32 comparison of tests/test_gpu_adversarial.py for nothing
    static const int two_level = getenv("PNP_SPLITK_TWO_LEVEL") ? atoi(getenv("PNP_SPLITK_TWO_LEVEL")) : 0;
    int nz = nsplit;
    size_t stride = n;
    if (two_level && nsplit >= 32 && (n % 4) == 0 && ((uintptr_t)ws % 16) == 0 && n / 4 * (size_t)nsplit >= (size_t)1 << 18) {
        const int zg = 16, ng = pnp_cdiv(nsplit, zg);
        hipLaunchKernelGGL(splitk_reduce_group_kernel, dim3((unsigned)pnp_cdiv((long long)(n / 4), 256), (unsigned)ng), dim3(256), 0, st, ws, n / 4, nsplit, n, zg);
        PNP_CHECK_LAUNCH("splitk_reduce_group_kernel");
        nz = ng;
        stride = n * zg;
    }
    int nb = pnp_cdiv((long long)n, 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, (const float*)ws, out, n, nz, stride, accumulate);
    PNP_CHECK_LAUNCH("splitk_reduce_kernel");
    return PNP_OK;
}

// ===================================== direct forward conv for narrow outputs ============================================
// K <= 16 output channels at stride 1 (the 40->5 logits conv and g1's 16->16 convs at 256^2, and their data gradients when the INPUT
// is that narrow): an MFMA tile is half to 5/6 empty in N (40->5 measured 15 TF/s, bound by the matrix pipe doing mostly zeros).  A thread owns one output pixel and its K accumulators.  The workgroup's input patch
// ((8 + (R-1)dil) x (32 + (S-1)dil) pixels x C channels) is staged once in LDS with a pixel stride of C+4 floats when C/4 is even (an odd
// number of 16-byte slots per pixel: the 64 lanes of a ds_read_b128 spread over all banks); the filter values are uniform over the
// wave and come through the scalar cache (constant address space), so the inner loop is v_pk_fma_f32 acc, x, s[w].
struct NarrowArgs {
    const float* x;
    const float* w;
    float* y;
    int N, H, W, C, K, R, S, OH, OW, dil, pad_t, pad_l;
    int PH, PW, CP;                 // patch extent (pixels) and padded pixel stride (floats)
    int do_drop;
    float drop_keep;
    uint32_t drop_thresh, drop_key;
    unsigned x_bytes;
    const pnp_step_params* sp;
    uint32_t drop_sid;
};

template <int KK, bool EXACT>
__global__ void __launch_bounds__(256) conv_fwd_narrow_kernel(NarrowArgs a) {
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int t = threadIdx.x;
    const int tx = t & 31, ty = t >> 5;
    const int ow0 = blockIdx.x * 32, oh0 = blockIdx.y * 8, n = blockIdx.z;
    const int K = EXACT ? KK : a.K;
    const int C4 = a.C >> 2;
    // ---- stage the patch (zero outside the image) --------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const int nvec = a.PH * a.PW * C4;
    for (int i = t; i < nvec; i += 256) {
        const int pix = i / C4, c4 = i - pix * C4;
        const int pr = pix / a.PW, pc = pix - pr * a.PW;
        const int ih = oh0 - a.pad_t + pr, iw = ow0 - a.pad_l + pc;
        const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const f32x4 v = bload4(rx, ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C + 4 * c4) * 4) : OOB);
        *reinterpret_cast<f32x4*>(xs + pix * a.CP + 4 * c4) = v;
    }
    __syncthreads();
    // ---- accumulate ---------------------------------------------------------------------------------------------------
    pnp_cfloat* wc = (pnp_cfloat*)(uintptr_t)a.w;
    constexpr int UNR = KK <= 8 ? 2 : 1;        // 4*KK filter values per channel quad live in SGPRs
    float acc[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) acc[k] = 0.f;
    for (int r = 0; r < a.R; ++r)
        for (int sx = 0; sx < a.S; ++sx) {
            const float* xp = xs + ((ty + r * a.dil) * a.PW + tx + sx * a.dil) * a.CP;
            const int wbase = (r * a.S + sx) * a.C * K;
#pragma unroll UNR
            for (int c4 = 0; c4 < C4; ++c4) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xp + 4 * c4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int k = 0; k < KK; ++k)
                        if (k < K) acc[k] = fmaf(xv[e], wc[wbase + (4 * c4 + e) * K + k], acc[k]);
            }
        }
    const int oh = oh0 + ty, ow = ow0 + tx;
    if (oh < a.OH && ow < a.OW) {
        const size_t m = ((size_t)n * a.OH + oh) * a.OW + ow;
#pragma unroll
        for (int k = 0; k < KK; ++k)
            if (k < K) {
                const size_t idx = m * K + k;
                float v = acc[k];
                if (a.do_drop) v = pnp_drop_keep((uint32_t)idx, pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid), a.drop_thresh) ? v / a.drop_keep : 0.f;
                a.y[idx] = v;
            }
    }
}

// ===================================== direct filter gradient for narrow outputs ============================================
// K <= 16 output channels (g1's 16-channel convs, the 40->5 logits conv, the mask critic's first conv): an MFMA tile is at most half
// full in N and the reduction (all pixels) is long, so this one runs on the vector ALUs.  A thread owns PPT (tap, channel) pairs x K
// accumulators, walks the pixels of its workgroup's slab, reads ONE x value per pair and pixel plus the pixel's K dy values (the
// same address for a whole pixel group: a broadcast load) and issues PPT*K FMAs.  Workgroups with few pairs split their slab over G
// pixel groups and combine through LDS.  Partials [nblk][R*S*C*K] are summed by splitk_reduce_kernel (fixed order: deterministic).
struct WgdArgs {
    const float* x;
    const float* dy;
    float* part;
    int N, H, W, C, K, R, S, OH, OW, stride, dil, pad_t, pad_l;
    int P;             // N*OH*OW
    int npairs;        // R*S*C
    int G;             // pixel groups per workgroup (1 when a thread owns several pairs)
    int ppb;           // pixels per workgroup
    unsigned x_bytes, dy_bytes;
};


// UNI (one pixel group): the pixel index is uniform over the workgroup, so the K dy values come through the SCALAR cache into SGPRs
// (v_fmac with an SGPR operand) instead of 64 lanes x 16 B of vector-memory traffic per wave and pixel for 64 useful bytes — the
// texture-address unit, not the FMAs, bounded the first version (16->16: 0.54 ms).
template <int KK, int PPT, bool EXACT, bool UNI>
__global__ void __launch_bounds__(256) wgrad_direct_kernel(WgdArgs a) {
    extern __shared__ float red[];      // [G][npairs*K], only when G > 1
    const int t = threadIdx.x;
    if (UNI && PPT == 1 && (t & ~63) >= a.npairs) return;      // whole wave without a pair (no barrier on this path)
    const int g = (PPT == 1 && !UNI) ? t / a.npairs : 0;
    const int j0 = (PPT == 1) ? t - g * a.npairs : t;
    const bool active = g < a.G;
    const int K = EXACT ? KK : a.K;
    int coff[PPT], dh[PPT], dw[PPT];
    bool pv[PPT];
#pragma unroll
    for (int q = 0; q < PPT; ++q) {
        const int j = j0 + q * 256;
        pv[q] = active && j < a.npairs;
        const int jj = pv[q] ? j : 0;
        const int tap = jj / a.C;
        coff[q] = jj - tap * a.C;
        const int r = tap / a.S;
        dh[q] = r * a.dil - a.pad_t;
        dw[q] = (tap - r * a.S) * a.dil - a.pad_l;
    }
    float acc[PPT][KK];
#pragma unroll
    for (int q = 0; q < PPT; ++q)
#pragma unroll
        for (int k = 0; k < KK; ++k) acc[q][k] = 0.f;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(a.dy, a.dy_bytes);
    const int p0 = blockIdx.x * a.ppb;
    const int p1 = (p0 + a.ppb < a.P) ? p0 + a.ppb : a.P;
    int p = UNI ? p0 : p0 + g;
    pnp_cfloat* dyc = (pnp_cfloat*)(uintptr_t)a.dy;
    const int OHW = a.OH * a.OW;
    int n = p / OHW;
    int rem = p - n * OHW;
    int oh = rem / a.OW;
    int ow = rem - oh * a.OW;
    // U pixels per trip: all their loads are issued before the first FMA (one pixel per trip is bound by the global-load latency:
    // 16->16 ran at 0.58 ms that way)
    constexpr int U = 4;
    for (; p < p1; p += U * a.G) {
        float dv[U][KK], xv[U][PPT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pu = p + u * a.G;
            const bool pin = pu < p1;
            if constexpr (UNI) {
                const size_t base = (size_t)(pin ? pu : p) * K;      // uniform; a pixel past the slab re-reads a valid row (its x is 0)
#pragma unroll
                for (int k = 0; k < KK; ++k) dv[u][k] = (k < K) ? dyc[base + k] : 0.f;
            } else if constexpr (EXACT && (KK % 4) == 0) {
#pragma unroll
                for (int k4 = 0; k4 < KK / 4; ++k4) {
                    const f32x4 v = bload4(rd, pin ? (unsigned)((pu * KK + 4 * k4) * 4) : OOB);
                    dv[u][4 * k4] = v[0]; dv[u][4 * k4 + 1] = v[1]; dv[u][4 * k4 + 2] = v[2]; dv[u][4 * k4 + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int k = 0; k < KK; ++k) dv[u][k] = (pin && k < K) ? bload1(rd, (unsigned)((pu * K + k) * 4)) : 0.f;
            }
            const int vh = oh * a.stride, vw = ow * a.stride;
#pragma unroll
            for (int q = 0; q < PPT; ++q) {
                const int ih = vh + dh[q], iw = vw + dw[q];
                const bool ok = pin & pv[q] & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                xv[u][q] = bload1(rx, ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C + coff[q]) * 4) : OOB);
            }
            ow += a.G;
            while (ow >= a.OW) {
                ow -= a.OW;
                if (++oh == a.OH) { oh = 0; ++n; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < PPT; ++q)
#pragma unroll
                for (int k = 0; k < KK; ++k) acc[q][k] = fmaf(xv[u][q], dv[u][k], acc[q][k]);
    }
    const int nout = a.npairs * K;
    float* outp = a.part + (size_t)blockIdx.x * nout;
    if (a.G > 1) {          // PPT == 1
        if (active) {
#pragma unroll
            for (int k = 0; k < KK; ++k)
                if (k < K) red[g * nout + j0 * K + k] = acc[0][k];
        }
        __syncthreads();
        for (int e = t; e < nout; e += 256) {
            float sum = 0.f;
            for (int gg = 0; gg < a.G; ++gg) sum += red[gg * nout + e];
            outp[e] = sum;
        }
    } else {
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (pv[q]) {
                const int j = j0 + q * 256;
#pragma unroll
                for (int k = 0; k < KK; ++k)
                    if (k < K) outp[j * K + k] = acc[q][k];
            }
        }
    }
}

// Same for C % 4 == 0 and many pairs (the 40->5 logits conv: 1000 pairs): a thread owns QPT quads of 4 consecutive channels of one
// tap and fetches each with ONE 16-byte load — with one dword per pair the 4 loads per lane and pixel kept the texture-address unit
// busy for the whole 0.41 ms of the kernel.  One pixel group (uniform pixel): dy through the scalar cache.
template <int KK, int QPT, bool EXACT>
__global__ void __launch_bounds__(256) wgrad_direct4_kernel(WgdArgs a) {
    const int t = threadIdx.x;
    const int nquads = a.npairs >> 2;
    const int K = EXACT ? KK : a.K;
    int coff[QPT], dh[QPT], dw[QPT];
    bool pv[QPT];
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        const int i = t + q * 256;
        pv[q] = i < nquads;
        const int j = pv[q] ? 4 * i : 0;
        const int tap = j / a.C;
        coff[q] = j - tap * a.C;
        const int r = tap / a.S;
        dh[q] = r * a.dil - a.pad_t;
        dw[q] = (tap - r * a.S) * a.dil - a.pad_l;
    }
    float acc[QPT][4][KK];
#pragma unroll
    for (int q = 0; q < QPT; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int k = 0; k < KK; ++k) acc[q][e][k] = 0.f;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    pnp_cfloat* dyc = (pnp_cfloat*)(uintptr_t)a.dy;
    const int p0 = blockIdx.x * a.ppb;
    const int p1 = (p0 + a.ppb < a.P) ? p0 + a.ppb : a.P;
    const int OHW = a.OH * a.OW;
    int n = p0 / OHW;
    int rem = p0 - n * OHW;
    int oh = rem / a.OW;
    int ow = rem - oh * a.OW;
    constexpr int U = 4;
    for (int p = p0; p < p1; p += U) {
        float dv[U][KK];
        f32x4 xv[U][QPT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int pu = p + u;
            const bool pin = pu < p1;
            const size_t base = (size_t)(pin ? pu : p) * K;
#pragma unroll
            for (int k = 0; k < KK; ++k) dv[u][k] = (k < K) ? dyc[base + k] : 0.f;
            const int vh = oh * a.stride, vw = ow * a.stride;
#pragma unroll
            for (int q = 0; q < QPT; ++q) {
                const int ih = vh + dh[q], iw = vw + dw[q];
                const bool ok = pin & pv[q] & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                xv[u][q] = bload4(rx, ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C + coff[q]) * 4) : OOB);
            }
            if (++ow == a.OW) {
                ow = 0;
                if (++oh == a.OH) { oh = 0; ++n; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < QPT; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int k = 0; k < KK; ++k) acc[q][e][k] = fmaf(xv[u][q][e], dv[u][k], acc[q][e][k]);
    }
    float* outp = a.part + (size_t)blockIdx.x * a.npairs * K;
#pragma unroll
    for (int q = 0; q < QPT; ++q) {
        if (pv[q]) {
            const int j = 4 * (t + q * 256);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < KK; ++k)
                    if (k < K) outp[(j + e) * K + k] = acc[q][e][k];
        }
    }
}

// forward conv with a split reduction: sum the partials, then the dropout of the fused conv->dropout (same element-index hash as the
// un-split epilogue)
__global__ void splitk_reduce_drop_kernel(const float* __restrict__ part, float* __restrict__ out, size_t n, int nsplit, size_t stride,
                                          uint32_t drop_key, uint32_t drop_thresh, float drop_keep, const pnp_step_params* sp,
                                          uint32_t drop_sid) {
    drop_key = pnp_eff_drop_key(drop_key, sp, drop_sid);
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += gs) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * stride + i];
        out[i] = pnp_drop_keep((uint32_t)i, drop_key, drop_thresh) ? s / drop_keep : 0.f;
    }
}

// inference-mode batch norm as one multiply-add per channel: scale = gamma * rsqrt(var + eps), shift = beta - mean * scale
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ mean,
                               const float* __restrict__ var, float* __restrict__ scale, float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] * (1.0f / sqrtf(var[c] + eps));
    scale[c] = sc;
    shift[c] = beta[c] - mean[c] * sc;
}

// many partials (one per workgroup of wgrad_direct_kernel), few outputs: 64 outputs x 16 slices of the partial list per workgroup
__global__ void __launch_bounds__(1024) splitk_reduce_many_kernel(const float* __restrict__ part, float* __restrict__ out, int n,
                                                                  int nsplit, int accumulate) {
    __shared__ float red[16][64];
    const int l = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + l;
    float s = 0.f;
    if (e < n)
        for (int z = sl; z < nsplit; z += 16) s += part[(size_t)z * n + e];
    red[sl][l] = s;
    __syncthreads();
    if (sl == 0 && e < n) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += red[j][l];
        out[e] = accumulate ? out[e] + tot : tot;
    }
}

// split partials of one stride-phase ([nsplit][M][K] row-major) -> summed and scattered to the phase's pixels of dx
__global__ void splitk_reduce_scatter_kernel(const float* __restrict__ part, ConvArgs a, int nsplit, size_t stride) {
    const size_t n = (size_t)a.M * a.K;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += gs) {
        float s = 0.f;
        for (int z = 0; z < nsplit; ++z) s += part[(size_t)z * stride + i];
        const int m = (int)(i / a.K);
        a.y[out_row(a, m, true) + (i - (size_t)m * a.K)] = s;
    }
}

// every stride-phase sub-filter of w [R][S][C][K] in one launch: tap (r, s) belongs to phase (pa, pb) = (r % st, s % st) and lands,
// flipped on both axes, in that phase's wt [T][U][K][C] block; the blocks are packed in (pa, pb) order (plan_phases' wt_off)
__global__ void flip_transpose_phase_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int S, int C, int K, int st) {
    __shared__ float tile[32][33];
    const int r = blockIdx.z / S, sx = blockIdx.z - r * S;
    const int pa = r % st, pb = sx % st;
    const int T = (R - pa + st - 1) / st, U = (S - pb + st - 1) / st;
    int taps_before = 0;                                             // taps of the phases packed in front of (pa, pb)
    for (int a = 0; a < st; ++a)
        for (int b = 0; b < st; ++b)
            if (a * st + b < pa * st + pb) taps_before += ((R - a + st - 1) / st) * ((S - b + st - 1) / st);
    const int tf = T - 1 - r / st, uf = U - 1 - sx / st;            // flipped position inside the phase filter
    const float* src = w + (size_t)blockIdx.z * C * K;
    float* dst = wt + ((size_t)taps_before + tf * U + uf) * K * C;
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, k = k0 + tx;
        tile[i][tx] = (c < C && k < K) ? src[(size_t)c * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int k = k0 + i, c = c0 + tx;
        if (k < K && c < C) dst[(size_t)k * C + c] = tile[tx][i];
    }
}

// w [R][S][C][K] -> wt [R][S][K][C] with both spatial axes flipped (filters of the dgrad convolution)
__global__ void flip_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int R, int S, int C, int K) {
    __shared__ float tile[32][33];
    const int rs = blockIdx.z;
    const int r = rs / S, s = rs - r * S;
    const int rs_f = (R - 1 - r) * S + (S - 1 - s);
    const float* src = w + (size_t)rs * C * K;
    float* dst = wt + (size_t)rs_f * K * C;
    const int c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int i = ty; i < 32; i += 8) {
        int c = c0 + i, k = k0 + tx;
        tile[i][tx] = (c < C && k < K) ? src[(size_t)c * K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        int k = k0 + i, c = c0 + tx;
        if (k < K && c < C) dst[(size_t)k * C + c] = tile[tx][i];
    }
}

__global__ void naive_conv_kernel(ConvArgs a) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)a.M * a.K;
    if (idx >= total) return;
    const int k = (int)(idx % a.K);
    const int m = (int)(idx / a.K);
    const int n = m / a.OHW;
    const int rem = m - n * a.OHW;
    const int oh = rem / a.OW, ow = rem - oh * a.OW;
    float acc = 0.f;
    for (int r = 0; r < a.R; ++r)
        for (int s = 0; s < a.S; ++s) {
            int ih = 0, iw = 0;
            if (!map_coord<true>(oh * a.stride - a.pad_t + r * a.dil, a.H, a.ups, a.pad_mode, ih)) continue;
            if (!map_coord<true>(ow * a.stride - a.pad_l + s * a.dil, a.W, a.ups, a.pad_mode, iw)) continue;
            const float* xp = a.x + ((size_t)(n * a.H + ih) * a.W + iw) * a.C;
            const float* wp = a.w + (size_t)((r * a.S + s) * a.C) * a.K + k;
            for (int c = 0; c < a.C; ++c) acc = fmaf(xp[c], wp[(size_t)c * a.K], acc);
        }
    a.y[idx] = acc;
}

// tf.pad(x, p, 'SYMMETRIC') in H and W (mirror including the edge): xp[N,H+2p,W+2p,C]; one float4 (or float) per thread
__global__ void sympad_fwd_kernel(const float* __restrict__ x, float* __restrict__ xp, int N, int H, int W, int C, int p, int vec) {
    const int CV = vec ? (C >> 2) : C;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const size_t total = (size_t)N * Hp * Wp * CV;
    const size_t gs = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gs) {
        const int cv = (int)(i % CV);
        size_t q = i / CV;
        const int wp = (int)(q % Wp);
        q /= Wp;
        const int hp = (int)(q % Hp);
        const int n = (int)(q / Hp);
        int h = hp - p, w = wp - p;
        h = h < 0 ? -1 - h : (h >= H ? 2 * H - 1 - h : h);
        w = w < 0 ? -1 - w : (w >= W ? 2 * W - 1 - w : w);
        const size_t src = (((size_t)n * H + h) * W + w) * C;
        if (vec) *reinterpret_cast<f32x4*>(xp + i * 4) = *reinterpret_cast<const f32x4*>(x + src + 4 * cv);
        else xp[i] = x[src + cv];
    }
}

// sympad backward: dx[n,h,w,c] = sum of dxp over every padded position that mirrors onto (h,w)
__global__ void sympad_bwd_kernel(const float* __restrict__ dxp, float* __restrict__ dx, int N, int H, int W, int C,
                                  int p) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)N * H * W * C;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    size_t q = idx / C;
    const int w = (int)(q % W);
    q /= W;
    const int h = (int)(q % H);
    const int n = (int)(q / H);
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    // padded rows mapping to h: h+p always; (p-1-h) if h<p; (2H-1-h+p) if h >= H-p
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + p;
    if (h < p) hs[nh++] = p - 1 - h;
    if (h >= H - p) hs[nh++] = 2 * H - 1 - h + p;
    ws[nw++] = w + p;
    if (w < p) ws[nw++] = p - 1 - w;
    if (w >= W - p) ws[nw++] = 2 * W - 1 - w + p;
    float s = 0.f;
    for (int i = 0; i < nh; ++i)
        for (int j = 0; j < nw; ++j) s += dxp[(((size_t)n * Hp + hs[i]) * Wp + ws[j]) * C + c];
    dx[idx] = s;
}

// ------------------------------------ host side -------------------------------------------------
int check_geom(const pnp_conv_geom* g, const char* who) {
    PNP_REQUIRE(g != nullptr, "%s: null geometry", who);
    PNP_REQUIRE(g->N > 0 && g->H > 0 && g->W > 0 && g->C > 0 && g->K > 0 && g->R > 0 && g->S > 0 && g->OH > 0 &&
                    g->OW > 0 && g->stride > 0 && g->dil > 0,
                "%s: non-positive dimension", who);
    PNP_REQUIRE(g->pad_mode == PNP_PAD_ZERO || g->pad_mode == PNP_PAD_SYMMETRIC, "%s: bad pad_mode %d", who, g->pad_mode);
    PNP_REQUIRE(g->pad_t >= 0 && g->pad_l >= 0, "%s: negative padding", who);
    PNP_REQUIRE(g->dtype == PNP_DTYPE_F32 || g->dtype == PNP_DTYPE_BF16, "%s: dtype %d (PNP_DTYPE_F32 or PNP_DTYPE_BF16)", who, g->dtype);
    if (g->pad_mode == PNP_PAD_SYMMETRIC)
        PNP_REQUIRE(g->pad_t <= g->H && g->pad_l <= g->W, "%s: symmetric pad larger than the image", who);
    // last tap of the last output must not run past the (padded) input by more than the implicit zero region in
    // symmetric mode (pad-then-VALID): (OH-1)*stride + (R-1)*dil - pad_t <= H-1+pad_t
    if (g->pad_mode == PNP_PAD_SYMMETRIC) {
        PNP_REQUIRE((g->OH - 1) * g->stride + (g->R - 1) * g->dil - g->pad_t <= g->H - 1 + g->pad_t &&
                        (g->OW - 1) * g->stride + (g->S - 1) * g->dil - g->pad_l <= g->W - 1 + g->pad_l,
                    "%s: SYMMETRIC geometry reads past the mirrored border", who);
    }
    const long long xin = (long long)g->N * g->H * g->W * g->C, yout = (long long)g->N * g->OH * g->OW * g->K;
    PNP_REQUIRE(xin < (1ll << 30) && yout < (1ll << 30) && (long long)g->R * g->S * g->C * g->K < (1ll << 30),
                "%s: tensor exceeds 2^30 elements (32-bit buffer offsets)", who);
    return PNP_OK;
}

ConvArgs make_args(const float* x, const float* w, float* y, const pnp_conv_geom* g) {
    ConvArgs a{};
    a.x = x; a.w = w; a.y = y;
    a.N = g->N; a.H = g->H; a.W = g->W; a.C = g->C; a.K = g->K; a.R = g->R; a.S = g->S;
    a.OH = g->OH; a.OW = g->OW; a.stride = g->stride; a.dil = g->dil; a.pad_t = g->pad_t; a.pad_l = g->pad_l;
    a.pad_mode = g->pad_mode;
    a.dtype = g->dtype;
    a.ups = 1;
    a.M = g->N * g->OH * g->OW;
    a.Kred = g->R * g->S * g->C;
    a.OHW = g->OH * g->OW;
    a.nsplit = 1; a.chunks_per_split = 0; a.split_stride = 0;
    const bool p2 = (a.OW & (a.OW - 1)) == 0 && (a.OHW & (a.OHW - 1)) == 0;
    a.ow_sh = p2 ? __builtin_ctz((unsigned)a.OW) : -1;
    a.ohw_sh = p2 ? __builtin_ctz((unsigned)a.OHW) : -1;
    a.do_drop = 0; a.drop_keep = 1.f; a.drop_key = 0; a.drop_thresh = 0;
    static const int env_noswz = getenv("PNP_CONV_NOSWIZZLE") ? 1 : 0;
    a.xcd_swizzle = env_noswz ? 0 : 1;
    a.x_bytes = (unsigned)((size_t)g->N * g->H * g->W * g->C * sizeof(float));
    a.w_bytes = (unsigned)((size_t)g->R * g->S * g->C * g->K * sizeof(float));
    static const int env_stagger = getenv("PNP_CONV_STAGGER") ? atoi(getenv("PNP_CONV_STAGGER")) : 0;
    a.stagger = env_stagger;
    // measured on 512->2560 @32^2, B=16 (tools/bench_conv.py): forward 121.7 TF/s plain, 126.0 with groups of 4 (2: 125.9, 8: 123.6)
    static const int env_gn = getenv("PNP_CONV_GN") ? atoi(getenv("PNP_CONV_GN")) : 4;
    a.gn = env_gn;
    return a;
}

// ---- tile and reduction-split planning for the forward / data-gradient kernel -----------------------------------------------
// Tile: the widest tile that still yields >= 384 workgroups (1.5 per CU; 2 fit), else the next narrower one.  Measured at B=16
// (tools/bench_conv.py, PNP_CONV_TILE sweep): 128->128@32^2 44 TF/s with 128x128 tiles (128 workgroups) vs 70 with 128x64;
// 256->512@32^2 107 vs 100; 128->64 (dgrad) 34 with 128x64 vs 46 with 128x32.
int choose_tile(long long M, int K) {
    static const int force = getenv("PNP_CONV_TILE") ? atoi(getenv("PNP_CONV_TILE")) : -1;   // experiments: 0/1/2 = 128x{128,64,32}
    if ((K & 3) != 0) return 3;      // scalar-B variant of the narrow tile (K = 5 logits, K = 1 critic FC, ...)
    const long long mt = pnp_cdiv(M, 128);
    int tile = 2;
    if (K > 64 && mt * pnp_cdiv(K, 128) >= 384) tile = 0;
    else if (K > 32 && (mt * pnp_cdiv(K, 64) >= 384 || K > 64)) tile = 1;
    if (force >= 0 && !(force == 0 && K <= 64) && !(force <= 1 && K <= 32)) tile = force;
    return tile;
}
// Split of the reduction (data gradients only — they have a workspace for the partials): 512 workgroup slots per dispatch round
// (2 per CU).  Fewer workgroups than slots, or a nearly empty last round (g10's dgrad: 580 workgroups = 1.13 rounds, i.e. the time of
// 2), waste the chip; shorter, more numerous workgroups fill it.  Partials are summed by splitk_reduce_kernel (deterministic).
int choose_split(long long M, int K, int Kred, int tile) {
    static const int off = getenv("PNP_CONV_NOSPLIT") ? 1 : 0;
    if (off) return 1;
    const int bn = tile == 0 ? 128 : (tile == 1 ? 64 : 32);
    const long long nblk = (long long)pnp_cdiv(M, 128) * pnp_cdiv(K, bn);
    const int nch = pnp_cdiv(Kred, BK);
    const double rounds = (double)nblk / 512.0;
    int nsplit = 1;
    // (at exactly half a round — 256 tiles — two-way splitting LOSES: g4 128->128 dgrad 0.054 ms unsplit, 0.063 split)
    if (rounds <= 0.4) nsplit = (int)(512 / nblk);
    else if (rounds > 1.0 && rounds < 3.0 && (ceil(rounds) - rounds) > 0.3) {
        // a nearly empty last round: pick the split that fills the rounds best (g10's dgrad, 580 tiles = 1.13 rounds: 2 splits ->
        // 2.27 rounds cost 3 (76 %), 7 splits -> 7.93 cost 8 (99 %): 3.43 -> 2.9 ms), mild preference for fewer partials
        double best = 0.0;
        for (int ns = 1; ns <= 8; ++ns) {
            const double r = rounds * ns;
            const double score = r / ceil(r) - 0.01 * ns;
            if (score > best + 1e-9) { best = score; nsplit = ns; }
        }
    }
    const int cap = nblk <= 64 ? 32 : 8;        // a handful of tiles (4x4 / 2x2 feature maps, M = 256 rows): split deeper
    if (nsplit > cap) nsplit = cap;
    if (nsplit > nch / 8) nsplit = nch / 8;
    return nsplit < 1 ? 1 : nsplit;
}

#ifndef PNP_TAPS3_DEFAULT
#define PNP_TAPS3_DEFAULT 1
#endif
// filter shapes the tap-unrolled kernel is instantiated for: forward 3x3 / 5x5; data gradient 3x3 and the stride-phase sub-filters
constexpr bool taps_shape(int kind, int R, int S) {
    if (R == 3 && S == 3) return true;
    if (kind == 0) return R == 5 && S == 5;
    if (kind == 1) return R >= 1 && R <= 3 && S >= 1 && S <= 3 && !(R == 3 && S == 1) && !(R == 1 && S == 3);
    return false;
}

template <int BM, int BN, int WM, int WN, int KIND>
bool launch_taps(const ConvArgs& a, dim3 grid, hipStream_t st) {
    // three-stage kernel (conv_taps3_kernel) for the narrow tiles: its channel-group loop runs in pairs for odd tap counts
    static const int env_t3 = getenv("PNP_CONV_TAPS3") ? atoi(getenv("PNP_CONV_TAPS3")) : PNP_TAPS3_DEFAULT;
    // conv_taps3_kernel contracts channel groups in PAIRS when the tap count is odd (CCU = 2): every reduction split must then hold an
    // even number of groups — nothing on the device checks it (a clamped, duplicated stage would be contracted past cc_end), so the
    // condition is stated here per split, not inferred from the planner (round-3 advisor finding)
    const int cc_per = a.chunks_per_split / (a.R * a.S);
    bool pairs_ok = (a.R * a.S) % 2 == 0;
    if (!pairs_ok && cc_per > 0) {
        pairs_ok = true;
        const int ncc = a.C / BK;
        for (int z = 0; z < a.nsplit; ++z) {
            const int n = (ncc - z * cc_per) < cc_per ? (ncc - z * cc_per) : cc_per;
            if (n <= 0 || (n & 1)) pairs_ok = false;
        }
    }
    const bool t3 = env_t3 && BN <= 64 && pairs_ok;
#define PNP_TAPS(RR, SS)                                                                                                   \
    if (a.R == RR && a.S == SS) {                                                                                          \
        if constexpr (BN <= 64 && RR * SS <= 9) {     /* 5x5: 50 unrolled stages per trip and spills — stays on the two-stage kernel */ \
            if (t3) {                                                                                                      \
                PnpProfScope ps(prof_class(KIND), st, conv_flops(a), conv_bytes(a), "conv_taps3_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, \
                                BN, WM, WN, KIND, RR, SS);                                                                  \
                hipLaunchKernelGGL((conv_taps3_kernel<BM, BN, WM, WN, KIND, RR, SS>), grid, dim3(NTHREADS), 0, st, a);      \
                return true;                                                                                               \
            }                                                                                                              \
        }                                                                                                                  \
        PnpProfScope ps(prof_class(KIND), st, conv_flops(a), conv_bytes(a), "conv_taps_kernel<%d, %d, %d, %d, %d, %d, %d>", BM, BN, \
                        WM, WN, KIND, RR, SS);                                                                              \
        hipLaunchKernelGGL((conv_taps_kernel<BM, BN, WM, WN, KIND, RR, SS>), grid, dim3(NTHREADS), 0, st, a);               \
        return true;                                                                                                       \
    }
    PNP_TAPS(3, 3)
    if constexpr (KIND == 0) { PNP_TAPS(5, 5) }
    if constexpr (KIND == 1) { PNP_TAPS(1, 1) PNP_TAPS(1, 2) PNP_TAPS(2, 1) PNP_TAPS(2, 2) PNP_TAPS(2, 3) PNP_TAPS(3, 2) }
#undef PNP_TAPS
    return false;
}

template <int BM, int BN, int WM, int WN, int KIND, bool VECB>
int launch_fwd_tile(ConvArgs& a, float* split_ws, int nsplit, hipStream_t st) {
    a.nblk_m = pnp_cdiv(a.M, BM);
    a.nblk_n = pnp_cdiv(a.K, BN);
    const int nch = pnp_cdiv(a.Kred, BK);
    if (!split_ws) nsplit = 1;
    // tap-unrolled fast path: instantiated filter shape, zero padding, C % 32 == 0, K % 4 == 0, both tensors < 2 GiB
    static const int env_notaps = getenv("PNP_CONV_NOTAPS") ? 1 : 0;
    const bool taps = !env_notaps && VECB && KIND != 2 && a.pad_mode == PNP_PAD_ZERO && taps_shape(KIND, a.R, a.S) &&
                      (a.C % 32) == 0 && a.x_bytes < 0x80000000u && a.w_bytes < 0x80000000u;
    if (taps) {
        const int ncc = a.C / BK;                                    // split in whole channel groups (R*S stages each)
        const int cc_per = pnp_cdiv(ncc, nsplit);
        nsplit = pnp_cdiv(ncc, cc_per);
        a.chunks_per_split = cc_per * a.R * a.S;
    } else {
        a.chunks_per_split = pnp_cdiv(nch, nsplit);
        nsplit = pnp_cdiv(nch, a.chunks_per_split);
    }
    a.nsplit = nsplit;
    a.split_stride = (long long)a.M * a.K;
    float* final_out = a.y;
    const int drop_in_reduce = (nsplit > 1) ? a.do_drop : 0;
    if (nsplit > 1) { a.y = split_ws; a.do_drop = 0; }
    dim3 grid((unsigned)(a.nblk_m * a.nblk_n * nsplit));
    const int mode = (a.C % 32 == 0) ? 0 : ((a.C % 4 == 0) ? 1 : 2);
    if constexpr (VECB && KIND != 2) {
        if (taps) {
            // bf16 MFMA operands (configs[4]): the same tiles, splits and epilogue, operands rounded while they are staged (conv_bf16.hip)
            const bool launched = (a.dtype == PNP_DTYPE_BF16) ? launch_taps_bf16(a, BN == 128 ? 0 : (BN == 64 ? 1 : 2), KIND, grid, st)
                                                              : launch_taps<BM, BN, WM, WN, KIND>(a, grid, st);
            PNP_REQUIRE(launched, "conv_taps_kernel: no instance for %dx%d", a.R, a.S);
            PNP_CHECK_LAUNCH("conv_taps_kernel");
        }
    }
    if (!taps) {
        static const int env_noincr = getenv("PNP_CONV_NOINCR") ? 1 : 0;
        const int kmode = mode == 0 ? 0 : (mode == 2 ? 2 : ((KIND != 2 && a.pad_mode == PNP_PAD_ZERO && a.C >= 16 && !env_noincr) ? 4 : 1));
        PnpProfScope ps(prof_class(KIND), st, conv_flops(a), conv_bytes(a), "conv_fwd_kernel<%d, %d, %d, %d, %d, %d, %s>", BM, BN, WM, WN,
                        kmode, KIND, VECB ? "true" : "false");
        if (kmode == 0) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 0, KIND, VECB>), grid, dim3(NTHREADS), 0, st, a);
        else if (kmode == 4) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 4, KIND, VECB>), grid, dim3(NTHREADS), 0, st, a);
        else if (kmode == 1) hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 1, KIND, VECB>), grid, dim3(NTHREADS), 0, st, a);
        else hipLaunchKernelGGL((conv_fwd_kernel<BM, BN, WM, WN, 2, KIND, VECB>), grid, dim3(NTHREADS), 0, st, a);
    }
    PNP_CHECK_LAUNCH("conv_fwd_kernel");
    if (nsplit > 1) {
        const size_t nout = (size_t)a.M * a.K;
        int nb = pnp_cdiv((long long)nout, 256);
        if (nb > 4096) nb = 4096;
        if (drop_in_reduce) {
            hipLaunchKernelGGL(splitk_reduce_drop_kernel, dim3(nb), dim3(256), 0, st, (const float*)split_ws, final_out, nout, nsplit, nout,
                               a.drop_key, a.drop_thresh, a.drop_keep, a.sp, a.drop_sid);
        } else if (a.o_s != 0) {
            ConvArgs ar = a;
            ar.y = final_out;
            hipLaunchKernelGGL(splitk_reduce_scatter_kernel, dim3(nb), dim3(256), 0, st, (const float*)split_ws, ar, nsplit, nout);
        } else {
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, st, (const float*)split_ws, final_out, nout, nsplit, nout, 0);
        }
        PNP_CHECK_LAUNCH("splitk_reduce_kernel");
    }
    return PNP_OK;
}

template <int KIND>
int launch_fwd(ConvArgs& a, hipStream_t st, float* split_ws = nullptr) {
    const int tile = choose_tile(a.M, a.K);
    const int nsplit = split_ws ? choose_split(a.M, a.K, a.Kred, tile) : 1;
    if (tile == 3) return launch_fwd_tile<128, 32, 4, 1, KIND, false>(a, split_ws, nsplit, st);
    if (tile == 0) return launch_fwd_tile<128, 128, 2, 2, KIND, true>(a, split_ws, nsplit, st);
    if (tile == 1) return launch_fwd_tile<128, 64, 2, 2, KIND, true>(a, split_ws, nsplit, st);
    return launch_fwd_tile<128, 32, 4, 1, KIND, true>(a, split_ws, nsplit, st);
}

// wgrad reduction split: enough workgroups for >= 1 dispatch round (512 slots), as few nearly-empty last rounds as possible
// (512->512: 144 tiles x 8 splits = 2.25 rounds costs 3; x 7 = 1.97 rounds costs 2), few splits preferred (partials are re-read).
int wgrad_plan_split(int nblk, int nchunks) {
    int max_split = nchunks / 8;
    if (max_split < 1) max_split = 1;
    const int ns_lo = pnp_cdiv(512, nblk);                      // splits needed to fill one dispatch round
    if (ns_lo >= max_split) return max_split;
    int ns_hi = ns_lo * 3 + 2;                                  // look up to ~3 rounds
    if (ns_hi > max_split) ns_hi = max_split;
    int best = ns_lo;
    double best_score = -1.0;
    for (int ns = ns_lo; ns <= ns_hi; ++ns) {
        const double rounds = (double)nblk * ns / 512.0;
        const double eff = rounds / ceil(rounds);               // fill of the dispatch rounds
        const double score = eff - 0.02 * rounds;               // mild preference for fewer, longer workgroups
        if (score > best_score) { best_score = score; best = ns; }
    }
    return best;
}

// global-load stages in flight in the linear filter-gradient kernel (conv_wgrad_ring_kernel); measured at B=16, see DESIGN.md §4.1
int wgrad_ring_depth() {
    static const int d = getenv("PNP_WGRAD_DEPTH") ? atoi(getenv("PNP_WGRAD_DEPTH")) : 2;
    return d < 1 ? 1 : (d > 3 ? 3 : d);
}

template <int BM, int BN, int WM, int WN, bool VECB>
int launch_wgrad_tile(ConvArgs& a, float* dw, float* ws, size_t ws_bytes, hipStream_t st) {
    a.nblk_m = pnp_cdiv(a.Kred, BM);
    a.nblk_n = pnp_cdiv(a.K, BN);
    const int nblk = a.nblk_m * a.nblk_n;
    const int nchunks = pnp_cdiv(a.M, BK);
    int nsplit = wgrad_plan_split(nblk, nchunks);
    const size_t nout = (size_t)a.Kred * a.K;
    if (nsplit > 1 && ws_bytes < nsplit * nout * sizeof(float)) {
        nsplit = (int)(ws_bytes / (nout * sizeof(float)));
        if (nsplit < 2) nsplit = 1;
    }
    a.chunks_per_split = pnp_cdiv(nchunks, nsplit);
    nsplit = pnp_cdiv(nchunks, a.chunks_per_split);
    a.nsplit = nsplit;
    a.split_stride = (long long)nout;
    a.y = (nsplit > 1) ? ws : dw;
    const int accumulate = a.accumulate;           // un-split: in the kernel's epilogue; split: in the reduce kernel
    if (nsplit > 1) a.accumulate = 0;
    dim3 grid((unsigned)(nblk * nsplit));
    static const int env_nolin = getenv("PNP_CONV_NOLIN") ? 1 : 0;
    // conv_wgrad_ring_kernel walks strided outputs too, and maps whose rows are shorter than a stage when a stage is a whole number of
    // rows inside one image (16^2, 8^2: the critics' last blocks ran on conv_wgrad_kernel's per-row divisions at 0.49 of peak before)
    const bool rows_ok = a.OW >= BK || ((BK % a.OW) == 0 && a.OH > BK / a.OW && (a.OHW % BK) == 0);
    const bool lin_any = !env_nolin && a.pad_mode == PNP_PAD_ZERO && (a.C % 4) == 0 && rows_ok && a.x_bytes < 0x80000000u &&
                         a.w_bytes < 0x80000000u;
    const bool lin = lin_any && a.stride == 1 && a.OW >= BK;
    if (lin && VECB && a.dtype == PNP_DTYPE_BF16) {
        const bool launched = launch_wgrad_bf16(a, BN == 128 ? 0 : (BN == 64 ? 1 : 2), grid, st);
        PNP_REQUIRE(launched, "conv_wgrad_bf16_kernel: no instance for a %dx%d tile", BM, BN);
    } else if (lin_any && VECB && wgrad_ring_depth() > 1) {
        // linear pixel walk with DEPTH global-load stages in flight (PNP_WGRAD_DEPTH = 1: conv_wgrad_kernel MODE 3, one stage in flight)
        const int depth = wgrad_ring_depth();
        static const int env_uni = getenv("PNP_WGRAD_UNI") ? atoi(getenv("PNP_WGRAD_UNI")) : 1;
        const bool uni = PNP_WGRAD_UNIFORM_ROWS != 0 && BM == 128 && env_uni && depth == 2 && (a.C % BM) == 0;
        // (the symbol rocprofv3 prints: the scalar-row variant carries 10 + depth as its last template argument)
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), conv_bytes(a), "conv_wgrad_ring_kernel<%d, %d, %d, %d, %d>", BM, BN, WM, WN,
                        uni ? 10 + depth : depth);
        if constexpr (VECB) {
            bool done = false;
            if constexpr (PNP_WGRAD_UNIFORM_ROWS != 0 && BM == 128) {          // scalar loader rows where a tile holds one tap
                if (uni) {
                    hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 12>), grid, dim3(NTHREADS), 0, st, a);
                    done = true;
                }
            }
            if (done) {
            } else if (depth == 2) hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 2>), grid, dim3(NTHREADS), 0, st, a);
            else hipLaunchKernelGGL((conv_wgrad_ring_kernel<BM, BN, WM, WN, 3>), grid, dim3(NTHREADS), 0, st, a);
        }
    } else {
        const int kmode = lin ? 3 : ((a.C % 4 == 0) ? 1 : 2);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), conv_bytes(a), "conv_wgrad_kernel<%d, %d, %d, %d, %d, %s>", BM, BN, WM, WN,
                        kmode, VECB ? "true" : "false");
        if (lin) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, 3, VECB>), grid, dim3(NTHREADS), 0, st, a);
        else if (a.C % 4 == 0) hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, 1, VECB>), grid, dim3(NTHREADS), 0, st, a);
        else hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, 2, VECB>), grid, dim3(NTHREADS), 0, st, a);
    }
    PNP_CHECK_LAUNCH("conv_wgrad_kernel");
    if (nsplit > 1) return launch_splitk_reduce(ws, dw, (size_t)nout, nsplit, accumulate, st);
    return PNP_OK;
}

// ---- direct forward for K <= 8 (conv_fwd_narrow_kernel): applicability + launch -------------------------------------------------
bool narrow_fwd_ok(const pnp_conv_geom* g, NarrowArgs* na) {
    static const int off = getenv("PNP_CONV_NONARROW") ? 1 : 0;
    if (off || g->K > 16 || g->stride != 1 || g->pad_mode != PNP_PAD_ZERO || (g->C & 3) != 0 || g->C < 8) return false;
    if ((long long)g->N * g->OH * g->OW < 8192) return false;
    // 9..16 outputs: the MFMA tile is half full, the vector ALUs only win while the work per pixel is small (16->16 3x3: 0.154 -> 0.130 ms;
    // 32->16 3x3, twice the work: 0.062 -> 0.071)
    if (g->K > 8 && (long long)g->R * g->S * g->C * g->K > 2304) return false;
    const int PH = 8 + (g->R - 1) * g->dil, PW = 32 + (g->S - 1) * g->dil;
    const int CP = ((g->C >> 2) & 1) ? g->C : g->C + 4;
    if ((size_t)PH * PW * CP * sizeof(float) > 150 * 1024) return false;
    if (na) { na->PH = PH; na->PW = PW; na->CP = CP; }
    return true;
}

template <int KK, bool EXACT>
int launch_narrow_inst(const NarrowArgs& na, dim3 grid, size_t lds, hipStream_t st) {
    PNP_REQUIRE(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_fwd_narrow_kernel<KK, EXACT>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess,
                "conv_fwd_narrow_kernel: cannot reserve %zu bytes of LDS", lds);
    const double npx = (double)na.N * na.OH * na.OW, red = (double)na.R * na.S * na.C;
    PnpProfScope ps(PNP_PROF_CONV_DIRECT, st, 2.0 * npx * red * na.K, 4.0 * ((double)na.N * na.H * na.W * na.C + npx * na.K + red * na.K),
                    "conv_fwd_narrow_kernel<%d, %s>", KK, EXACT ? "true" : "false");
    hipLaunchKernelGGL((conv_fwd_narrow_kernel<KK, EXACT>), grid, dim3(256), lds, st, na);
    PNP_CHECK_LAUNCH("conv_fwd_narrow_kernel");
    return PNP_OK;
}

// x, w, y and geometry of a stride-1 zero-padded conv with K <= 16 (forward, or a data gradient expressed as one); dropout from `a`
int launch_narrow(const float* x, const float* w, float* y, const pnp_conv_geom* g, const ConvArgs& a, hipStream_t st) {
    NarrowArgs na{};
    narrow_fwd_ok(g, &na);
    na.x = x; na.w = w; na.y = y;
    na.N = g->N; na.H = g->H; na.W = g->W; na.C = g->C; na.K = g->K; na.R = g->R; na.S = g->S; na.OH = g->OH; na.OW = g->OW;
    na.dil = g->dil; na.pad_t = g->pad_t; na.pad_l = g->pad_l;
    na.do_drop = a.do_drop; na.drop_keep = a.drop_keep; na.drop_thresh = a.drop_thresh; na.drop_key = a.drop_key;
    na.sp = a.sp; na.drop_sid = a.drop_sid;
    na.x_bytes = a.x_bytes;
    const size_t lds = (size_t)na.PH * na.PW * na.CP * sizeof(float);
    dim3 grid((unsigned)pnp_cdiv(g->OW, 32), (unsigned)pnp_cdiv(g->OH, 8), (unsigned)g->N);
    if (g->K == 5) return launch_narrow_inst<5, true>(na, grid, lds, st);
    if (g->K == 16) return launch_narrow_inst<16, true>(na, grid, lds, st);
    if (g->K <= 8) return launch_narrow_inst<8, false>(na, grid, lds, st);
    return launch_narrow_inst<16, false>(na, grid, lds, st);
}

// ---- direct (vector-ALU) filter gradient for K <= 16: plan shared by the workspace query and the launch -------------------------
struct WgdPlan { int use, nblk, ppb, G, ppt; size_t ws_bytes; };

WgdPlan wgd_plan(const pnp_conv_geom* g) {
    static const int off = getenv("PNP_CONV_NODIRECT") ? 1 : 0;
    WgdPlan pl{};
    const long long P = (long long)g->N * g->OH * g->OW;
    const int npairs = g->R * g->S * g->C;
    if (off || g->K > 16 || g->pad_mode != PNP_PAD_ZERO || npairs > 1024 || P < 8192) return pl;
    pl.use = 1;
    pl.ppt = npairs <= 256 ? 1 : pnp_cdiv(npairs, 256);
    pl.G = pl.ppt == 1 ? 256 / npairs : 1;
    if (pl.G > 16) pl.G = 16;
    long long nblk = P / 128;
    if (nblk > 2048) nblk = 2048;
    pl.ppb = (int)pnp_cdiv(P, nblk);
    pl.nblk = (int)pnp_cdiv(P, pl.ppb);
    pl.ws_bytes = (size_t)pl.nblk * npairs * g->K * sizeof(float);
    return pl;
}

template <int KK, bool EXACT>
int launch_wgd(const WgdArgs& a, const WgdPlan& pl, hipStream_t st) {
    const size_t lds = pl.G > 1 ? (size_t)pl.G * a.npairs * a.K * sizeof(float) : 0;
    dim3 grid((unsigned)pl.nblk), blk(256);
    const int nquads = a.npairs / 4;
    PnpProfScope ps(PNP_PROF_CONV_DIRECT, st, 2.0 * (double)a.P * a.npairs * a.K,
                    4.0 * ((double)a.N * a.H * a.W * a.C + (double)a.P * a.K + (double)a.npairs * a.K), "wgrad_direct_kernel<%d> (all variants)", KK);
    if ((a.C % 4) == 0 && nquads >= 128 && nquads <= 512 && KK <= 8) {      // quads: one 16-byte x load per 4 pairs
        if (nquads <= 256) hipLaunchKernelGGL((wgrad_direct4_kernel<KK, 1, EXACT>), grid, blk, 0, st, a);
        else hipLaunchKernelGGL((wgrad_direct4_kernel<KK, 2, EXACT>), grid, blk, 0, st, a);
    } else if (pl.ppt == 1 && pl.G > 1) hipLaunchKernelGGL((wgrad_direct_kernel<KK, 1, EXACT, false>), grid, blk, lds, st, a);
    else if (pl.ppt == 1) hipLaunchKernelGGL((wgrad_direct_kernel<KK, 1, EXACT, true>), grid, blk, lds, st, a);
    else if (pl.ppt == 2) hipLaunchKernelGGL((wgrad_direct_kernel<KK, 2, EXACT, true>), grid, blk, lds, st, a);
    else if (pl.ppt == 3) hipLaunchKernelGGL((wgrad_direct_kernel<KK, 3, EXACT, true>), grid, blk, lds, st, a);
    else hipLaunchKernelGGL((wgrad_direct_kernel<KK, 4, EXACT, true>), grid, blk, lds, st, a);
    PNP_CHECK_LAUNCH("wgrad_direct_kernel");
    return PNP_OK;
}

size_t wgrad_ws(const pnp_conv_geom* g) {
    if (wino_wgrad_chosen(g)) return wino_wgrad_workspace_bytes(g);         // Winograd route (conv_wino.hip)
    const size_t nout = (size_t)g->R * g->S * g->C * g->K;
    if (n16_wgrad_ok(g)) return (size_t)n16_wgrad_blocks(g) * nout * sizeof(float);
    const long long P = (long long)g->N * g->OH * g->OW;
    const int bn = ((g->K & 3) != 0 || g->K <= 32) ? 32 : (g->K > 64 ? 128 : 64);
    const int nblk = pnp_cdiv((long long)g->R * g->S * g->C, 128) * pnp_cdiv(g->K, bn);
    const int nsplit = wgrad_plan_split(nblk, pnp_cdiv(P, BK));
    const size_t mfma_ws = nsplit <= 1 ? 0 : (size_t)nsplit * nout * sizeof(float);
    const WgdPlan pl = wgd_plan(g);
    return (pl.use && pl.ws_bytes > mfma_ws) ? pl.ws_bytes : mfma_ws;
}

// ---- strided data gradient, one stride-phase at a time ---------------------------------------------------------------------
// dx[h] only receives filter taps r with r = (h + pad) mod stride (mod stride), so the pixels of one residue class ("phase") form an
// ordinary stride-1 convolution of dy with the sub-filter {r = a, a+stride, ...}:
//     dx[h0 + stride*i] = sum_t dy[i + q - t] * W[a + stride*t],   q = (h0 + pad - a) / stride
// i.e. T = ceil((R-a)/stride) taps, zero padding T-1-q, output scattered with pixel stride `stride`.  The stride^2 phases together
// perform exactly the useful multiply-adds; the zero-upsampled formulation (KIND 2) multiplies stride^2 - 1 zeros for every product.

}  // namespace

namespace pnpconv {

// -> number of phases (0: decomposition not applicable, use the zero-upsampled kernel)
int plan_phases(const pnp_conv_geom* g, DgradPhase* ph) {
    static const int off = getenv("PNP_CONV_NOPHASE") ? 1 : 0;
    const int st = g->stride;
    if (off || st < 2 || st > 4 || g->dil != 1 || g->R < st || g->S < st) return 0;
    const bool sym = g->pad_mode == PNP_PAD_SYMMETRIC;
    const int Ho = sym ? g->H + 2 * g->pad_t : g->H, Wo = sym ? g->W + 2 * g->pad_l : g->W;
    const int fpt = sym ? 0 : g->pad_t, fpl = sym ? 0 : g->pad_l;
    int n = 0;
    size_t off_f = 0;
    for (int a = 0; a < st; ++a)
        for (int b = 0; b < st; ++b) {
            DgradPhase p{};
            p.pa = a; p.pb = b;
            p.T = (g->R - a + st - 1) / st;
            p.U = (g->S - b + st - 1) / st;
            p.h0 = (((a - fpt) % st) + st) % st;
            p.w0 = (((b - fpl) % st) + st) % st;
            const int qa = (p.h0 + fpt - a) / st, qb = (p.w0 + fpl - b) / st;
            p.pad_t = p.T - 1 - qa;
            p.pad_l = p.U - 1 - qb;
            if (p.pad_t < 0 || p.pad_l < 0) return 0;
            p.I = p.h0 < Ho ? (Ho - 1 - p.h0) / st + 1 : 0;
            p.J = p.w0 < Wo ? (Wo - 1 - p.w0) / st + 1 : 0;
            p.wt_off = off_f;
            off_f += (size_t)p.T * p.U * g->K * g->C;
            ph[n++] = p;
        }
    return n;
}

pnp_conv_geom phase_geom(const pnp_conv_geom* g, const DgradPhase& p) {
    pnp_conv_geom d{};
    d.N = g->N; d.H = g->OH; d.W = g->OW; d.C = g->K; d.K = g->C; d.R = p.T; d.S = p.U;
    d.OH = p.I; d.OW = p.J;
    d.stride = 1; d.dil = 1;
    d.pad_t = p.pad_t; d.pad_l = p.pad_l;
    d.pad_mode = PNP_PAD_ZERO;
    d.dtype = g->dtype;
    return d;
}

}  // namespace pnpconv

namespace {

// All phases in one launch (conv_dgrad_phases_kernel) when no single phase fills the chip: the phases' pixel tiles x 64-wide channel
// tiles together stay under one dispatch round (512 slots).  Needs the general kernel's fast mode (K % 32 == 0 channels of dy) and
// float4 filter rows (C % 4 == 0).
int phase_group_tiles(const pnp_conv_geom* g, const DgradPhase* ph, int nph, int bn) {
    int tiles = 0;
    for (int i = 0; i < nph; ++i)
        if (ph[i].I > 0 && ph[i].J > 0) tiles += pnp_cdiv((long long)g->N * ph[i].I * ph[i].J, 128) * pnp_cdiv(g->C, bn);
    return tiles;
}
bool phases_in_one_launch(const pnp_conv_geom* g, const DgradPhase* ph, int nph) {
    static const int off = getenv("PNP_CONV_NOPHASEGROUP") ? 1 : 0;
    if (off || nph < 2 || nph > 16 || (g->K % 32) != 0 || (g->C % 4) != 0 || g->C < 32) return false;
    if ((size_t)g->N * g->OH * g->OW * g->K * sizeof(float) >= 0x80000000u) return false;
    return phase_group_tiles(g, ph, nph, 64) <= 512;
}

int launch_phase_group(const float* dy, const float* wt, float* outp, const pnp_conv_geom* g, const DgradPhase* ph, int nph, int Ho, int Wo,
                       hipStream_t st) {
    GroupArgs ga{};
    pnp_conv_geom d0 = phase_geom(g, ph[0]);
    ga.base = make_args(dy, wt, outp, &d0);
    ga.base.o_s = g->stride; ga.base.o_H = Ho; ga.base.o_W = Wo;
    const bool narrow = phase_group_tiles(g, ph, nph, 64) < 256;          // more, narrower tiles when even 64-wide ones leave CUs idle
    const int bn = narrow ? 32 : 64;
    ga.base.nblk_n = pnp_cdiv(g->C, bn);
    int blk = 0, n = 0;
    double flops = 0.0, bytes = 0.0;
    for (int i = 0; i < nph; ++i) {
        const DgradPhase& p = ph[i];
        if (p.I == 0 || p.J == 0) continue;
        PhaseDesc& q = ga.ph[n++];
        q.w_off = (int)p.wt_off;
        q.R = p.T; q.S = p.U; q.OH = p.I; q.OW = p.J; q.pad_t = p.pad_t; q.pad_l = p.pad_l; q.o_h0 = p.h0; q.o_w0 = p.w0;
        q.first_blk = blk;
        q.nblk_m = pnp_cdiv((long long)g->N * p.I * p.J, 128);
        blk += q.nblk_m * ga.base.nblk_n;
        const double rows = (double)g->N * p.I * p.J;
        flops += 2.0 * rows * g->C * (double)(p.T * p.U * g->K);
        bytes += 4.0 * (rows * g->C + (double)p.T * p.U * g->K * g->C);
    }
    ga.nph = n;
    if (n == 0) return PNP_OK;
    bytes += 4.0 * (double)g->N * g->OH * g->OW * g->K;
    PnpProfScope ps(PNP_PROF_CONV_DGRAD, st, flops, bytes, "conv_dgrad_phases_kernel<128, %d, %d, %d>", bn, narrow ? 4 : 2, narrow ? 1 : 2);
    if (narrow) hipLaunchKernelGGL((conv_dgrad_phases_kernel<128, 32, 4, 1>), dim3((unsigned)blk), dim3(NTHREADS), 0, st, ga);
    else hipLaunchKernelGGL((conv_dgrad_phases_kernel<128, 64, 2, 2>), dim3((unsigned)blk), dim3(NTHREADS), 0, st, ga);
    PNP_CHECK_LAUNCH("conv_dgrad_phases_kernel");
    return PNP_OK;
}

}  // namespace

extern "C" {

// forward: only layers that leave at least 3/4 of the workgroup slots empty — at half a dispatch round the extra pass over the partials
// costs more than the idle CUs (128->128@32^2, 256 tiles: 0.049 ms unsplit, 0.060 split in two)
static int fwd_split(const pnp_conv_geom* g) {
    if (g->pad_mode != PNP_PAD_ZERO) return 1;
    const long long M = (long long)g->N * g->OH * g->OW;
    const int tile = choose_tile(M, g->K);
    const int bn = tile == 0 ? 128 : (tile == 1 ? 64 : 32);
    if ((long long)pnp_cdiv(M, 128) * pnp_cdiv(g->K, bn) > 128) return 1;
    return choose_split(M, g->K, g->R * g->S * g->C, tile);
}

size_t pnp_conv2d_fwd_workspace_bytes(const pnp_conv_geom* g) {
    if (!g) return 0;
    if (wino_chosen(g)) return wino_workspace_bytes(g);           // Winograd route: transformed filter + input + product (conv_wino.hip)
    const int ns = fwd_split(g);
    const size_t split = ns > 1 ? (size_t)ns * g->N * g->OH * g->OW * g->K * sizeof(float) : 0;
    // direct split-bf16 convolution of a narrow layer the Winograd planner leaves alone (conv_x3_direct.hip): its filter image
    const size_t x3d = x3d_chosen(g) ? x3d_filter_bytes(g->C, g->K) : 0;
    return split > x3d ? split : x3d;
}

int pnp_conv2d_fwd(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                   uint32_t stream_id, void* stream) {
    return pnp_conv2d_fwd_ws(x, w, y, g, keep_prob, seed, stream_id, nullptr, 0, stream);
}

int pnp_conv2d_fwd_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                      uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_geom(g, "pnp_conv2d_fwd")) return e;
    PNP_REQUIRE(x && w && y, "pnp_conv2d_fwd: null pointer");
    PNP_REQUIRE(keep_prob > 0.f, "pnp_conv2d_fwd: keep_prob must be > 0");
    ConvArgs a = make_args(x, w, y, g);
    if (keep_prob < 1.f) {
        a.do_drop = 1;
        a.drop_keep = keep_prob;
        a.drop_key = pnp_drop_key(seed, stream_id);
        a.drop_thresh = pnp_drop_thresh(keep_prob);
        a.sp = pnp_step_params_ptr(); a.drop_sid = stream_id;
    }
    if (n16_geom_ok(g)) return launch_n16_fwd(a, 0, (hipStream_t)stream);
    if (narrow_fwd_ok(g, nullptr)) return launch_narrow(x, w, y, g, a, (hipStream_t)stream);
    if (wino_chosen(g)) {          // (without the workspace the direct kernel runs: same result up to fp32 rounding)
        if (workspace && workspace_bytes >= wino_workspace_bytes(g)) return launch_wino(a, 0, false, workspace, workspace_bytes, (hipStream_t)stream);
        return launch_fwd<0>(a, (hipStream_t)stream, nullptr);
    }
    if (x3d_chosen(a) && workspace && workspace_bytes >= x3d_filter_bytes(a.C, a.K))
        return launch_x3_direct(a, 0, false, workspace, workspace_bytes, (hipStream_t)stream);
    float* split_ws = (workspace && workspace_bytes >= pnp_conv2d_fwd_workspace_bytes(g) && pnp_conv2d_fwd_workspace_bytes(g) > 0)
                          ? (float*)workspace : nullptr;
    return launch_fwd<0>(a, (hipStream_t)stream, split_ws);
}

// ---- forward convolution that also leaves the batch-norm statistics partials of its output (training-mode conv -> dropout -> BN) ----
// Only on the MFMA kernels with an un-split reduction (the epilogue owns complete output rows there); 0 parts = not available for
// this geometry (narrow-output vector-ALU kernels, reduction-split tiny layers): the caller runs pnp_bn_stats on the output instead.
static int fwd_stats_impl(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                          uint32_t stream_id, const float* shift, float* parts, size_t parts_bytes, void* workspace, size_t workspace_bytes,
                          void* stream, bool with_ws);

int32_t pnp_conv2d_fwd_stats_parts(const pnp_conv_geom* g) {
    if (!g || check_geom(g, "pnp_conv2d_fwd_stats_parts") != PNP_OK) return 0;
    if (n16_geom_ok(g) || narrow_fwd_ok(g, nullptr) || fwd_split(g) > 1) return 0;
    const long long M = (long long)g->N * g->OH * g->OW;
    const int tile = choose_tile(M, g->K);
    return pnp_cdiv(M, 128) * ((tile == 0 || tile == 1) ? 2 : 4);        // pixel tiles x wave rows of the tile (WM)
}

int pnp_conv2d_fwd_stats(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                         uint32_t stream_id, const float* shift, float* parts, size_t parts_bytes, void* stream) {
    return fwd_stats_impl(x, w, y, g, keep_prob, seed, stream_id, shift, parts, parts_bytes, nullptr, 0, stream, false);
}

// The same with a workspace (pnp_conv2d_fwd_workspace_bytes): layers the planner gives to the Winograd route (conv_wino.hip) leave one
// partial row per tile slab of its output transform — pnp_conv2d_fwd_stats_ws_parts says how many; every other layer is
// pnp_conv2d_fwd_stats / pnp_conv2d_fwd_stats_parts.
int32_t pnp_conv2d_fwd_stats_ws_parts(const pnp_conv_geom* g) {
    if (!g || check_geom(g, "pnp_conv2d_fwd_stats_ws_parts") != PNP_OK) return 0;
    if (wino_chosen(g)) return wino_stats_parts(g);
    if (x3d_chosen(g) && !n16_geom_ok(g) && !narrow_fwd_ok(g, nullptr)) return x3d_stats_parts(g);
    return pnp_conv2d_fwd_stats_parts(g);
}

int pnp_conv2d_fwd_stats_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                            uint32_t stream_id, const float* shift, float* parts, size_t parts_bytes, void* workspace,
                            size_t workspace_bytes, void* stream) {
    return fwd_stats_impl(x, w, y, g, keep_prob, seed, stream_id, shift, parts, parts_bytes, workspace, workspace_bytes, stream, true);
}

static int fwd_stats_impl(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                          uint32_t stream_id, const float* shift, float* parts, size_t parts_bytes, void* workspace, size_t workspace_bytes,
                          void* stream, bool with_ws) {
    if (int e = check_geom(g, "pnp_conv2d_fwd_stats")) return e;
    PNP_REQUIRE(x && w && y && parts, "pnp_conv2d_fwd_stats: null pointer");
    PNP_REQUIRE(keep_prob > 0.f, "pnp_conv2d_fwd_stats: keep_prob must be > 0");
    const bool wino = with_ws && wino_chosen(g);
    if (wino) PNP_REQUIRE(workspace && workspace_bytes >= wino_workspace_bytes(g), "pnp_conv2d_fwd_stats_ws: workspace too small (pnp_conv2d_fwd_workspace_bytes)");
    const bool x3d = with_ws && !wino && x3d_chosen(g) && !n16_geom_ok(g) && !narrow_fwd_ok(g, nullptr);
    if (x3d) PNP_REQUIRE(workspace && workspace_bytes >= x3d_filter_bytes(g->C, g->K), "pnp_conv2d_fwd_stats_ws: workspace too small (pnp_conv2d_fwd_workspace_bytes)");
    const int nparts = wino ? wino_stats_parts(g) : (x3d ? x3d_stats_parts(g) : pnp_conv2d_fwd_stats_parts(g));
    PNP_REQUIRE(nparts > 0, "pnp_conv2d_fwd_stats: no epilogue statistics for this geometry (pnp_conv2d_fwd_stats_parts == 0)");
    if (parts_bytes < (size_t)nparts * 2 * g->K * sizeof(float)) {
        pnp_set_error("pnp_conv2d_fwd_stats: parts buffer too small (%zu < %zu)", parts_bytes, (size_t)nparts * 2 * g->K * sizeof(float));
        return PNP_EWORKSPACE;
    }
    ConvArgs a = make_args(x, w, y, g);
    if (keep_prob < 1.f) {
        a.do_drop = 1;
        a.drop_keep = keep_prob;
        a.drop_key = pnp_drop_key(seed, stream_id);
        a.drop_thresh = pnp_drop_thresh(keep_prob);
        a.sp = pnp_step_params_ptr(); a.drop_sid = stream_id;
    }
    a.stat_ws = parts;
    a.stat_shift = shift;
    if (wino) return launch_wino(a, 0, false, workspace, workspace_bytes, (hipStream_t)stream);
    if (x3d) return launch_x3_direct(a, 0, false, workspace, workspace_bytes, (hipStream_t)stream);
    return launch_fwd<0>(a, (hipStream_t)stream);
}

int pnp_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, float* scale, float* shift, int32_t C,
                float eps, void* stream) {
    PNP_REQUIRE(gamma && beta && mean && var && scale && shift && C > 0, "pnp_bn_fold: bad argument");
    hipLaunchKernelGGL(bn_fold_kernel, dim3((unsigned)pnp_cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, scale,
                       shift, C, eps);
    PNP_CHECK_LAUNCH("bn_fold_kernel");
    return PNP_OK;
}

int pnp_conv2d_fwd_bn(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                      uint32_t stream_id, const float* scale, const float* shift, const float* shortcut, int32_t Cs, float alpha,
                      void* stream) {
    return pnp_conv2d_fwd_bn_ws(x, w, y, g, keep_prob, seed, stream_id, scale, shift, shortcut, Cs, alpha, nullptr, 0, stream);
}

// with a workspace of pnp_conv2d_fwd_workspace_bytes(g) bytes the layers the planner gives to the Winograd route take it (conv_wino.hip)
int pnp_conv2d_fwd_bn_ws(const float* x, const float* w, float* y, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                         uint32_t stream_id, const float* scale, const float* shift, const float* shortcut, int32_t Cs, float alpha,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_geom(g, "pnp_conv2d_fwd_bn")) return e;
    PNP_REQUIRE(x && w && y && scale && shift, "pnp_conv2d_fwd_bn: null pointer");
    PNP_REQUIRE(keep_prob > 0.f, "pnp_conv2d_fwd_bn: keep_prob must be > 0");
    if (shortcut) PNP_REQUIRE(Cs > 0 && Cs <= g->K && ((g->K - Cs) % 2) == 0, "pnp_conv2d_fwd_bn: bad shortcut channels");
    ConvArgs a = make_args(x, w, y, g);
    if (keep_prob < 1.f) {
        a.do_drop = 1;
        a.drop_keep = keep_prob;
        a.drop_key = pnp_drop_key(seed, stream_id);
        a.drop_thresh = pnp_drop_thresh(keep_prob);
        a.sp = pnp_step_params_ptr(); a.drop_sid = stream_id;
    }
    a.ep_scale = scale; a.ep_shift = shift; a.ep_res = shortcut; a.ep_cs = shortcut ? Cs : g->K; a.ep_alpha = alpha;
    if (n16_geom_ok(g)) return launch_n16_fwd(a, 0, (hipStream_t)stream);
    if (wino_chosen(g) && workspace && workspace_bytes >= wino_workspace_bytes(g))
        return launch_wino(a, 0, false, workspace, workspace_bytes, (hipStream_t)stream);
    return launch_fwd<0>(a, (hipStream_t)stream);      // the reduction is never split on this path
}

int pnp_conv2d_fwd_naive(const float* x, const float* w, float* y, const pnp_conv_geom* g, void* stream) {
    if (int e = check_geom(g, "pnp_conv2d_fwd_naive")) return e;
    ConvArgs a = make_args(x, w, y, g);
    const size_t total = (size_t)a.M * a.K;
    hipLaunchKernelGGL(naive_conv_kernel, dim3((unsigned)pnp_cdiv((long long)total, 256)), dim3(256), 0,
                       (hipStream_t)stream, a);
    PNP_CHECK_LAUNCH("naive_conv_kernel");
    return PNP_OK;
}

// dgrad workspace: [flipped/transposed filters R*S*K*C] [+ padded dx for SYMMETRIC]
// the data gradient of a stride-1 zero-padded convolution AS a convolution of dy (flipped, transposed filter)
static pnp_conv_geom dgrad_as_conv(const pnp_conv_geom* g) {
    pnp_conv_geom d{};
    d.N = g->N; d.H = g->OH; d.W = g->OW; d.C = g->K; d.K = g->C; d.R = g->R; d.S = g->S;
    d.OH = g->H; d.OW = g->W;
    d.stride = 1; d.dil = g->dil;
    d.pad_t = g->dil * (g->R - 1) - g->pad_t;
    d.pad_l = g->dil * (g->S - 1) - g->pad_l;
    d.pad_mode = PNP_PAD_ZERO;
    d.dtype = g->dtype;
    return d;
}
static bool dgrad_wino(const pnp_conv_geom* g, pnp_conv_geom* d) {
    if (g->stride != 1 || g->pad_mode != PNP_PAD_ZERO) return false;
    *d = dgrad_as_conv(g);
    return d->pad_t >= 0 && d->pad_l >= 0 && wino_chosen(d);
}

// 1: the planner gives this layer to the Winograd route (kind 0: forward; 1: data gradient; 2: filter gradient — g = the FORWARD geometry); 0: direct kernels
int32_t pnp_conv2d_wino_chosen(const pnp_conv_geom* g, int32_t kind) {
    if (!g || check_geom(g, "pnp_conv2d_wino_chosen") != PNP_OK) return 0;
    pnp_conv_geom d;
    if (kind == 2) return wino_wgrad_tile(g);
    return kind == 0 ? wino_tile(g) : (dgrad_wino(g, &d) ? wino_tile(&d) : 0);
}

// the data gradient as a direct split-bf16 convolution of dy (conv_x3_direct.hip) where the Winograd planner leaves the layer alone
static bool dgrad_x3d(const pnp_conv_geom* g, pnp_conv_geom* d) {
    if (g->stride != 1 || g->pad_mode != PNP_PAD_ZERO) return false;
    *d = dgrad_as_conv(g);
    return d->pad_t >= 0 && d->pad_l >= 0 && !wino_chosen(d) && !n16_geom_ok(d) && !narrow_fwd_ok(d, nullptr) && x3d_chosen(d);
}

static size_t dgrad_ws_base(const pnp_conv_geom* g);

size_t pnp_conv2d_dgrad_workspace_bytes(const pnp_conv_geom* g) {
    if (!g) return 0;
    const size_t b = dgrad_ws_base(g);
    pnp_conv_geom d;
    if (dgrad_x3d(g, &d)) {
        const size_t f = x3d_filter_bytes(d.C, d.K);
        return b > f ? b : f;
    }
    return b;
}

static size_t dgrad_ws_base(const pnp_conv_geom* g) {
    {
        pnp_conv_geom d;
        if (dgrad_wino(g, &d)) return wino_workspace_bytes(&d);
    }
    size_t b = (size_t)g->R * g->S * g->C * g->K * sizeof(float);
    b = (b + 255) & ~(size_t)255;
    size_t outb = (size_t)g->N * g->H * g->W * g->C * sizeof(float);
    if (g->pad_mode == PNP_PAD_SYMMETRIC) {
        outb = (size_t)g->N * (g->H + 2 * g->pad_t) * (g->W + 2 * g->pad_l) * g->C * sizeof(float);
        b += (outb + 255) & ~(size_t)255;
    }
    // reduction-split partials of the data-gradient GEMM (M = output pixels of the dgrad, N = C, reduction R*S*K)
    DgradPhase ph[16];
    const int nph = plan_phases(g, ph);
    if (nph > 0) {                                       // one stride-phase at a time: the largest phase's partials
        size_t mx = 0;
        for (int i = 0; i < nph; ++i) {
            const long long Mp = (long long)g->N * ph[i].I * ph[i].J;
            if (Mp == 0) continue;
            const int ns = choose_split(Mp, g->C, ph[i].T * ph[i].U * g->K, choose_tile(Mp, g->C));
            if (ns > 1 && (size_t)ns * Mp * g->C * sizeof(float) > mx) mx = (size_t)ns * Mp * g->C * sizeof(float);
        }
        return b + mx;
    }
    const long long Md = (long long)(outb / sizeof(float)) / g->C;
    const int ns = choose_split(Md, g->C, g->R * g->S * g->K, choose_tile(Md, g->C));
    if (ns > 1) b += (size_t)ns * outb;
    return b;
}

static int dgrad_impl(const float* dy, const float* w, float* dx, const pnp_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream,
                      const float* residual);

int pnp_conv2d_dgrad(const float* dy, const float* w, float* dx, const pnp_conv_geom* g, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return dgrad_impl(dy, w, dx, g, workspace, workspace_bytes, stream, nullptr);
}

int pnp_conv2d_dgrad_add(const float* dy, const float* w, const float* residual, float* dx, const pnp_conv_geom* g, void* workspace,
                         size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(residual && residual != dx, "pnp_conv2d_dgrad_add: residual must be a tensor of its own");
    return dgrad_impl(dy, w, dx, g, workspace, workspace_bytes, stream, residual);
}

// residual != null: dx = data gradient + residual.  Fused into the epilogue on the stride-1 MFMA paths (the residual blocks); every
// other geometry computes the gradient and adds the residual with one more pass (pnp_axpby).
static int dgrad_impl(const float* dy, const float* w, float* dx, const pnp_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream,
                      const float* residual) {
    if (int e = check_geom(g, "pnp_conv2d_dgrad")) return e;
    PNP_REQUIRE(dy && w && dx && workspace, "pnp_conv2d_dgrad: null pointer");
    PNP_REQUIRE(workspace_bytes >= pnp_conv2d_dgrad_workspace_bytes(g), "pnp_conv2d_dgrad: workspace too small");
    if (g->pad_mode == PNP_PAD_SYMMETRIC) PNP_REQUIRE(g->pad_t == g->pad_l, "pnp_conv2d_dgrad: SYMMETRIC needs pad_t == pad_l");
    hipStream_t st = (hipStream_t)stream;
    float* wt = (float*)workspace;
    size_t woff = ((size_t)g->R * g->S * g->C * g->K * sizeof(float) + 255) & ~(size_t)255;
    const size_t dx_elems = (size_t)g->N * g->H * g->W * g->C;
    auto add_residual = [&](float* out, const float* res) -> int {
        return res ? pnp_axpby(res, out, dx_elems, 1.f, 1.f, stream) : PNP_OK;
    };
    DgradPhase ph[16];
    const int nph = plan_phases(g, ph);
    if (nph > 0) {
        const bool symp = g->pad_mode == PNP_PAD_SYMMETRIC;
        const int Ho = symp ? g->H + 2 * g->pad_t : g->H, Wo = symp ? g->W + 2 * g->pad_l : g->W;
        float* outp = symp ? (float*)((char*)workspace + woff) : dx;
        size_t poff = woff;
        if (symp) poff += ((size_t)g->N * Ho * Wo * g->C * sizeof(float) + 255) & ~(size_t)255;
        float* split_ws = (workspace_bytes > poff) ? (float*)((char*)workspace + poff) : nullptr;
        dim3 tgp((unsigned)pnp_cdiv(g->K, 32), (unsigned)pnp_cdiv(g->C, 32), (unsigned)(g->R * g->S));
        hipLaunchKernelGGL(flip_transpose_phase_kernel, tgp, dim3(256), 0, st, w, wt, g->R, g->S, g->C, g->K, g->stride);
        PNP_CHECK_LAUNCH("flip_transpose_phase_kernel");
        if (phases_in_one_launch(g, ph, nph)) {
            if (int e = launch_phase_group(dy, wt, outp, g, ph, nph, Ho, Wo, st)) return e;
        } else {
            for (int i = 0; i < nph; ++i) {
                const DgradPhase& p = ph[i];
                if (p.I == 0 || p.J == 0) continue;
                const pnp_conv_geom d = phase_geom(g, p);
                ConvArgs a = make_args(dy, wt + p.wt_off, outp, &d);
                a.o_s = g->stride; a.o_H = Ho; a.o_W = Wo; a.o_h0 = p.h0; a.o_w0 = p.w0;
                if (int e = launch_fwd<1>(a, st, split_ws)) return e;
            }
        }
        if (symp) {
            const size_t total = (size_t)g->N * g->H * g->W * g->C;
            hipLaunchKernelGGL(sympad_bwd_kernel, dim3((unsigned)pnp_cdiv((long long)total, 256)), dim3(256), 0, st,
                               (const float*)outp, dx, g->N, g->H, g->W, g->C, g->pad_t);
            PNP_CHECK_LAUNCH("sympad_bwd_kernel");
        }
        return add_residual(dx, residual);
    }
    {
        pnp_conv_geom dw_;
        if (dgrad_wino(g, &dw_)) {           // Winograd route: the filter transform flips and transposes on the way (no flip launch)
            ConvArgs a = make_args(dy, w, dx, &dw_);
            a.res_add = residual;
            return launch_wino(a, 1, true, workspace, workspace_bytes, st);
        }
    }
    {
        pnp_conv_geom dx_;
        if (dgrad_x3d(g, &dx_)) {            // direct split-bf16 convolution of dy: its filter image flips and transposes (no flip launch)
            ConvArgs a = make_args(dy, w, dx, &dx_);
            a.res_add = residual;
            return launch_x3_direct(a, 1, true, workspace, workspace_bytes, st);
        }
    }
    dim3 tg((unsigned)pnp_cdiv(g->K, 32), (unsigned)pnp_cdiv(g->C, 32), (unsigned)(g->R * g->S));
    hipLaunchKernelGGL(flip_transpose_kernel, tg, dim3(256), 0, st, w, wt, g->R, g->S, g->C, g->K);
    PNP_CHECK_LAUNCH("flip_transpose_kernel");

    // dgrad as a stride-1 convolution over dy (zero-upsampled by `stride`), output = (padded) input image
    const bool sym = g->pad_mode == PNP_PAD_SYMMETRIC;
    pnp_conv_geom d{};
    d.N = g->N; d.H = g->OH; d.W = g->OW; d.C = g->K; d.K = g->C; d.R = g->R; d.S = g->S;
    d.OH = sym ? g->H + 2 * g->pad_t : g->H;
    d.OW = sym ? g->W + 2 * g->pad_l : g->W;
    d.stride = 1; d.dil = g->dil;
    d.pad_t = g->dil * (g->R - 1) - (sym ? 0 : g->pad_t);
    d.pad_l = g->dil * (g->S - 1) - (sym ? 0 : g->pad_l);
    d.pad_mode = PNP_PAD_ZERO;
    d.dtype = g->dtype;
    PNP_REQUIRE(d.pad_t >= 0 && d.pad_l >= 0, "pnp_conv2d_dgrad: forward padding exceeds the filter extent");
    float* out = sym ? (float*)((char*)workspace + woff) : dx;
    ConvArgs a = make_args(dy, wt, out, &d);
    a.ups = g->stride;
    if (!sym && n16_geom_ok(&d)) {                           // 16 INPUT channels: the data gradient is a 16-output conv of dy
        a.res_add = residual;
        return launch_n16_fwd(a, 1, st);
    }
    if (g->stride == 1 && narrow_fwd_ok(&d, nullptr)) {       // few INPUT channels: the data gradient is a narrow-output conv of dy
        if (int e = launch_narrow(dy, wt, out, &d, a, st)) return e;
        if (sym) {
            const size_t total = (size_t)g->N * g->H * g->W * g->C;
            hipLaunchKernelGGL(sympad_bwd_kernel, dim3((unsigned)pnp_cdiv((long long)total, 256)), dim3(256), 0, st,
                               (const float*)out, dx, g->N, g->H, g->W, g->C, g->pad_t);
            PNP_CHECK_LAUNCH("sympad_bwd_kernel");
        }
        return add_residual(dx, residual);
    }
    size_t poff = woff;
    if (sym) poff += ((size_t)d.N * d.OH * d.OW * d.K * sizeof(float) + 255) & ~(size_t)255;
    float* split_ws = (workspace_bytes > poff) ? (float*)((char*)workspace + poff) : nullptr;
    const bool fuse_res = residual && !sym && g->stride == 1;          // rows of the GEMM == pixels of dx, plain row-major output
    if (fuse_res) a.res_add = residual;
    if (int e = (g->stride > 1 ? launch_fwd<2>(a, st, split_ws) : launch_fwd<1>(a, st, split_ws))) return e;
    if (sym) {
        const size_t total = (size_t)g->N * g->H * g->W * g->C;
        hipLaunchKernelGGL(sympad_bwd_kernel, dim3((unsigned)pnp_cdiv((long long)total, 256)), dim3(256), 0, st,
                           (const float*)out, dx, g->N, g->H, g->W, g->C, g->pad_t);
        PNP_CHECK_LAUNCH("sympad_bwd_kernel");
    }
    return fuse_res ? PNP_OK : add_residual(dx, residual);
}

size_t pnp_conv2d_wgrad_workspace_bytes(const pnp_conv_geom* g) { return g ? wgrad_ws(g) : 0; }

static int wgrad_impl(const float* x, const float* dy, float* dw, const pnp_conv_geom* g, void* workspace, size_t workspace_bytes, void* stream,
                      int accumulate) {
    if (int e = check_geom(g, "pnp_conv2d_wgrad")) return e;
    PNP_REQUIRE(x && dy && dw, "pnp_conv2d_wgrad: null pointer");
    ConvArgs a = make_args(x, dy, dw, g);
    a.accumulate = accumulate;
    a.w_bytes = (unsigned)((size_t)g->N * g->OH * g->OW * g->K * sizeof(float));   // a.w is dy [P][K] here
    hipStream_t st = (hipStream_t)stream;
    float* ws = (float*)workspace;
    if (!ws) workspace_bytes = 0;
    if (wino_wgrad_chosen(g) && workspace_bytes >= wino_wgrad_workspace_bytes(g))
        return launch_wino_wgrad(a, dw, accumulate, workspace, workspace_bytes, st);
    if (n16_wgrad_ok(g) && workspace_bytes >= wgrad_ws(g)) {
        if (int e = launch_n16_wgrad(a, ws, st)) return e;
        const size_t nout = (size_t)a.Kred * a.K;
        hipLaunchKernelGGL(splitk_reduce_many_kernel, dim3((unsigned)pnp_cdiv((long long)nout, 64)), dim3(1024), 0, st,
                           (const float*)ws, dw, (int)nout, n16_wgrad_blocks(g), accumulate);
        PNP_CHECK_LAUNCH("splitk_reduce_many_kernel");
        return PNP_OK;
    }
    const WgdPlan pl = wgd_plan(g);
    if (pl.use && workspace_bytes >= pl.ws_bytes) {
        WgdArgs d{};
        d.x = x; d.dy = dy; d.part = ws;
        d.N = g->N; d.H = g->H; d.W = g->W; d.C = g->C; d.K = g->K; d.R = g->R; d.S = g->S; d.OH = g->OH; d.OW = g->OW;
        d.stride = g->stride; d.dil = g->dil; d.pad_t = g->pad_t; d.pad_l = g->pad_l;
        d.P = a.M; d.npairs = a.Kred; d.G = pl.G; d.ppb = pl.ppb;
        d.x_bytes = a.x_bytes; d.dy_bytes = a.w_bytes;
        int e;
        if (g->K == 16) e = launch_wgd<16, true>(d, pl, st);
        else if (g->K == 5) e = launch_wgd<5, true>(d, pl, st);
        else if (g->K <= 8) e = launch_wgd<8, false>(d, pl, st);
        else e = launch_wgd<16, false>(d, pl, st);
        if (e) return e;
        const size_t nout = (size_t)a.Kred * a.K;
        hipLaunchKernelGGL(splitk_reduce_many_kernel, dim3((unsigned)pnp_cdiv((long long)nout, 64)), dim3(1024), 0, st,
                           (const float*)ws, dw, (int)nout, pl.nblk, accumulate);
        PNP_CHECK_LAUNCH("splitk_reduce_many_kernel");
        return PNP_OK;
    }
    if ((a.K & 3) != 0) return launch_wgrad_tile<128, 32, 4, 1, false>(a, dw, ws, workspace_bytes, st);
    if (a.K > 64) return launch_wgrad_tile<128, 128, 2, 2, true>(a, dw, ws, workspace_bytes, st);
    if (a.K > 32) return launch_wgrad_tile<128, 64, 2, 2, true>(a, dw, ws, workspace_bytes, st);
    return launch_wgrad_tile<128, 32, 4, 1, true>(a, dw, ws, workspace_bytes, st);
}

int pnp_conv2d_wgrad(const float* x, const float* dy, float* dw, const pnp_conv_geom* g, void* workspace,
                     size_t workspace_bytes, void* stream) {
    return wgrad_impl(x, dy, dw, g, workspace, workspace_bytes, stream, 0);
}

int pnp_conv2d_wgrad_acc(const float* x, const float* dy, float* dw, const pnp_conv_geom* g, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return wgrad_impl(x, dy, dw, g, workspace, workspace_bytes, stream, 1);
}

// ---- bf16-resident filter gradient (conv_bf16r.hip): x and dy as bf16 tensors; the split planner and the partial-sum kernel are the
// fp32 path's own
static int wgrad_bf16r_plan(const pnp_conv_geom* g, int* bm, int* bn) {
    const int tile = wgrad_bf16r_tile(g);
    if (tile < 0) return 0;
    *bm = tile < 2 ? 128 : 64;
    *bn = (tile & 1) ? 64 : 128;
    const long long P = (long long)g->N * g->OH * g->OW;
    const int nblk = (g->R * g->S * g->C / *bm) * (g->K / *bn);
    static const int force = getenv("PNP_BF16R_WSPLIT") ? atoi(getenv("PNP_BF16R_WSPLIT")) : 0;      // experiments: force the split count
    if (force > 0) return force;
    const int nchunks = pnp_cdiv(P, kWgradBf16rChunk);
    // The resident kernel is ~3x faster than the fp32 ring kernel while a split partial costs the same bytes, so the fp32 planner's
    // "fill >= one dispatch round of 512" over-splits here (256->256@32^2: 15 splits 0.052 ms, 7 splits 0.048 ms; round 4 sweep,
    // tools/experiments/README.md).  Cost model, microseconds: dispatch rounds x (stages per workgroup x 0.7 x tile/128^2 + 6 fixed)
    // + partials written and read back at 4 TB/s; the split with the lowest estimate wins (ties: fewer splits).  Against the fp32
    // planner on every layer at B = 16: better or equal everywhere (64->64@64^2 0.044 -> 0.030 ms, 256->256@32^2 0.053 -> 0.041,
    // cls1 64->64@256^2 0.188 -> 0.165, 512->512 equal); bf16 joint step, same box x2: 457.0 -> 460.8 slices/s.
    const double t_stage = 0.7 * ((double)*bm * *bn) / (128.0 * 128.0);
    const double nout_mb = (double)g->R * g->S * g->C * g->K * 4.0 / 1e6;
    int best = 1;
    double best_t = 1e30;
    const int max_split = nchunks / 4 < 1 ? 1 : (nchunks / 4 > 64 ? 64 : nchunks / 4);
    for (int ns = 1; ns <= max_split; ++ns) {
        const double rounds = ceil((double)nblk * ns / 512.0);
        const double t = rounds * (ceil((double)nchunks / ns) * t_stage + 6.0) + (ns > 1 ? 2.0 * ns * nout_mb / 4.0 + 3.0 : 0.0);
        if (t < best_t - 1e-9) { best_t = t; best = ns; }
    }
    return best;
}

size_t pnp_conv2d_wgrad_bf16r_workspace_bytes(const pnp_conv_geom* g) {
    int bm, bn;
    const int ns = g ? wgrad_bf16r_plan(g, &bm, &bn) : 0;
    return ns > 1 ? (size_t)ns * g->R * g->S * g->C * g->K * sizeof(float) : 0;
}

int pnp_conv2d_wgrad_bf16r(const void* xh, const void* dyh, float* dw, int32_t accumulate, const pnp_conv_geom* g, void* workspace,
                           size_t workspace_bytes, void* stream) {
    if (int e = check_geom(g, "pnp_conv2d_wgrad_bf16r")) return e;
    PNP_REQUIRE(xh && dyh && dw, "pnp_conv2d_wgrad_bf16r: null pointer");
    int bm, bn;
    int nsplit = wgrad_bf16r_plan(g, &bm, &bn);
    PNP_REQUIRE(nsplit >= 1, "pnp_conv2d_wgrad_bf16r: geometry not served (pnp_conv2d_bf16r_served(g, 2))");
    ConvArgs a = make_args((const float*)xh, (const float*)dyh, dw, g);
    a.dtype = PNP_DTYPE_BF16;
    a.x_bytes = (unsigned)((size_t)g->N * g->H * g->W * g->C * 2);
    a.w_bytes = (unsigned)((size_t)g->N * g->OH * g->OW * g->K * 2);      // a.w is dy [P][K]
    a.nblk_m = a.Kred / bm;
    a.nblk_n = a.K / bn;
    const size_t nout = (size_t)a.Kred * a.K;
    float* ws = (float*)workspace;
    if (!ws) workspace_bytes = 0;
    if (nsplit > 1 && workspace_bytes < nsplit * nout * sizeof(float)) {
        nsplit = (int)(workspace_bytes / (nout * sizeof(float)));
        if (nsplit < 2) nsplit = 1;
    }
    const int nchunks = pnp_cdiv(a.M, kWgradBf16rChunk);
    a.chunks_per_split = pnp_cdiv(nchunks, nsplit);
    nsplit = pnp_cdiv(nchunks, a.chunks_per_split);
    a.nsplit = nsplit;
    a.split_stride = (long long)nout;
    a.y = (nsplit > 1) ? ws : dw;
    a.accumulate = (nsplit > 1) ? 0 : accumulate;       // un-split: in the kernel's epilogue; split: in the reduce kernel
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(a.nblk_m * a.nblk_n * nsplit));
    PNP_REQUIRE(launch_wgrad_bf16r(a, wgrad_bf16r_tile(g), grid, st), "conv_wgrad_bf16r_kernel: no instance");
    PNP_CHECK_LAUNCH("conv_wgrad_bf16r_kernel");
    if (nsplit > 1) return launch_splitk_reduce(ws, dw, (size_t)nout, nsplit, (int)accumulate, st);
    return PNP_OK;
}

int pnp_sympad_fwd(const float* x, float* xp, int32_t N, int32_t H, int32_t W, int32_t C, int32_t p, void* stream) {
    PNP_REQUIRE(x && xp && N > 0 && H > 0 && W > 0 && C > 0 && p >= 0 && p <= H && p <= W, "pnp_sympad_fwd: bad argument");
    const int vec = (C % 4 == 0) ? 1 : 0;
    const size_t total = (size_t)N * (H + 2 * p) * (W + 2 * p) * (vec ? C / 4 : C);
    long long nb = (long long)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(sympad_fwd_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, x, xp, N, H, W, C, p, vec);
    PNP_CHECK_LAUNCH("sympad_fwd_kernel");
    return PNP_OK;
}

int pnp_sympad_bwd(const float* dxp, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t p, void* stream) {
    PNP_REQUIRE(dxp && dx && N > 0 && H > 0 && W > 0 && C > 0 && p >= 0 && p <= H && p <= W, "pnp_sympad_bwd: bad argument");
    const size_t total = (size_t)N * H * W * C;
    hipLaunchKernelGGL(sympad_bwd_kernel, dim3((unsigned)pnp_cdiv((long long)total, 256)), dim3(256), 0,
                       (hipStream_t)stream, dxp, dx, N, H, W, C, p);
    PNP_CHECK_LAUNCH("sympad_bwd_kernel");
    return PNP_OK;
}

}  // extern "C"
