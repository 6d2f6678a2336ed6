// conv_wgrad_dma.hip — fp32 filter gradient with LDS-DMA staging (round 4).
//
// dW[m' = (tap, c)][k] = sum_p x[p shifted by tap][c] * dy[p][k].  The fp32 MFMA (v_mfma_f32_32x32x2_f32) takes ONE float per lane and
// operand — A[i = lane & 31][k = lane >> 5] — so with the reduction index = pixel, a wave's A fragment is 32 consecutive channels of
// pixel p (lanes 0-31) and of pixel p + 1 (lanes 32-63): two contiguous 128-byte runs of x exactly as it lies in memory ([pixel][channel]),
// and the same for dy.  The LDS image of a stage is therefore the MEMORY image — rows = pixels, BM (BN) floats wide — and
// conv_wgrad_ring_kernel's global -> VGPR -> ds_write staging (the measured 9-10 % "load -> LDS-store dependency on cache-missing loads"
// of DESIGN.md §4.1, plus the register ring) is replaced by buffer_load_dwordx4 ... lds straight into it: no staging registers, no
// ds_write, NBUF stages in flight in LDS.  Fragment reads are ds_read_b32 of 32 consecutive dwords per half-wave: conflict-free with no
// padding or swizzle.  A stage is 32 pixels = 16 k-pairs x TM*TN MFMAs of 64 cycles: 4 096 matrix-pipe cycles per wave against ONE
// barrier and 8 DMA instructions.
// Served: zero padding, C % BM == 0 (a tile holds one tap), K % BN == 0, power-of-two output extents; everything else stays on
// conv_wgrad_ring_kernel / conv_wgrad_kernel.  Split planning, partial sums and "add into" are the caller's (conv_igemm.hip), unchanged.
#include "conv_common.h"

using namespace pnpconv;

namespace {

typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// (the builtin only exists in the device pass: in the host pass clang silently drops the kernel's launch stub over it)
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ void __launch_bounds__(NTHREADS, (NBUF * BK * (BM + BN) * 4 <= 80 * 1024 ? 2 : 1)) conv_wgrad_dma_kernel(ConvArgs a) {
    constexpr int NW = WM * WN;
    static_assert(NW == 4, "four waves");
    constexpr int ROWA = BM * 4, ROWB_ = BN * 4;                // bytes per LDS row (one pixel)
    constexpr int LPRA = ROWA / 16, LPRB = ROWB_ / 16;          // lanes per row
    static_assert(LPRA <= 64 && LPRB <= 64 && 64 % LPRA == 0 && 64 % LPRB == 0, "rows of at most 1 KiB");
    constexpr int RPIA = 64 / LPRA, RPIB = 64 / LPRB;           // rows per wave-instruction
    constexpr int NRA = BK / (NW * RPIA), NRB = BK / (NW * RPIB);
    static_assert(NRA >= 1 && NRB >= 1, "a stage must give every wave at least one instruction per operand");
    constexpr int LPS = NRA + NRB;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ASZ = BK * ROWA, STG = BK * (ROWA + ROWB_);
    static_assert(NBUF >= 2 && (NBUF - 2) * LPS < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(256))) unsigned char lds[NBUF * STG];

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
    const int nst = c_end - c_begin;

    const int tap = mm0 / a.C, c0 = mm0 - tap * a.C;             // the tile's tap (uniform) and first channel
    const int tr_ = tap / a.S, ts_ = tap - tr_ * a.S;
    const int dh = tr_ * a.dil - a.pad_t, dw = ts_ * a.dil - a.pad_l;
    const int lrowA = lane / LPRA, lchkA = lane % LPRA, lrowB = lane / LPRB, lchkB = lane % LPRB;
    int arow[NRA], brow[NRB];
    unsigned bvo[NRB];
#pragma unroll
    for (int i = 0; i < NRA; ++i) arow[i] = i * (NW * RPIA) + wave * RPIA + lrowA;
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        brow[i] = i * (NW * RPIB) + wave * RPIB + lrowB;
        bvo[i] = (unsigned)(((c_begin * BK + brow[i]) * a.K + n0) * 4 + lchkB * 16);
    }
    const int acol = c0 * 4 + lchkA * 16;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    int l_chunk = c_begin;
    auto issue = [&](int buf) {
        unsigned char* baseA = lds + buf * STG + wave * (RPIA * ROWA);
        unsigned char* baseB = lds + buf * STG + ASZ + wave * (RPIB * ROWB_);
        const int p0 = l_chunk * BK;
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int p = p0 + arow[i];
            const int n = p >> a.ohw_sh;
            const int oh = (p >> a.ow_sh) & (a.OH - 1), ow = p & (a.OW - 1);
            const int ih = oh * a.stride + dh, iw = ow * a.stride + dw;
            const bool ok = (p < P) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const unsigned vo = ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C) * 4 + acol) : OOB2;
            dma16(rx, (lds_void*)(baseA + i * (NW * RPIA * ROWA)), vo, 0);
        }
        const int sb = (l_chunk - c_begin) * (BK * a.K * 4);
        // (the hardware's range check covers the per-lane offset only, not the scalar one: rows past the last pixel are masked here)
#pragma unroll
        for (int i = 0; i < NRB; ++i) dma16(rw, (lds_void*)(baseB + i * (NW * RPIB * ROWB_)), (p0 + brow[i] < P) ? bvo[i] : OOB2, sb);
        l_chunk = min(l_chunk + 1, nchunks_total);
    };

    // fragments: lane (l31, h): A = x[pixel 2 kp + h][c = wm0 + tm*32 + l31], B = dy[pixel 2 kp + h][k = wn0 + tn*32 + l31]
    const int l31 = lane & 31, h = lane >> 5;
    const int foffA = h * ROWA + (wm0 + l31) * 4, foffB = h * ROWB_ + (wn0 + l31) * 4;
    Acc<TM, TN> acc;
    acc.zero();
    // Fragment pipeline: the operands of k-pair kp + 2 are read while k-pair kp is contracted (left to itself hipcc issues the two
    // ds_read2_b32 of a k-pair right in front of its four MFMAs, into the same registers, behind s_waitcnt lgkmcnt(0): one LDS round
    // trip exposed per 256 matrix-pipe cycles — the first version of this kernel ran at 124.6 TF/s, below the ring kernel's 126.9).
    // Three named register sets, indices compile-time; sched_group_barrier pins "2 LDS reads, then 4 MFMAs" per k-pair.
    auto compute = [&](int buf) {
        const unsigned char* A = lds + buf * STG + foffA;
        const unsigned char* B = lds + buf * STG + ASZ + foffB;
        float af[3][TM], bfr[3][TN];
        auto rd = [&](int s_, int kp) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[s_][tm] = *reinterpret_cast<const float*>(A + (2 * kp) * ROWA + tm * 128);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bfr[s_][tn] = *reinterpret_cast<const float*>(B + (2 * kp) * ROWB_ + tn * 128);
        };
        rd(0, 0);
        rd(1, 1);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * ((TM + TN + 1) / 2), 0);    // the two k-pairs read ahead
#pragma unroll
        for (int kp = 0; kp < BK / 2; ++kp) {
            if (kp + 2 < BK / 2) rd((kp + 2) % 3, kp + 2);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kp % 3][tm], bfr[kp % 3][tn], acc.v[tm][tn], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, (TM + TN + 1) / 2, 0);      // DS reads (hipcc pairs them into ds_read2_b32)
            __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                // MFMAs
        }
    };
    auto stage = [&](int d) {
        wait_vm<(NBUF - 2) * LPS>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        issue((d + NBUF - 1) % NBUF);
        compute(d);
    };
    if (nst > 0) {
#pragma unroll
        for (int d = 0; d < NBUF - 1; ++d) issue(d);
        const int nmain = (nst / NBUF) * NBUF;
        for (int j0 = 0; j0 < nmain; j0 += NBUF) {
#pragma unroll
            for (int d = 0; d < NBUF; ++d) stage(d);
        }
#pragma unroll
        for (int d = 0; d < NBUF - 1; ++d)
            if (nst - nmain > d) stage(d);
        wait_vm<0>();
    }
    wgrad_epilogue<TM, TN>(a, acc, a.y + (size_t)z * a.split_stride, mm0, n0, wm0, wn0, lane);
}

}  // namespace

namespace pnpconv {

bool wgrad_dma_ok(const ConvArgs& a, int bm, int bn) {
    static const int off = getenv("PNP_WGRAD_DMA") ? (atoi(getenv("PNP_WGRAD_DMA")) == 0) : 0;
    if (off || a.pad_mode != PNP_PAD_ZERO || a.dtype != PNP_DTYPE_F32) return false;
    if (bm != 128 || (bn != 128 && bn != 64) || a.C % bm != 0 || a.K % bn != 0) return false;     // (the planner's tiles are 128 rows)
    if (a.ow_sh < 0 || a.ohw_sh < 0) return false;
    return a.x_bytes < 0x80000000u && a.w_bytes < 0x80000000u;
}

bool launch_wgrad_dma(const ConvArgs& a, int bm, int bn, dim3 grid, hipStream_t st) {
#define PNP_WG(BM_, BN_, NBUF_)                                                                                                   \
    if (bm == BM_ && bn == BN_) {                                                                                                 \
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), conv_bytes(a), "conv_wgrad_dma_kernel<%d, %d, 2, 2, %d>", BM_, BN_, NBUF_); \
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM_, BN_, 2, 2, NBUF_>), grid, dim3(NTHREADS), 0, st, a);                        \
        return true;                                                                                                              \
    }
    PNP_WG(128, 128, 2)
    PNP_WG(128, 64, 3)
#undef PNP_WG
    return false;
}

}  // namespace pnpconv
