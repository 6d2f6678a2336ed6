// conv_bf16r.hip — bf16-RESIDENT MFMA convolutions (BASELINE.json configs[4]; round 4).
//
// conv_bf16.hip (round 2/3) rounds fp32 tensors to bf16 WHILE a tile is staged: every operand crosses HBM -> L2 -> L1 -> VGPR as fp32, is
// packed by the vector ALU and written to LDS by ds_write — measured 0.17 of the bf16 matrix peak, with exactly the fp32 kernel's bytes
// (profiles/r03_bf16_pmc_counters.json).  Here the operands ARE bf16 in HBM:
//   * activations / upstream gradients: a bf16 copy [N][H][W][C] written by the PRODUCING kernel (the conv / BN epilogues; pnp_cast_bf16
//     where a producer has no such output yet),
//   * filters: two bf16 shadows of the fp32 master filter [R][S][C][K] (pnp_filter_bf16): `w_oi` = [tap][K][C] (reduction index C
//     contiguous: forward) and `w_io` = [tap][C][K] (reduction index K contiguous: data gradient, taps walked in reverse — no
//     flip/transpose launch),
// so that BOTH MFMA operands of a stage are rows of BKC contiguous reduction elements and go global -> LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR staging, no v_cvt, no ds_write).  The DMA writes lane-linear (wave-uniform base + 16 B x lane), so
// the LDS image is [row][BKC] with rows of 128 B (BKC = 64) or 64 B (BKC = 32); the bank swizzle (16-byte chunk c of row r lives at
// chunk c ^ swz(r)) is applied on the SOURCE side (the lane that fills LDS chunk c' fetches global chunk c' ^ swz(r): the same cache
// line) and again on the fragment reads: every ds_read_b128 lane group of 16 then covers 16 distinct 16-byte slots of the 256-byte bank
// row.  Out-of-range rows (TF zero padding, ragged tile edges) carry a voffset past the buffer: the hardware returns zeros, also into LDS.
//   tiles: 256 x 128 / 8 waves (64 x 64 per wave), 128 x 128 and 128 x 64 / 4 waves; NBUF LDS stages, ONE raw s_barrier per stage,
//   loads of stage j + NBUF - 1 in flight under the MFMAs of stage j (counted s_waitcnt vmcnt, never a drain inside the loop).
// fp32 accumulation, fp32 output (+ optional bf16 copy of the output from the same epilogue: the next convolution's operand).
#include "conv_common.h"

using namespace pnpconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// one LDS-DMA piece: 16 bytes per lane, global (buffer descriptor + per-lane voffset + scalar soffset) -> LDS (wave-uniform dst + 16 x lane).
// The builtin only exists in the device pass (in the host pass clang silently drops the whole kernel's launch stub over it).
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// 16-byte chunk swizzle of an LDS row (see the file header)
template <int BKC>
__device__ __forceinline__ int swz(int row) {
    return BKC == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}

template <int BM, int BN, int WM, int WN, int BKC, int KIND, int R, int S, int NBUF, int ILV = 1>
__global__ void __launch_bounds__(64 * WM * WN, (WM * WN == 8 ? 2 : (NBUF * (BM + BN) * BKC * 2 <= 80 * 1024 ? 2 : 1)))
    conv_bf16r_kernel(ConvArgs a) {
    constexpr int NTAP = R * S;
    static_assert(NTAP <= 32, "one validity bit per tap");
    static_assert(BKC == 64 || BKC == 32, "rows of 128 or 64 bytes");
    constexpr int NW = WM * WN;
    constexpr int ROWB = BKC * 2;              // bytes per LDS row
    constexpr int LPR = ROWB / 16;             // lanes per row
    constexpr int RPI = 64 / LPR;              // rows per wave-instruction (1 KiB)
    constexpr int RPP = NW * RPI;              // rows per pass of the whole workgroup
    constexpr int NRA = BM / RPP, NRB = BN / RPP;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must split into whole passes");
    constexpr int LPS = NRA + NRB;             // LDS-DMA instructions per wave per stage
    constexpr int KS = BKC / 16;               // 16-deep MFMA slices per stage
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int STG = (BM + BN) * ROWB;      // bytes per LDS stage
    static_assert(NBUF >= 2 && (NBUF - 2) * LPS < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(256))) unsigned char lds[NBUF * STG];      // ONE shared object (a second one de-pipelines the DMA)

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = a.nblk_m * a.nblk_n;
    int bid = blockIdx.x;
    if (a.xcd_swizzle) bid = xcd_remap(bid, nblk);
    int mt, nt;
    tile_coords(bid, a.nblk_m, a.nblk_n, a.gn, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    // ---- loader rows: A = output pixels (byte offset of the pixel shifted by -pad, one validity bit per tap), B = filters
    const int lrow = lane / LPR, lchk = lane % LPR;
    int abase[NRA];
    unsigned amask[NRA];
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
        const int r = i * RPP + wave * RPI + lrow;
        int m = m0 + r;
        const bool ok = m < a.M;
        if (!ok) m = 0;
        int n, oh, ow;
        split_row(a, m, n, oh, ow);
        const int vh0 = oh * a.stride - a.pad_t, vw0 = ow * a.stride - a.pad_l;
        abase[i] = (((n * a.H + vh0) * a.W + vw0) * a.C) * 2 + ((lchk ^ swz<BKC>(r)) << 4);
        unsigned mk = 0;
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int ih = vh0 + (tap / S) * a.dil, iw = vw0 + (tap % S) * a.dil;
            const bool v = ok & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            mk |= (v ? 1u : 0u) << tap;
        }
        amask[i] = mk;
    }
    unsigned bvo[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        const int r = i * RPP + wave * RPI + lrow;
        const int n = n0 + r;
        bvo[i] = (n < a.K) ? (unsigned)((n * a.C) * 2 + ((lchk ^ swz<BKC>(r)) << 4)) : OOB2;
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);

    const int ncc = a.C / BKC;
    const int nst = ncc * NTAP;
    int l_cc = 0, l_tap = 0;            // (channel group, tap) of the next stage to fetch: uniform
    const int tapKC = a.K * a.C * 2;     // bytes per tap of the filter shadow
    // which tap of the filter SHADOW stage tap (tr, ts) reads: forward the same one; data gradient the filter walked in reverse
    // (correlation -> convolution); a stride phase its sub-filter's tap inside the full filter (ConvArgs::ph_*)
    auto tap_eff = [&](int tap, int tr, int ts) {
        if (KIND == 1 && a.ph_st) return (a.ph_pa + a.ph_st * (R - 1 - tr)) * a.ph_S + a.ph_pb + a.ph_st * (S - 1 - ts);
        return (KIND == 1) ? (NTAP - 1 - tap) : tap;
    };
    auto issue = [&](int buf) {
        unsigned char* base = lds + buf * STG + wave * (RPI * ROWB);
        const int tr = l_tap / S, ts = l_tap - tr * S;
        const int tshift = ((tr * a.dil * a.W + ts * a.dil) * a.C) * 2;
        const int sa = l_cc * (BKC * 2);
        const int sb = tap_eff(l_tap, tr, ts) * tapKC + l_cc * (BKC * 2);
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const unsigned vo = ((amask[i] >> l_tap) & 1u) ? (unsigned)(abase[i] + tshift) : OOB2;
            dma16(rx, (lds_void*)(base + i * (RPP * ROWB)), vo, sa);
        }
#pragma unroll
        for (int i = 0; i < NRB; ++i)
            dma16(rw, (lds_void*)(base + BM * ROWB + i * (RPP * ROWB)), bvo[i], sb);
        // fetches past the last stage re-read the last one (valid addresses, into a buffer nobody reads any more): the stage body has
        // no conditional load and the vmcnt arithmetic stays uniform
        const int wrap = (l_tap + 1 == NTAP) ? 1 : 0;
        l_tap = wrap ? 0 : l_tap + 1;
        l_cc = min(l_cc + wrap, ncc - 1);
    };

    // ---- fragment reads: lane (l31, h) reads row l31 of a 32-row block, 16-byte chunk (2 ks + h) ^ swz(row); the swizzle of a row
    // depends on its low 5 bits only (blocks start at multiples of 32), so ONE set of per-lane offsets serves A and B
    const int l31 = lane & 31, h = lane >> 5;
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = l31 * ROWB + (((2 * ks + h) ^ swz<BKC>(l31)) << 4);

    Acc<TM, TN> acc;
    acc.zero();
    auto compute = [&](int buf) {
        const unsigned char* A = lds + buf * STG + wm0 * ROWB;
        const unsigned char* B = lds + buf * STG + BM * ROWB + wn0 * ROWB;
        bf16x8 af[KS][TM], bfr[KS][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[ks][tm] = *reinterpret_cast<const bf16x8*>(A + tm * (32 * ROWB) + foff[ks]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bfr[ks][tn] = *reinterpret_cast<const bf16x8*>(B + tn * (32 * ROWB) + foff[ks]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][tm], bfr[ks][tn], acc.v[tm][tn], 0, 0, 0);
    };
    // stage j: its DMA was issued NBUF - 1 stages ago.  Own loads landed (counted vmcnt: the NBUF - 2 younger stages stay in flight), own
    // fragment reads of stage j - 1 retired (lgkmcnt), barrier: now EVERY wave's part of stage j is in LDS and nobody reads buffer
    // (j - 1) % NBUF any more — refill it with stage j + NBUF - 1, then contract stage j.
    // ILV = 1 (the default; ILV = 0 is the first version, kept for A/B builds): the DMA pieces of the stage being fetched are issued
    // BETWEEN the MFMAs of the stage being contracted (one piece costs the wave 60-150 issue cycles — MI355X_MICROARCH.md — and with all
    // pieces in front of the MFMAs both waves of a SIMD, released by the same barrier, issue pieces at the same time and leave the matrix
    // pipe idle), and the fragments of slice ks + 1 are read under the MFMAs of slice ks.  sched_group_barrier pins the order; the
    // address arithmetic of the pieces floats.  Within-run A/B x2 at B = 16: 512->512 998 -> 1063-1073 TF/s, g10 data gradient 691 -> 787,
    // cls2 128->128@128^2 795 -> 852, cls3 256->256@64^2 930 -> 998.
    auto stage_ilv = [&](int d) {
        wait_vm<(NBUF - 2) * LPS>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        const int nb = (d + NBUF - 1) % NBUF;
        unsigned char* base = lds + nb * STG + wave * (RPI * ROWB);
        const int tr = l_tap / S, ts = l_tap - tr * S;
        const int tshift = ((tr * a.dil * a.W + ts * a.dil) * a.C) * 2;
        const int sa = l_cc * (BKC * 2);
        const int sb = tap_eff(l_tap, tr, ts) * tapKC + l_cc * (BKC * 2);
        auto piece = [&](int i) {
            if (i < NRA) {
                const unsigned vo = ((amask[i] >> l_tap) & 1u) ? (unsigned)(abase[i] + tshift) : OOB2;
                dma16(rx, (lds_void*)(base + i * (RPP * ROWB)), vo, sa);
            } else {
                dma16(rw, (lds_void*)(base + BM * ROWB + (i - NRA) * (RPP * ROWB)), bvo[i - NRA], sb);
            }
        };
        const unsigned char* A = lds + d * STG + wm0 * ROWB;
        const unsigned char* B = lds + d * STG + BM * ROWB + wn0 * ROWB;
        bf16x8 af[2][TM], bfr[2][TN];
        auto rd = [&](int s_, int ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) af[s_][tm] = *reinterpret_cast<const bf16x8*>(A + tm * (32 * ROWB) + foff[ks]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bfr[s_][tn] = *reinterpret_cast<const bf16x8*>(B + tn * (32 * ROWB) + foff[ks]);
        };
        rd(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) rd((ks + 1) & 1, ks + 1);
            const int p0 = (ks * LPS) / KS, p1 = ((ks + 1) * LPS) / KS;        // pieces issued under this slice
            int mi = 0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks & 1][tm], bfr[ks & 1][tn], acc.v[tm][tn], 0, 0, 0);
                    const int pi = p0 + mi;
                    if (pi < p1) piece(pi);
                    ++mi;
                }
            if (ks + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int m = 0; m < TM * TN; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (p0 + m < p1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        }
        const int wrap = (l_tap + 1 == NTAP) ? 1 : 0;
        l_tap = wrap ? 0 : l_tap + 1;
        l_cc = min(l_cc + wrap, ncc - 1);
    };
    auto stage = [&](int d) {
        if constexpr (ILV != 0) {
            stage_ilv(d);
        } else {
            wait_vm<(NBUF - 2) * LPS>();
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            issue((d + NBUF - 1) % NBUF);
            compute(d);
        }
    };
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
    conv_epilogue<TM, TN>(a, acc, a.y, m0, n0, wm0, wn0, lane, mt * WM + wave / WN, true);
}

// ===================================== filter gradient ============================================
// dW[m' = (tap, c)][k] = sum_p x[p shifted by tap][c] * dy[p][k]: the reduction index is the PIXEL, and both operands are pixel-major in
// memory ([p][c], [p][k]).  They are staged exactly as they lie — LDS rows = pixels, BM (BN) channels wide, by the same LDS-DMA — and
// the transposition the MFMA wants (8 consecutive reduction elements per lane) is done by the LDS read itself: ds_read_b64_tr_b16 hands
// lane i of a 16-lane group the i-th COLUMN of the 4 x 16 block whose rows the lanes 4e .. 4e+3 point at (profiles/r04_micro_lds_dma_tr16.txt),
// i.e. 4 consecutive pixels of one channel; two such reads are one MFMA operand.  Bank swizzle: the four rows of a block are 256 (128)
// bytes apart, so the 16-byte chunk index is XOR-ed with 4 * (row & 3) (4 * ((row >> 1) & 1) for 128-byte rows): the 4 rows x 4 chunks
// a half-wave reads then cover the 256-byte bank row exactly once.
// a.x = x (bf16), a.w = dy (bf16, [P][K]); reduction split over workgroups in chunks of 64 pixels (grid.x = tiles * nsplit; partials
// summed by conv_igemm.hip's splitk_reduce_kernel).  A tile holds ONE tap (C % BM == 0), OW and OH*OW are powers of two.
typedef short s16x4 __attribute__((ext_vector_type(4)));
// (inline asm, not __builtin_amdgcn_ds_read_tr16_b64_v4i16: behind an LDS-DMA in flight hipcc fences the builtin with s_waitcnt vmcnt(0))
__device__ __forceinline__ s16x4 lds_tr16_asm(unsigned lds_byte_addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte_addr));
    return v;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const unsigned char* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) unsigned char*)p;
}
template <int CPR>
__device__ __forceinline__ int swz_px(int row) {
    return CPR == 16 ? ((row & 3) << 2) : (((row >> 1) & 1) << 2);
}

template <int BM, int BN, int WM, int WN, int NBUF>
__global__ void __launch_bounds__(256, (NBUF * (BM + BN) * 128 <= 80 * 1024 ? 2 : 1)) conv_wgrad_bf16r_kernel(ConvArgs a) {
    constexpr int BKP = 64;                      // pixels per stage
    constexpr int NW = WM * WN;
    static_assert(NW == 4, "four waves");
    constexpr int ROWA = BM * 2, ROWB_ = BN * 2;                 // bytes per LDS row
    constexpr int CPRA = ROWA / 16, CPRB = ROWB_ / 16;           // 16-byte chunks (= lanes) per row
    static_assert((CPRA == 16 || CPRA == 8) && (CPRB == 16 || CPRB == 8), "rows of 256 or 128 bytes");
    constexpr int RPIA = 64 / CPRA, RPIB = 64 / CPRB;            // rows per wave-instruction
    constexpr int NRA = BKP / (NW * RPIA), NRB = BKP / (NW * RPIB);
    constexpr int LPS = NRA + NRB;
    constexpr int KS = BKP / 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int ASZ = BKP * ROWA, STG = BKP * (ROWA + ROWB_);
    static_assert(NBUF >= 2 && (NBUF - 2) * LPS < 64, "vmcnt is a 6-bit counter");
    static_assert(2 * (TM + TN) < 16, "lgkmcnt is a 4-bit counter");
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
    const int nchunks_total = (P + BKP - 1) / BKP;
    const int c_begin = z * a.chunks_per_split;
    int c_end = c_begin + a.chunks_per_split;
    if (c_end > nchunks_total) c_end = nchunks_total;
    const int nst = c_end - c_begin;

    // the tile's tap (uniform) and first channel
    const int tap = mm0 / a.C, c0 = mm0 - tap * a.C;
    const int tr_ = tap / a.S, ts_ = tap - tr_ * a.S;
    const int dh = tr_ * a.dil - a.pad_t, dw = ts_ * a.dil - a.pad_l;
    // loader rows: lane (lrow, lchk) of pass i fills LDS row i * NW * RPI + wave * RPI + lrow, chunk lchk, with source chunk lchk ^ swz
    const int lrowA = lane / CPRA, lchkA = lane % CPRA, lrowB = lane / CPRB, lchkB = lane % CPRB;
    int arow[NRA], acol[NRA];
#pragma unroll
    for (int i = 0; i < NRA; ++i) {
        arow[i] = i * (NW * RPIA) + wave * RPIA + lrowA;
        acol[i] = (c0 * 2) + ((lchkA ^ swz_px<CPRA>(arow[i])) << 4);
    }
    unsigned bvo[NRB];           // dy rows: [p][K]
    int brow[NRB];
#pragma unroll
    for (int i = 0; i < NRB; ++i) {
        brow[i] = i * (NW * RPIB) + wave * RPIB + lrowB;
        bvo[i] = (unsigned)(((c_begin * BKP + brow[i]) * a.K + n0) * 2 + ((lchkB ^ swz_px<CPRB>(brow[i])) << 4));
    }
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(a.w, a.w_bytes);
    int l_chunk = c_begin;
    auto issue = [&](int buf) {
        unsigned char* baseA = lds + buf * STG + wave * (RPIA * ROWA);
        unsigned char* baseB = lds + buf * STG + ASZ + wave * (RPIB * ROWB_);
        const int p0 = l_chunk * BKP;
#pragma unroll
        for (int i = 0; i < NRA; ++i) {
            const int p = p0 + arow[i];
            const int n = p >> a.ohw_sh;
            const int oh = (p >> a.ow_sh) & (a.OH - 1), ow = p & (a.OW - 1);
            const int ih = oh * a.stride + dh, iw = ow * a.stride + dw;
            const bool ok = (p < P) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
            const unsigned vo = ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C) * 2 + acol[i]) : OOB2;
            dma16(rx, (lds_void*)(baseA + i * (NW * RPIA * ROWA)), vo, 0);
        }
        const int sb = (l_chunk - c_begin) * (BKP * a.K * 2);
        // (the hardware's range check covers the per-lane offset only, not the scalar one: rows past the last pixel are masked here)
#pragma unroll
        for (int i = 0; i < NRB; ++i) dma16(rw, (lds_void*)(baseB + i * (NW * RPIB * ROWB_)), (p0 + brow[i] < P) ? bvo[i] : OOB2, sb);
        l_chunk = min(l_chunk + 1, nchunks_total);      // past the end: every row is masked
    };

    // fragment reads (see the header of this section): lane = (group g, e, q); rows pbase + 4 r + e of a 16-pixel slice, 4 channels
    const int g16 = lane >> 4, e4 = (lane >> 2) & 3, q4 = lane & 3;
    int foffA[TM], foffB[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int chunk = ((wm0 + tm * 32) >> 3) + (g16 & 1) * 2 + (q4 >> 1);
        foffA[tm] = ((g16 >> 1) * 8 + e4) * ROWA + ((chunk ^ swz_px<CPRA>(e4)) << 4) + (q4 & 1) * 8;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int chunk = ((wn0 + tn * 32) >> 3) + (g16 & 1) * 2 + (q4 >> 1);
        foffB[tn] = ((g16 >> 1) * 8 + e4) * ROWB_ + ((chunk ^ swz_px<CPRB>(e4)) << 4) + (q4 & 1) * 8;
    }
    Acc<TM, TN> acc;
    acc.zero();
    // the DMA pieces of the stage being fetched go BETWEEN the MFMAs of the stage being contracted, fragments one slice ahead (see
    // conv_bf16r_kernel's stage_ilv)
    auto stage = [&](int d) {
        wait_vm<(NBUF - 2) * LPS>();
        wait_lgkm0();
        __builtin_amdgcn_s_barrier();
        const int nb = (d + NBUF - 1) % NBUF;
        unsigned char* baseA = lds + nb * STG + wave * (RPIA * ROWA);
        unsigned char* baseB = lds + nb * STG + ASZ + wave * (RPIB * ROWB_);
        const int p0_ = l_chunk * BKP;
        const int sb = (l_chunk - c_begin) * (BKP * a.K * 2);
        auto piece = [&](int i) {
            if (i < NRA) {
                const int p = p0_ + arow[i];
                const int n = p >> a.ohw_sh;
                const int oh = (p >> a.ow_sh) & (a.OH - 1), ow = p & (a.OW - 1);
                const int ih = oh * a.stride + dh, iw = ow * a.stride + dw;
                const bool ok = (p < P) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                const unsigned vo = ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * a.C) * 2 + acol[i]) : OOB2;
                dma16(rx, (lds_void*)(baseA + i * (NW * RPIA * ROWA)), vo, 0);
            } else {
                const int j = i - NRA;
                dma16(rw, (lds_void*)(baseB + j * (NW * RPIB * ROWB_)), (p0_ + brow[j] < P) ? bvo[j] : OOB2, sb);
            }
        };
        // Fragment reads are INLINE ASM: behind an LDS-DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in front of the ds_read_tr builtin
        // (it does not for plain LDS loads) — the first version of this kernel therefore waited for the stage it had just requested
        // before contracting the current one.  An asm read is invisible to the waitcnt pass, so the waits are counted by hand: the
        // 2 (TM + TN) reads of slice ks + 1 may be outstanding when slice ks is contracted (LDS returns in order), and a full scheduling
        // barrier behind each wait keeps the MFMAs from being hoisted above it.
        const unsigned abase_ = lds_addr(lds) + d * STG, bbase_ = lds_addr(lds) + d * STG + ASZ;
        s16x4 alo[2][TM], ahi[2][TM], blo[2][TN], bhi[2][TN];
        auto rd = [&](int s_, int ks) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                alo[s_][tm] = lds_tr16_asm(abase_ + foffA[tm] + (ks * 16) * ROWA);
                ahi[s_][tm] = lds_tr16_asm(abase_ + foffA[tm] + (ks * 16 + 4) * ROWA);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                blo[s_][tn] = lds_tr16_asm(bbase_ + foffB[tn] + (ks * 16) * ROWB_);
                bhi[s_][tn] = lds_tr16_asm(bbase_ + foffB[tn] + (ks * 16 + 4) * ROWB_);
            }
        };
        rd(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 1 < KS) {
                rd((ks + 1) & 1, ks + 1);
                wait_lgkm<2 * (TM + TN)>();
            } else {
                wait_lgkm<0>();
            }
            __builtin_amdgcn_sched_barrier(0);
            const int q0 = (ks * LPS) / KS, q1 = ((ks + 1) * LPS) / KS;        // pieces issued under this slice
            int mi = 0;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const bf16x8 av = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo[ks & 1][tm], ahi[ks & 1][tm], 0, 1, 2, 3, 4, 5, 6, 7));
                    const bf16x8 bv = __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[ks & 1][tn], bhi[ks & 1][tn], 0, 1, 2, 3, 4, 5, 6, 7));
                    acc.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc.v[tm][tn], 0, 0, 0);
                    // piece j of this slice's n goes behind MFMA floor(j * m / n) of its m (8 pieces per 16 MFMAs on the 128 x 128 tile)
#pragma unroll
                    for (int j = 0; j < q1 - q0; ++j)
                        if ((j * TM * TN) / (q1 - q0) == mi) piece(q0 + j);
                    ++mi;
                }
#pragma unroll
            for (int m = 0; m < TM * TN; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
#pragma unroll
                for (int j = 0; j < q1 - q0; ++j)
                    if ((j * TM * TN) / (q1 - q0) == m) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        l_chunk = min(l_chunk + 1, nchunks_total);
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

// ---------------------------------------- casts -------------------------------------------------
__global__ void cast_bf16_kernel(const float* __restrict__ x, __bf16* __restrict__ y, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(x + i), b = *reinterpret_cast<const f32x4*>(x + i + 4);
        bf16x8 o;
        o[0] = (__bf16)a[0]; o[1] = (__bf16)a[1]; o[2] = (__bf16)a[2]; o[3] = (__bf16)a[3];
        o[4] = (__bf16)b[0]; o[5] = (__bf16)b[1]; o[6] = (__bf16)b[2]; o[7] = (__bf16)b[3];
        *reinterpret_cast<bf16x8*>(y + i) = o;
    } else {
        for (size_t j = i; j < n; ++j) y[j] = (__bf16)x[j];
    }
}

// fp32 master filter [tap][C][K] -> bf16 shadows  w_io = [tap][C][K] (plain cast)  and  w_oi = [tap][K][C] (32 x 32 tiles through LDS)
__global__ void __launch_bounds__(256) filter_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ w_io, __bf16* __restrict__ w_oi,
                                                          int C, int K) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, c0 = blockIdx.y * 32, k0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const size_t tb = (size_t)tap * C * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, k = k0 + tx;
        float v = 0.f;
        if (c < C && k < K) {
            v = w[tb + (size_t)c * K + k];
            if (w_io) w_io[tb + (size_t)c * K + k] = (__bf16)v;
        }
        tile[ty + 8 * i][tx] = v;
    }
    __syncthreads();
    if (w_oi) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = k0 + ty + 8 * i, c = c0 + tx;
            if (c < C && k < K) w_oi[tb + (size_t)k * C + c] = (__bf16)tile[tx][ty + 8 * i];
        }
    }
}

// ---------------------------------------- host side ----------------------------------------------
constexpr bool r_shape(int kind, int R, int S) { return (R == 3 && S == 3) || (R == 5 && S == 5 && kind == 0); }
bool strided_dgrad_served(const pnp_conv_geom* g);

// tile plan: 0 = 256x128 (8 waves, 3 LDS stages), 1 = 128x128 (4 waves, 2 stages), 2 = 128x64 (4 waves, 3 stages); -1: not served
int plan_tile(long long M, int K) {
    static const int force = getenv("PNP_BF16R_TILE") ? atoi(getenv("PNP_BF16R_TILE")) : -1;
    if (K % 64 != 0) return -1;
    // 256 x 128 needs one tile per CU (one workgroup of 8 waves fits), 128 x 128 two per CU, else 128 x 64 — measured with the
    // interleaved schedule (round 4 tile sweep): 256->256@32^2 630 (128x128, one tile per CU) vs 663 TF/s (128x64), 256->512 data
    // gradient 674 vs 772, 512->512@16^2 388 / 363 vs 662; whole bf16 joint step, same box x2: 455.5 -> 460.8 slices/s
    int tile;
    if (K % 128 == 0 && pnp_cdiv(M, 256) * (K / 128) >= 256) tile = 0;
    else if (K % 128 == 0 && pnp_cdiv(M, 128) * (K / 128) >= 512) tile = 1;
    else tile = 2;
    if (force >= 0 && !(force <= 1 && K % 128 != 0)) tile = force;
    return tile;
}

template <int BM, int BN, int WM, int WN, int BKC, int KIND, int NBUF>
int launch_tile(ConvArgs& a, hipStream_t st) {
    a.nblk_m = pnp_cdiv(a.M, BM);
    a.nblk_n = pnp_cdiv(a.K, BN);
    a.nsplit = 1;
    dim3 grid((unsigned)(a.nblk_m * a.nblk_n));
#define PNP_R(RR, SS)                                                                                                            \
    if (a.R == RR && a.S == SS) {                                                                                                \
        PnpProfScope ps(prof_class(KIND), st, conv_flops(a), 0.5 * conv_bytes(a) + 2.0 * (double)a.M * a.K,                      \
                        "conv_bf16r_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %d, 1>", BM, BN, WM, WN, BKC, KIND, RR, SS, NBUF);     \
        hipLaunchKernelGGL((conv_bf16r_kernel<BM, BN, WM, WN, BKC, KIND, RR, SS, NBUF>), grid, dim3(64 * WM * WN), 0, st, a);    \
        PNP_CHECK_LAUNCH("conv_bf16r_kernel");                                                                                   \
        return PNP_OK;                                                                                                           \
    }
    PNP_R(3, 3)
    if constexpr (KIND == 0) { PNP_R(5, 5) }
    if constexpr (KIND == 1 && BKC == 64) {       // stride-phase sub-filters of the k3 / k5 strided layers
        PNP_R(1, 1) PNP_R(1, 2) PNP_R(2, 1) PNP_R(2, 2) PNP_R(2, 3) PNP_R(3, 2)
    }
#undef PNP_R
    pnp_set_error("conv_bf16r_kernel: no instance for %dx%d", a.R, a.S);
    return PNP_EINVAL;
}

template <int KIND>
int launch_kind(ConvArgs& a, hipStream_t st) {
    const int tile = plan_tile(a.M, a.K);
    PNP_REQUIRE(tile >= 0, "conv_bf16r: geometry not served");
    if (a.C % 64 == 0) {
        if (tile == 0) return launch_tile<256, 128, 4, 2, 64, KIND, 3>(a, st);
        if (tile == 1) return launch_tile<128, 128, 2, 2, 64, KIND, 2>(a, st);
        return launch_tile<128, 64, 2, 2, 64, KIND, 3>(a, st);
    }
    if (tile == 0) return launch_tile<256, 128, 4, 2, 32, KIND, 3>(a, st);
    if (tile == 1) return launch_tile<128, 128, 2, 2, 32, KIND, 2>(a, st);
    return launch_tile<128, 64, 2, 2, 32, KIND, 3>(a, st);
}

// strided data gradient: one launch per stride phase (conv_igemm.hip's plan_phases), every phase a stride-1 resident convolution over dy
// with its sub-filter taken out of the full filter shadow.  Served when the phases exist, their sub-filters are instantiated shapes,
// K (the reduction channels) comes in whole 64-groups and every non-empty phase has >= 4096 pixels.
bool strided_dgrad_served(const pnp_conv_geom* g) {
    if (g->K % 64 != 0 || g->C % 64 != 0) return false;
    DgradPhase ph[16];
    const int nph = plan_phases(g, ph);
    if (nph <= 0) return false;
    for (int i = 0; i < nph; ++i) {
        if (ph[i].I == 0 || ph[i].J == 0) continue;
        if (ph[i].T > 3 || ph[i].U > 3 || (ph[i].T == 3 && ph[i].U == 1) || (ph[i].T == 1 && ph[i].U == 3)) return false;
        if ((long long)g->N * ph[i].I * ph[i].J < 4096) return false;
    }
    const long long xin = (long long)g->N * g->H * g->W * g->C, yout = (long long)g->N * g->OH * g->OW * g->K;
    return xin < (1ll << 30) && yout < (1ll << 30);
}

// what the resident kernels serve: zero padding, instantiated filter shapes, reduction channels in whole 32-groups, output channels in
// whole 64-groups, tensors < 2 GiB; data gradient: stride 1 only (the strided ones stay on the stride-phase kernels)
bool served(const pnp_conv_geom* g, int kind) {
    if (!g || g->pad_mode != PNP_PAD_ZERO) return false;
    static const int off = getenv("PNP_BF16R_OFF") ? 1 : 0;
    if (off) return false;
    if (kind == 1 && g->stride != 1) return strided_dgrad_served(g);
    const int red = kind == 1 ? g->K : g->C, outc = kind == 1 ? g->C : g->K;
    if (!r_shape(kind, g->R, g->S) || red % 32 != 0 || outc % 64 != 0) return false;
    if (kind == 1 && (g->dil * (g->R - 1) < g->pad_t || g->dil * (g->S - 1) < g->pad_l)) return false;
    const long long xin = (long long)g->N * g->H * g->W * g->C, yout = (long long)g->N * g->OH * g->OW * g->K;
    if (xin >= (1ll << 30) || yout >= (1ll << 30)) return false;
    const long long M = kind == 1 ? (long long)g->N * g->H * g->W : (long long)g->N * g->OH * g->OW;
    if (M < 4096) return false;                      // a handful of tiles: the reduction-split fp32-operand paths serve those
    return plan_tile(M, outc) >= 0;
}

ConvArgs base_args(const void* x, const void* w, float* y, const pnp_conv_geom* g) {
    ConvArgs a{};
    a.x = (const float*)x; a.w = (const float*)w; a.y = y;
    a.N = g->N; a.H = g->H; a.W = g->W; a.C = g->C; a.K = g->K; a.R = g->R; a.S = g->S;
    a.OH = g->OH; a.OW = g->OW; a.stride = g->stride; a.dil = g->dil; a.pad_t = g->pad_t; a.pad_l = g->pad_l;
    a.pad_mode = g->pad_mode;
    a.dtype = PNP_DTYPE_BF16;
    a.ups = 1;
    a.M = g->N * g->OH * g->OW;
    a.Kred = g->R * g->S * g->C;
    a.OHW = g->OH * g->OW;
    a.nsplit = 1;
    const bool p2 = (a.OW & (a.OW - 1)) == 0 && (a.OHW & (a.OHW - 1)) == 0;
    a.ow_sh = p2 ? __builtin_ctz((unsigned)a.OW) : -1;
    a.ohw_sh = p2 ? __builtin_ctz((unsigned)a.OHW) : -1;
    a.drop_keep = 1.f;
    static const int env_noswz = getenv("PNP_CONV_NOSWIZZLE") ? 1 : 0;
    a.xcd_swizzle = env_noswz ? 0 : 1;
    a.x_bytes = (unsigned)((size_t)g->N * g->H * g->W * g->C * 2);
    a.w_bytes = (unsigned)((size_t)g->R * g->S * g->C * g->K * 2);
    static const int env_gn = getenv("PNP_CONV_GN") ? atoi(getenv("PNP_CONV_GN")) : 4;
    a.gn = env_gn;
    return a;
}

int wm_of_tile(int tile) { return tile == 0 ? 4 : 2; }
int bm_of_tile(int tile) { return tile == 0 ? 256 : 128; }

}  // namespace

namespace pnpconv {

// which tile serves the filter gradient of `g` on the resident kernel (-1: none): zero padding, channels of one tap in whole 64-groups,
// filters in whole 64-groups, power-of-two output extents (every layer of the model), tensors < 2 GiB, >= 4096 output pixels
int wgrad_bf16r_tile(const pnp_conv_geom* g) {
    static const int off = getenv("PNP_BF16R_OFF") ? 1 : 0;
    if (off || !g || g->pad_mode != PNP_PAD_ZERO || g->C % 64 != 0 || g->K % 64 != 0) return -1;
    const long long P = (long long)g->N * g->OH * g->OW, ohw = (long long)g->OH * g->OW;
    if (P < 4096 || (g->OW & (g->OW - 1)) != 0 || (ohw & (ohw - 1)) != 0) return -1;
    const long long xin = (long long)g->N * g->H * g->W * g->C, yout = P * g->K;
    if (xin >= (1ll << 30) || yout >= (1ll << 30)) return -1;
    return (g->C % 128 == 0 ? 0 : 2) + (g->K % 128 == 0 ? 0 : 1);
}

// tile: 0 = 128 x 128, 1 = 128 x 64, 2 = 64 x 128, 3 = 64 x 64 (channels of one tap x filters); grid / nsplit / chunks_per_split (in
// units of wgrad_bf16r_chunk() pixels) planned by the caller (conv_igemm.hip: it owns the split planner and the reduce kernel)
bool launch_wgrad_bf16r(const ConvArgs& a, int tile, dim3 grid, hipStream_t st) {
#define PNP_WG(BM_, BN_, NBUF_)                                                                                                   \
    {                                                                                                                             \
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), 0.5 * conv_bytes(a), "conv_wgrad_bf16r_kernel<%d, %d, 2, 2, %d>", BM_, BN_, \
                        NBUF_);                                                                                                   \
        hipLaunchKernelGGL((conv_wgrad_bf16r_kernel<BM_, BN_, 2, 2, NBUF_>), grid, dim3(256), 0, st, a);                           \
        return true;                                                                                                              \
    }
    if (tile == 0) PNP_WG(128, 128, 2)
    if (tile == 1) PNP_WG(128, 64, 3)
    if (tile == 2) PNP_WG(64, 128, 3)
    if (tile == 3) PNP_WG(64, 64, 3)
#undef PNP_WG
    return false;
}

}  // namespace pnpconv

extern "C" {

int pnp_cast_bf16(const float* x, void* y, size_t n, void* stream) {
    PNP_REQUIRE(x && y && n > 0, "pnp_cast_bf16: bad argument");
    PNP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "pnp_cast_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)pnp_cdiv((long long)((n + 7) / 8), 256)), dim3(256), 0, (hipStream_t)stream, x,
                       (__bf16*)y, n);
    PNP_CHECK_LAUNCH("cast_bf16_kernel");
    return PNP_OK;
}

int pnp_filter_bf16(const float* w, void* w_io, void* w_oi, int32_t R, int32_t S, int32_t C, int32_t K, void* stream) {
    PNP_REQUIRE(w && (w_io || w_oi) && R > 0 && S > 0 && C > 0 && K > 0, "pnp_filter_bf16: bad argument");
    dim3 grid((unsigned)pnp_cdiv(K, 32), (unsigned)pnp_cdiv(C, 32), (unsigned)(R * S));
    hipLaunchKernelGGL(filter_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, (__bf16*)w_io, (__bf16*)w_oi, C, K);
    PNP_CHECK_LAUNCH("filter_bf16_kernel");
    return PNP_OK;
}

int32_t pnp_conv2d_bf16r_served(const pnp_conv_geom* g, int32_t kind) {
    if (kind == 2) return wgrad_bf16r_tile(g) >= 0 ? 1 : 0;
    return (kind == 0 || kind == 1) && served(g, kind) ? 1 : 0;
}

// batch-norm statistics partial rows the resident forward leaves behind (pixel tiles x wave rows of ITS tile)
int32_t pnp_conv2d_fwd_bf16r_stats_parts(const pnp_conv_geom* g) {
    if (!served(g, 0)) return 0;
    const long long M = (long long)g->N * g->OH * g->OW;
    const int tile = plan_tile(M, g->K);
    return pnp_cdiv(M, bm_of_tile(tile)) * wm_of_tile(tile);
}

int pnp_conv2d_fwd_bf16r(const void* xh, const void* w_oi, float* y, void* yh, const pnp_conv_geom* g, float keep_prob, uint64_t seed,
                         uint32_t stream_id, const float* stat_shift, float* stat_parts, size_t stat_parts_bytes, const float* scale,
                         const float* shift, const float* shortcut, int32_t Cs, float alpha, void* stream) {
    PNP_REQUIRE(served(g, 0), "pnp_conv2d_fwd_bf16r: geometry not served (pnp_conv2d_bf16r_served)");
    PNP_REQUIRE(xh && w_oi && y, "pnp_conv2d_fwd_bf16r: null pointer");
    PNP_REQUIRE(keep_prob > 0.f, "pnp_conv2d_fwd_bf16r: keep_prob must be > 0");
    ConvArgs a = base_args(xh, w_oi, y, g);
    a.y_h = (unsigned short*)yh;
    if (keep_prob < 1.f) {
        a.do_drop = 1;
        a.drop_keep = keep_prob;
        a.drop_key = pnp_drop_key(seed, stream_id);
        a.drop_thresh = pnp_drop_thresh(keep_prob);
        a.sp = pnp_step_params_ptr(); a.drop_sid = stream_id;
    }
    if (stat_parts) {
        const size_t need = (size_t)pnp_conv2d_fwd_bf16r_stats_parts(g) * 2 * g->K * sizeof(float);
        if (stat_parts_bytes < need) {
            pnp_set_error("pnp_conv2d_fwd_bf16r: parts buffer too small (%zu < %zu)", stat_parts_bytes, need);
            return PNP_EWORKSPACE;
        }
        a.stat_ws = stat_parts;
        a.stat_shift = stat_shift;
    }
    if (scale) {
        PNP_REQUIRE(shift, "pnp_conv2d_fwd_bf16r: scale without shift");
        if (shortcut) PNP_REQUIRE(Cs > 0 && Cs <= g->K && ((g->K - Cs) % 2) == 0, "pnp_conv2d_fwd_bf16r: bad shortcut channels");
        a.ep_scale = scale; a.ep_shift = shift; a.ep_res = shortcut; a.ep_cs = shortcut ? Cs : g->K; a.ep_alpha = alpha;
    }
    return launch_kind<0>(a, (hipStream_t)stream);
}

// dx = data gradient (+ residual): a stride-1 convolution of dy with the filter walked in reverse, operand rows from w_io = [tap][C][K]
int pnp_conv2d_dgrad_bf16r(const void* dyh, const void* w_io, const float* residual, float* dx, void* dxh, const pnp_conv_geom* g,
                           void* stream) {
    PNP_REQUIRE(served(g, 1), "pnp_conv2d_dgrad_bf16r: geometry not served (pnp_conv2d_bf16r_served)");
    PNP_REQUIRE(dyh && w_io && dx && residual != dx, "pnp_conv2d_dgrad_bf16r: bad pointer");
    if (g->stride != 1) {                   // one resident launch per stride phase, rows scattered with pixel stride `stride`
        PNP_REQUIRE(!residual && !dxh, "pnp_conv2d_dgrad_bf16r: strided geometry takes neither a residual nor a bf16 output");
        DgradPhase ph[16];
        const int nph = plan_phases(g, ph);
        for (int i = 0; i < nph; ++i) {
            const DgradPhase& p = ph[i];
            if (p.I == 0 || p.J == 0) continue;
            const pnp_conv_geom d = phase_geom(g, p);
            ConvArgs a = base_args(dyh, w_io, dx, &d);
            a.w_bytes = (unsigned)((size_t)g->R * g->S * g->C * g->K * 2);      // the FULL filter shadow
            a.o_s = g->stride; a.o_H = g->H; a.o_W = g->W; a.o_h0 = p.h0; a.o_w0 = p.w0;
            a.ph_st = g->stride; a.ph_pa = p.pa; a.ph_pb = p.pb; a.ph_S = g->S;
            if (int e = launch_kind<1>(a, (hipStream_t)stream)) return e;
        }
        return PNP_OK;
    }
    pnp_conv_geom d{};
    d.N = g->N; d.H = g->OH; d.W = g->OW; d.C = g->K; d.K = g->C; d.R = g->R; d.S = g->S;
    d.OH = g->H; d.OW = g->W;
    d.stride = 1; d.dil = g->dil;
    d.pad_t = g->dil * (g->R - 1) - g->pad_t;
    d.pad_l = g->dil * (g->S - 1) - g->pad_l;
    d.pad_mode = PNP_PAD_ZERO;
    d.dtype = PNP_DTYPE_BF16;
    ConvArgs a = base_args(dyh, w_io, dx, &d);
    a.y_h = (unsigned short*)dxh;
    a.res_add = residual;
    return launch_kind<1>(a, (hipStream_t)stream);
}

}  // extern "C"
