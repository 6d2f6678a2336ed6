// conv_wino_x3.hip — the GEMMs of the Winograd F(4x4, 3x3) / F(2x2, 3x3) routes on the bf16 matrix pipe with SPLIT operands ("x3": round 6).
//
// gfx950 has no tf32 and v_mfma_f32_32x32x2_f32 runs at 1/16 of v_mfma_f32_32x32x16_bf16's rate (MI355X_MICROARCH.md): the route's fp32
// GEMMs (conv_wino.hip: wino_gemm_kernel) were 61 % of the joint step's kernel time at 0.73 of a 157 TF/s roof.  A float32 value is the
// EXACT sum of three bf16 values (hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid): 3 x 8 significand bits, every difference exact
// in fp32), and a bf16 x bf16 product is exact in fp32, so
//     a b = hi.hi + (hi.mid + mid.hi) + (hi.lo + mid.mid + lo.hi) + [mid.lo + lo.mid + lo.lo  <= 2^-26 |a b|: dropped]
// — six MFMAs with fp32 accumulation reproduce the fp32 product to 2^-26, BELOW fp32's own rounding of it (2^-24).  The transform kernels
// write the three planes instead of one fp32 plane (6 instead of 4 bytes per value: wino_in_kernel / wino_filter_kernel / wino_dy_kernel,
// X3 = true), this file contracts them.  Roof: 2.5 PF / 6 = 417 TF/s fp32-equivalent (2.65x the fp32 MFMA peak).
//
// Accuracy is BETTER than the fp32 pipe's, not merely equal: the route's error is the fp32 accumulation chain over the channels in the
// transform domain (tools/wino_f43_study.py: 6.7e-7 with an fp64 GEMM, 4e-6 with fp32), so the kernel accumulates in CHUNKS of 96 channels:
// the MFMA accumulator of a chunk is added into a second register set (`total`) and restarted from zero.  tools/wino_bf16x3_study.py ->
// profiles/r06_wino_bf16x3_tolerance.txt: 512->512 6.1e-6 (fp32 chain) -> 1.1e-6..1.6e-6; 2 560-channel reduction 1.7e-5 -> 1.7e-6.
//
// Layout ("stage-major": what ONE stage of a tile reads of one plane is one contiguous run of whole 128-byte lines):
//          V3 [pos][C / 32][plane][T][32] bf16   (A operand: rows = tiles, reduction index C)
//          U3 [pos][C / 32][plane][K][32] bf16   (B operand: rows = output columns — the filter transform writes it transposed)
//          M  [pos][T][K] fp32                   (as the fp32 route's: wino_out_kernel is unchanged)
// What bounds it (tools/experiments/README.md round 6, tools/experiments/mfma_vmem_overlap.hip): a 128 x 128 tile streams 48 KB per
// 32-channel stage for 1 536 MFMA cycles per SIMD — 905 MB per 512->512 launch against ~9-10 TB/s that the CUs can pull out of L2 / MALL
// whatever the access pattern (measured: 2 MB, 12 MB and 64 MB footprints alike) = ~95 us of data movement next to 51 us of MFMAs.  And a
// wave that issues its own loads does not overlap the two AT ALL (LDS-DMA or register staging alike: T(both) = T(data) + T(MFMA), also
// in the micro-benchmark): in-order issue parks the wave behind its memory instructions.  So the workgroup is SPECIALISED: waves 0..3
// only contract (fragments by ds_read_b128, MFMAs, the accumulator chunks), waves 4..7 only stream (LDS-DMA buffer_load_dwordx4 ... lds,
// no registers, three LDS stages = 144 KB) — the micro-benchmark's 4 + 4 layout runs at the data rate with the MFMAs hidden under it.
// Stage = 32 channels of all three planes of both operands, LDS image [operand][plane][row][64 B] with conv_bf16r's bank swizzle applied on
// the source address; one raw s_barrier per stage for all eight waves.  Per 16-channel slice a consumer wave reads 12 fragments for 24 MFMAs.
// Chunked accumulation: every 64 channels the MFMA accumulator is added into `total` and restarted from zero (first MFMA with C = 0); the
// adds cost the consumer ~25 % of its time, which the data-bound stage hides.
#include <type_traits>
#include "conv_common.h"

using namespace pnpconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void lds_void;
#ifndef PNP_X3_AUX
#define PNP_X3_AUX 0      // cache policy bits of the LDS-DMA loads (experiment: 1 = sc0, 2 = nt, 3 = both: no measurable difference, tools/experiments/README.md)
#endif

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, PNP_X3_AUX);
#endif
}

struct X3Args {
    const unsigned short* V;      // [npos][C / X3_SW][3][T][X3_SW]
    const unsigned short* U;      // [npos][C / X3_SW][3][K][X3_SW]
    float* Mm;                    // [nsplit][npos][T][K]
    int T, C, K, npos;            // rows of A, reduction length, rows of B (= output columns), transform points
    int nblk_m, nblk_n, gn, xcd;
    int nsplit, spz;              // reduction splits (filter gradient: the reduction runs over the tiles), 32-channel stages per split (even)
};

// Consumers: BM / 32 waves, a 64 x (BN / 2) wave tile each ((BM / 64) x 2 of them); loaders: 4 waves.  BM = 128 (shipped): 8 waves; BM = 256
// (experiment): 12 waves, 3 per SIMD.  A stage = X3_SW (32) reduction elements of all three planes of both operands, LDS image
// [operand][plane][row][2 X3_SW bytes] with a bank swizzle applied on the source address of the DMA and again on the fragment reads; as many
// stages as fit 144 KB (128 x 128: 3 of 48 KB), all but one of them in flight.  C a multiple of 64 (an accumulation
// chunk).  PERSISTENT: one workgroup per CU (grid = min(items, 256)); hardware puts workgroup b on XCD b % 8 and every XCD owns a contiguous
// range of the logical item ids (whole transform points / reduction splits: V3[pos], U3[pos] in ONE L2), of which workgroup j of the XCD
// works through items j, j + 32, ...  The loaders run AHEAD across item boundaries: the stage stream never stops, so while the consumers store
// a finished tile the first stages of the next one are already on their way.  One descriptor per operand for the whole launch.
template <int SW>
__device__ __forceinline__ int swz(int row) {          // 16-byte chunk swizzle of an LDS row of 2 SW bytes
    return SW == 32 ? ((row >> 2) & 3) : ((row >> 3) & 1);
}

#ifndef PNP_X3_NL
#define PNP_X3_NL 4          // loader waves (experiment: 8 — tools/experiments/README.md round 6)
#endif
template <int BM, int BN, int KIND>
__global__ void __launch_bounds__(64 * (BM / 32 + PNP_X3_NL), (BM / 32 + PNP_X3_NL) / 4) wino_gemm_x3_kernel(X3Args g) {
    constexpr int SW = X3_SW;
    constexpr int NCW = BM / 32;               // consumer waves
    constexpr int kTermA[6] = {2, 1, 0, 1, 0, 0}, kTermB[6] = {0, 1, 2, 0, 1, 0};      // the kept products, smallest first: (plane of A, plane of B)
    constexpr int NL = PNP_X3_NL;              // loader waves
    constexpr int ROWB = 2 * SW;               // bytes per LDS row
    constexpr int LPR = ROWB / 16;             // 16-byte chunks (= lanes of a DMA piece) per row
    constexpr int RPI = 64 / LPR;              // rows per DMA instruction (64 lanes x 16 B = 1 KiB contiguous)
    constexpr int KS = SW / 16;                // 16-deep MFMA slices per stage
    constexpr int NPA = 3 * BM / RPI, NPB = 3 * BN / RPI, NP = NPA + NPB;      // DMA pieces of a stage: A's, then B's
    constexpr int LPS = (NP + NL - 1) / NL;    // pieces per loader wave and stage (a wave past the end of the list repeats its previous piece)
    constexpr int TM = 2, TN = BN / 2 / 32;
    constexpr int APL = BM * ROWB, BPL = BN * ROWB;            // bytes per plane image
    constexpr int STG = 3 * (APL + BPL);
    constexpr int NBUF = (147456 / STG) < 8 ? (147456 / STG) : 8;
    constexpr int SPC = 64 / SW;               // stages per accumulation chunk (64 channels)
    static_assert(NBUF >= 2 && (NBUF - 2) * LPS < 64, "vmcnt is a 6-bit counter");
    __shared__ __attribute__((aligned(256))) unsigned char lds[NBUF * STG];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = g.nblk_m * g.nblk_n;
    const int nitems = nblk * g.npos * g.nsplit;
    // this workgroup's items: base + first, base + first + step, ... < base + cnt
    int base = 0, cnt = nitems, first = (int)blockIdx.x, step = (int)gridDim.x;
    if (g.xcd) {
        const int NX = 8, x = (int)blockIdx.x % NX, q = nitems / NX, r = nitems % NX;
        base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        cnt = q + (x < r ? 1 : 0);
        first = (int)blockIdx.x / NX;
        step = ((int)gridDim.x - x + NX - 1) / NX;
    }
    if (first >= cnt) return;                 // (uniform for the workgroup)
    const int nstage_all = g.C / SW;
    // item -> (point, split, tile origin, first stage, stages)
    auto decode = [&](int item, int& pos, int& z, int& m0, int& n0, int& s0, int& nst) {
        const int pz = item / nblk;
        pos = pz / g.nsplit;
        z = pz - pos * g.nsplit;
        int mt, nt;
        tile_coords(item - pz * nblk, g.nblk_m, g.nblk_n, g.gn, mt, nt);
        m0 = mt * BM;
        n0 = nt * BN;
        s0 = z * g.spz;
        nst = min(g.spz, nstage_all - s0);   // (a multiple of SPC: host)
    };
    int gstages = 0;                          // stages of all items of this workgroup (both roles count the same barriers)
    for (int it = first; it < cnt; it += step) {
        int pos, z, m0, n0, s0, nst;
        decode(base + it, pos, z, m0, n0, s0, nst);
        gstages += nst;
    }

    if (wave >= NCW) {
        // ================================================= loader =================================================
        // piece q of an operand = plane q / (rows / RPI), row block q % (rows / RPI); this wave owns q = i NL + lw.  Lane (lrow, lchk) of a
        // piece moves 16 bytes: global chunk lchk ^ swz(row) of row lrow (2 SW contiguous bytes per row) -> LDS lane-linear
        const int lw = wave - NCW;
        const int lrow = lane / LPR, lchk = lane % LPR;
        const size_t planeV = (size_t)3 * g.T * g.C, planeU = (size_t)3 * g.K * g.C;       // elements per transform point
        const __amdgpu_buffer_rsrc_t rv = make_rsrc(reinterpret_cast<const float*>(g.V), (unsigned)(planeV * g.npos * 2));
        const __amdgpu_buffer_rsrc_t ru = make_rsrc(reinterpret_cast<const float*>(g.U), (unsigned)(planeU * g.npos * 2));
        const int sstrideV = 3 * g.T * ROWB, sstrideU = 3 * g.K * ROWB;          // bytes per stage (SW channels of the three planes)
        // this wave's pieces: q = i NL + lw (past the end: its previous piece again — the same bytes to the same place, so that every wave
        // issues exactly LPS pieces per stage); q < NPA: A, plane q / (BM / RPI), row block q % (BM / RPI); else B likewise
        unsigned pvo[LPS];
        int pdst[LPS];
        bool pisa[LPS];
#pragma unroll
        for (int i = 0; i < LPS; ++i) {
            int q = i * NL + lw;
            if (q >= NP) q -= NL;
            pisa[i] = q < NPA;
            const int qq = pisa[i] ? q : q - NPA, rows = pisa[i] ? BM : BN;
            pdst[i] = (pisa[i] ? 0 : 3 * APL) + (qq / (rows / RPI)) * (pisa[i] ? APL : BPL) + (qq % (rows / RPI)) * (RPI * ROWB);
        }
        // the cursor: item `c_it` (index into this workgroup's list), its next stage `c_j` of `c_nst`; offsets of the item's rows
        int c_it = first, c_j = 0, c_nst = 0, c_s0 = 0;
        auto setup = [&](int it) {
            int pos, z, m0, n0;
            decode(base + it, pos, z, m0, n0, c_s0, c_nst);
            const unsigned pv = (unsigned)(pos * planeV * 2), pu = (unsigned)(pos * planeU * 2);
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                int q = i * NL + lw;
                if (q >= NP) q -= NL;
                const bool isa = q < NPA;
                const int qq = isa ? q : q - NPA, rows = isa ? BM : BN, lim = isa ? g.T : g.K;
                const int plane = qq / (rows / RPI), rb = qq % (rows / RPI);
                const int r = rb * RPI + lrow, m = (isa ? m0 : n0) + r;
                pvo[i] = (m < lim) ? (isa ? pv : pu) + (unsigned)((plane * lim + m) * ROWB + ((lchk ^ swz<SW>(r)) << 4)) : OOB2;
            }
        };
        setup(c_it);
        // issues the cursor's stage into buffer `buf` and advances; past the last item: the last stage again (into a buffer nobody reads
        // any more) so that every iteration issues exactly LPS pieces and the vmcnt arithmetic stays uniform
        auto issue_next = [&](int buf) {
            unsigned char* bp = lds + buf * STG;
            const int cc = c_s0 + c_j;
#pragma unroll
            for (int i = 0; i < LPS; ++i) {
                if (pisa[i]) dma16(rv, (lds_void*)(bp + pdst[i]), pvo[i], cc * sstrideV);
                else dma16(ru, (lds_void*)(bp + pdst[i]), pvo[i], cc * sstrideU);
            }
            if (c_j + 1 < c_nst) {
                ++c_j;
            } else if (c_it + step < cnt) {
                c_it += step;
                c_j = 0;
                setup(c_it);
            }
        };
#pragma unroll
        for (int b_ = 0; b_ < NBUF - 1; ++b_) issue_next(b_);
        int nb = NBUF - 1;
        for (int gsi = 0; gsi < gstages; ++gsi) {
            wait_vm<(NBUF - 2) * LPS>();         // stage gsi has landed (the NBUF - 2 younger ones may be in flight)
            __builtin_amdgcn_s_barrier();        // consumers: done with stage gsi - 1, i.e. with buffer (gsi + NBUF - 1) % NBUF
            issue_next(nb);
            nb = nb == NBUF - 1 ? 0 : nb + 1;
        }
        wait_vm<0>();
        return;
    }
    // ================================================= consumer =================================================
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (BN / 2);
    // fragment reads: lane (l31, h) reads row l31 of a 32-row block, 16-byte chunk (2 ks + h) ^ swz(row)
    const int l31 = lane & 31, h = lane >> 5;
    int foff[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = l31 * ROWB + (((2 * ks + h) ^ swz<SW>(l31)) << 4);

    Acc<TM, TN> cur, total;
    int buf = 0;
    auto stage = [&](auto fc) {
        constexpr bool first_ = decltype(fc)::value;
        wait_lgkm0();                        // own fragment reads of the previous stage have retired
        __builtin_amdgcn_s_barrier();        // the loaders' part of this stage is in LDS
        asm volatile("" ::: "memory");
        const unsigned char* A = lds + buf * STG + wm0 * ROWB;
        const unsigned char* B = lds + buf * STG + 3 * APL + wn0 * ROWB;
        bf16x8 af[KS][3][TM], bfr[KS][3][TN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) af[ks][p][tm] = *reinterpret_cast<const bf16x8*>(A + p * APL + tm * (32 * ROWB) + foff[ks]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bfr[ks][p][tn] = *reinterpret_cast<const bf16x8*>(B + p * BPL + tn * (32 * ROWB) + foff[ks]);
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int tr = 0; tr < 6; ++tr) {
                const int pa = kTermA[tr], pb = kTermB[tr];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        if (first_ && ks == 0 && tr == 0) {
                            const f32x16 z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                            cur.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][pa][tm], bfr[ks][pb][tn], z, 0, 0, 0);
                        } else {
                            cur.v[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][pa][tm], bfr[ks][pb][tn], cur.v[tm][tn], 0, 0, 0);
                        }
                    }
            }
        buf = buf == NBUF - 1 ? 0 : buf + 1;
    };
    ConvArgs e{};            // plain [T][K] rows of a transform point: conv_epilogue with every feature off
    e.M = g.T;
    e.K = g.K;
    e.nsplit = 1;
    for (int it = first; it < cnt; it += step) {
        int pos, z, m0, n0, s0, nst;
        decode(base + it, pos, z, m0, n0, s0, nst);
        total.zero();
        for (int j = 0; j < nst; j += SPC) {        // one accumulation chunk = 64 channels (host: C % 64 == 0)
            stage(std::true_type{});
#pragma unroll
            for (int q = 1; q < SPC; ++q) stage(std::false_type{});
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) total.v[tm][tn] += cur.v[tm][tn];
        }
        conv_epilogue<TM, TN>(e, total, g.Mm + ((size_t)z * g.npos + pos) * g.T * g.K, m0, n0, wm0, wn0, lane, 0);
    }
}

}  // namespace

namespace pnpconv {

bool wino_x3_dims_ok(int T, int C, int K) {
    // ONE buffer descriptor per operand for the whole launch: up to 36 transform points x 3 planes of T x C (K x C) bf16 below 2 GiB
    return (C % 64) == 0 && 36.0 * T * C * 6.0 < 2147483648.0 && 36.0 * K * C * 6.0 < 2147483648.0;
}

// sym = 0 / 1: F(2x2) forward / data gradient, 2 / 3: F(4x4) (names the symbol like wino_gemm_kernel's last template argument), 4 / 5: the
// filter gradient's GEMMs S[pos] = V[pos]^T x Y[pos] of F(2x2) / F(4x4) (T := channels, C := tiles padded to 64, K := filters; the
// reduction over the tiles is split nsplit ways, stages_per_split 32-tile stages each: M = [nsplit][npos][T][K])
int launch_wino_gemm_x3(const unsigned short* V3, const unsigned short* U3, float* Mm, int T, int C, int K, int npos, int sym, int gn, int xcd,
                        int nsplit, int stages_per_split, hipStream_t st) {
    PNP_REQUIRE(wino_x3_dims_ok(T, C, K), "launch_wino_gemm_x3: operands too large for one buffer descriptor per transform point");
    PNP_REQUIRE(sym >= 0 && sym < 6, "launch_wino_gemm_x3: bad symbol index");
    // (callers count stages of 32 reduction elements; the kernel's stages are X3_SW wide)
    stages_per_split *= 32 / X3_SW;
    PNP_REQUIRE(nsplit >= 1 && (stages_per_split % (64 / X3_SW)) == 0 && (long long)nsplit * stages_per_split >= C / X3_SW &&
                    (long long)(nsplit - 1) * stages_per_split < C / X3_SW,
                "launch_wino_gemm_x3: bad reduction split");
    X3Args g{};
    g.V = V3; g.U = U3; g.Mm = Mm; g.T = T; g.C = C; g.K = K; g.npos = npos;
    const bool narrow = K <= 64;
    // 256-row tiles (8 consumer waves + 4 loaders = 3 waves per SIMD, 0.75 x the operand traffic of 128 x 128): built and measured —
    // SLOWER (512->512 108 -> 114-123 us with two LDS stages of 72 KB as well as with four of 36 KB: the data rate per CU falls from 13 to
    // 9 bytes per clock; tools/experiments/README.md round 6).  Off; PNP_X3_BM=256 selects it for A/B runs.
    static const int bm_force = getenv("PNP_X3_BM") ? atoi(getenv("PNP_X3_BM")) : 0;
    const bool tall = !narrow && bm_force == 256 && T >= 256;
    g.nblk_m = pnp_cdiv(T, tall ? 256 : 128); g.nblk_n = pnp_cdiv(K, narrow ? 64 : 128);
    g.gn = gn; g.xcd = xcd ? 1 : 0;
    g.nsplit = nsplit; g.spz = stages_per_split;
    const long long nitems = (long long)g.nblk_m * g.nblk_n * npos * nsplit;
    static const int persist = getenv("PNP_X3_PERSIST") ? atoi(getenv("PNP_X3_PERSIST")) : 1;      // (0: one item per workgroup, for A/B runs)
    const dim3 grid((unsigned)((persist && nitems > 256) ? 256 : nitems));
    // flops = the bf16 MFMA flops the kernel EXECUTES (six products per fp32 multiply-add): its roof is the dense bf16 peak
    const double fl = 6.0 * 2.0 * npos * (double)T * C * K;
    const double by = (double)npos * (6.0 * ((double)T * C + (double)C * K) + 4.0 * (double)nsplit * T * K);
    PnpProfScope ps(sym >= 4 ? PNP_PROF_CONV_WGRAD : prof_class(sym & 1), st, fl, by, "wino_gemm_x3_kernel<%d, %d, %d>", tall ? 256 : 128, narrow ? 64 : 128, sym);
#define PNP_X3_LAUNCH(BN_, SYM_) hipLaunchKernelGGL((wino_gemm_x3_kernel<128, BN_, SYM_>), grid, dim3(64 * (4 + PNP_X3_NL)), 0, st, g)
#define PNP_X3_LAUNCH_TALL(SYM_) hipLaunchKernelGGL((wino_gemm_x3_kernel<256, 128, SYM_>), grid, dim3(64 * (8 + PNP_X3_NL)), 0, st, g)
    if (tall) {
        switch (sym) {
            case 0: PNP_X3_LAUNCH_TALL(0); break;
            case 1: PNP_X3_LAUNCH_TALL(1); break;
            case 2: PNP_X3_LAUNCH_TALL(2); break;
            case 3: PNP_X3_LAUNCH_TALL(3); break;
            case 4: PNP_X3_LAUNCH_TALL(4); break;
            default: PNP_X3_LAUNCH_TALL(5); break;
        }
    } else if (narrow) {
        switch (sym) {
            case 0: PNP_X3_LAUNCH(64, 0); break;
            case 1: PNP_X3_LAUNCH(64, 1); break;
            case 2: PNP_X3_LAUNCH(64, 2); break;
            case 3: PNP_X3_LAUNCH(64, 3); break;
            case 4: PNP_X3_LAUNCH(64, 4); break;
            default: PNP_X3_LAUNCH(64, 5); break;
        }
    } else {
        switch (sym) {
            case 0: PNP_X3_LAUNCH(128, 0); break;
            case 1: PNP_X3_LAUNCH(128, 1); break;
            case 2: PNP_X3_LAUNCH(128, 2); break;
            case 3: PNP_X3_LAUNCH(128, 3); break;
            case 4: PNP_X3_LAUNCH(128, 4); break;
            default: PNP_X3_LAUNCH(128, 5); break;
        }
    }
#undef PNP_X3_LAUNCH
#undef PNP_X3_LAUNCH_TALL
    PNP_CHECK_LAUNCH("wino_gemm_x3_kernel");
    return PNP_OK;
}

}  // namespace pnpconv
