// conv_x3_direct.hip — DIRECT 3x3 stride-1 convolutions of the narrow layers (32 / 64 input channels) on the bf16 matrix pipe with split
// operands (round 6; prototype and ablations: tools/experiments/x3_direct_conv.hip, profiles/r06_x3_direct_conv_prototype.txt).
//
// The 64-channel 3x3 layers on the large maps (the critics' cls1 / cls2 blocks, reference adversarial.py:337-366) are where neither route is
// good: the direct fp32-MFMA kernels stop at 0.6-0.7 of 157 TF/s, F(4x4) on the fp32 pipe is bound by its own 2.25x transformed tensors
// (cls1 64->64 at B = 16: 2.95 GB through HBM per convolution, 0.593 ms), and split-bf16 Winograd operands would add 50 % to that traffic.
// Here the split-bf16 idea (conv_wino_x3.hip: an fp32 value IS the sum of three bf16 planes, six plane products reproduce the fp32 product
// to 2^-26) runs WITHOUT Winograd:
//   * a workgroup owns a 16 x 16 output tile x 64 filters.  The tile's 18 x 18 halo patch of 32 input channels is loaded ONCE per channel
//     half by three loader waves (global fp32 -> VGPR -> hi / mid / lo bf16 -> LDS [plane][pixel][64 B], the split spread over five stages),
//     double-buffered across halves: the nine taps are LDS address offsets, every input value crosses L2 -> CU once per tile;
//   * filters come as a pre-split image [filter block][half][tap][plane][64 filters][32 channels] bf16 (x3d_filter_kernel: forward as is,
//     data gradient flipped and transposed) — 12 KB per (half, tap) stage, streamed by one loader wave with LDS-DMA into three buffers;
//   * four consumer waves (64 pixels x 64 filters each): per stage 12 filter-fragment reads after the barrier, the 12 pixel-fragment reads
//     of the NEXT tap issued before it (same patch: no barrier in between), 48 v_mfma_f32_32x32x16_bf16; the second pixel row of a 32-pixel
//     block is rotated by two columns so that every ds_read_b128 lane group sees 16 distinct 16-byte slots; one raw s_barrier per stage;
//   * persistent workgroups (one per CU) over (tile, filter block) items, loaders running ahead; the stages of an item fully unrolled.
// Epilogue = conv_common.h's conv_epilogue restated for a TILED pixel order (a 32-row MFMA block is two image rows of 16 pixels): dropout
// (same counter hash on the flat output index, a division like tf.nn.dropout), residual add (data gradients), batch-norm statistics
// partials (one per consumer wave = 64 pixels), then stores of 128 contiguous bytes per pixel.  Accuracy: every product exact, one fp32
// chain of 9 C terms — 8e-7 of max|ref| on 64 -> 64 (the direct fp32 kernel 2.9e-6, F(4x4) 1.2e-6 .. 8e-6).
// Entry: launch_wino() hands the layers x3d_chosen() takes to launch_x3_direct() (they are layers the Winograd planner owns: workspace and
// partial-count queries go through the same functions).  PNP_X3_DIRECT / pnp_conv2d_x3_direct(): 0 off, 1 where a launch fills the chip, 2 wherever the shapes allow.
#include <atomic>
#include <cstdlib>
#include "conv_common.h"

using namespace pnpconv;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

constexpr int TH = 16, TW = 16, PH = TH + 2, PW = TW + 2, NPX = PH * PW;      // 324 patch pixels
constexpr int ROWB = 64;                           // bytes per LDS row: 32 channels of one plane
constexpr int PLANE_P = NPX * ROWB;                // 20 736 B per patch plane
constexpr int PATCH = 3 * PLANE_P;                 // 62 208 B per patch (one channel half)
constexpr int NF = 3;                              // filter stage buffers: NF - 1 stages in flight
constexpr int LDS_BYTES = 2 * PATCH + NF * 3 * 64 * ROWB;      // 161 280 B (filter stage of a 64-filter block: 3 planes x 4 096 B = 12 288 B)
constexpr int NPL = 192;                           // patch-loader lanes (3 waves)
constexpr int NPI = (NPX * 8 + NPL - 1) / NPL;     // float4 loads per patch-loader lane: 14

__device__ __forceinline__ int swz(int row) { return (row >> 2) & 3; }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t r, lds_void* dst, unsigned voff, int soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, dst, 16, voff, soff, 0, 0);
#endif
}

struct X3dArgs {
    const float* x;               // [N][H][W][C]
    const unsigned short* w3;     // [K / 64][C / 32][9][3][64][32] bf16
    float* y;                     // [N][OH][OW][K]
    int N, H, W, C, K, OH, OW, pad_t, pad_l;
    int M;                        // N OH OW
    int do_drop;
    uint32_t drop_thresh, drop_key;
    float drop_keep;
    const pnp_step_params* sp;
    uint32_t drop_sid;
    const float* res_add;         // [M][K] or null
    float* stat_ws;               // [parts][2][K] or null; part = 4 tile + consumer wave
    const float* stat_shift;
};

// NH = C / 32 channel halves (1 or 2); KW = filters per block (64; 32 for the 32-filter layers: the data gradient of a 32 -> 64 layer);
// KIND 0 / 1 only names the symbol (forward / data gradient: the difference is the filter image)
template <int NH, int KW, int KIND>
__global__ void __launch_bounds__(512, 1) conv_x3_direct_kernel(X3dArgs a) {
    constexpr int NSTG = 9 * NH;
    constexpr int PLANE_F = KW * ROWB;             // bytes per filter plane of a stage
    constexpr int FSTG = 3 * PLANE_F;              // bytes per (half, tap) stage
    constexpr int NPC = FSTG / 1024;               // LDS-DMA pieces per stage (1 KiB each)
    constexpr int TN = KW / 32;
    static_assert(NSTG % NF == 0, "the filter buffer of a stage must not depend on the item");
    __shared__ __attribute__((aligned(256))) unsigned char lds[LDS_BYTES];
    unsigned char* const patch0 = lds;
    unsigned char* const filt0 = lds + 2 * PATCH;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int tiles_x = a.OW / TW, tiles_y = a.OH / TH, KB = a.K / KW;
    const int nitems = a.N * tiles_x * tiles_y * KB;
    int myitems = 0;
    for (int i = blockIdx.x; i < nitems; i += gridDim.x) ++myitems;
    if (myitems == 0) return;                       // (uniform for the workgroup)
    const int gstages = myitems * NSTG;
    // item -> (filter block, image, tile origin); consecutive items = the filter blocks of one tile (its patch is in L2 for the second)
    auto item_origin = [&](int it, int& kb, int& tile, int& n, int& oh0, int& ow0) {
        const int id = blockIdx.x + it * gridDim.x;
        tile = id / KB;
        kb = id - tile * KB;
        n = tile / (tiles_x * tiles_y);
        const int r = tile - n * tiles_x * tiles_y;
        oh0 = (r / tiles_x) * TH;
        ow0 = (r % tiles_x) * TW;
    };

    if (wave == 4) {
        // ============================ filter loader: one 12 KB stage per barrier, LDS-DMA ============================
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(reinterpret_cast<const float*>(a.w3), (unsigned)(KB * NSTG * FSTG));
        // piece i (1 KiB = 16 rows of 64 B): lane (lrow = lane / 4, lchk = lane % 4) fetches chunk lchk ^ swz(row) of its row
        const int lrow = lane >> 2, lchk = lane & 3;
        constexpr int PPP = KW / 16;                                   // pieces per plane
        unsigned vo[NPC];
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int row = (i % PPP) * 16 + lrow;                    // row within the plane (filter index); plane = i / PPP
            vo[i] = (unsigned)((i / PPP) * PLANE_F + row * ROWB + ((lchk ^ swz(row)) << 4));
        }
        auto issue = [&](int g) {
            const int it = g / NSTG, s = g - it * NSTG;
            const int kb = (int)((blockIdx.x + it * gridDim.x) % KB);
            unsigned char* bp = filt0 + (g % NF) * FSTG;
#pragma unroll
            for (int i = 0; i < NPC; ++i) dma16(rw, (lds_void*)(bp + i * 1024), vo[i], (kb * NSTG + s) * FSTG);
        };
#pragma unroll
        for (int b = 0; b < NF - 1; ++b)
            if (b < gstages) issue(b);
        for (int g = 0; g < gstages; ++g) {
            // stage g has landed; up to NF - 2 younger stages stay in flight (the tail issues nothing: drain)
            if (g + NF - 2 < gstages) wait_vm<NPC * (NF - 2)>();
            else wait_vm0();
            __builtin_amdgcn_s_barrier();
            if (g + NF - 1 < gstages) issue(g + NF - 1);             // into the buffer of stage g - 1: every consumer is past it
        }
        wait_vm0();
        return;
    }
    if (wave > 4) {
        // ============================ patch loaders: global fp32 -> three bf16 planes -> LDS ============================
        const int pl = (wave - 5) * 64 + lane;                        // 0..191
        f32x4 st[NPI];
        auto load = [&](int slot) {                                   // slot = item * NH + half
            int kb, tile, n, oh0, ow0;
            item_origin(slot / NH, kb, tile, n, oh0, ow0);
            const int hsel = slot % NH;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                const int e = pl + i * NPL;
                const int q = e >> 3, j = e & 7;
                const int pr = q / PW, pc = q - pr * PW;
                const int ih = oh0 - a.pad_t + pr, iw = ow0 - a.pad_l + pc;
                const bool ok = (e < NPX * 8) & ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (ok) v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)n * a.H + ih) * a.W + iw) * a.C + hsel * 32 + j * 4);
                st[i] = v;
            }
        };
        auto store = [&](int slot, int i0, int i1) {
            unsigned char* pb = patch0 + (slot & 1) * PATCH;
#pragma unroll
            for (int i = 0; i < NPI; ++i) {
                if (i < i0 || i >= i1) continue;
                const int e = pl + i * NPL;
                if (e >= NPX * 8) continue;
                const int q = e >> 3, j = e & 7;
                bf16x4 hi, mi, lo;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = st[i][c];
                    const __bf16 h = (__bf16)v;
                    const float r1 = v - (float)h;
                    const __bf16 m = (__bf16)r1;
                    const float r2 = r1 - (float)m;
                    hi[c] = h; mi[c] = m; lo[c] = (__bf16)r2;
                }
                const int off = q * ROWB + (((j >> 1) ^ swz(q)) << 4) + (j & 1) * 8;
                *reinterpret_cast<bf16x4*>(pb + off) = hi;
                *reinterpret_cast<bf16x4*>(pb + PLANE_P + off) = mi;
                *reinterpret_cast<bf16x4*>(pb + 2 * PLANE_P + off) = lo;
            }
        };
        const int nslots = myitems * NH;
        load(0);
        wait_vm0();
        store(0, 0, NPI);
        for (int g = 0; g < gstages; ++g) {
            const int slot = g / 9, j = g - slot * 9;
            if (j == 0) wait_lgkm0();                                  // this slot's patch is in LDS
            __builtin_amdgcn_s_barrier();
            if (j == 0 && slot + 1 < nslots) load(slot + 1);
            // split + LDS stores of the next patch spread over stages 3..7 (3 float4 per lane each): the vector ALU work of one stage stays
            // well under the consumers' MFMA time, so this wave is never the last at a barrier
            if (j >= 3 && j <= 7 && slot + 1 < nslots) {
                if (j == 3) wait_vm0();
                if (j == 3) store(slot + 1, 0, 3);
                else if (j == 4) store(slot + 1, 3, 6);
                else if (j == 5) store(slot + 1, 6, 9);
                else if (j == 6) store(slot + 1, 9, 12);
                else store(slot + 1, 12, NPI);
            }
        }
        return;
    }
    // ============================ consumers: 64 pixels (4 output rows x 16) x 64 filters per wave ============================
    // MFMA: A = pixels (rows), B = filters (columns): D[pixel][filter], lane = filter l31 of block tn, rows (i & 3) + 8 (i >> 2) + 4 h.
    // Pixel of MFMA row r in 32-pixel block tm: r < 16: image row 4 w + 2 tm, column r; else the next image row, column (r - 2) mod 16 — the
    // rotation makes the patch index q of the second row congruent (mod 16) to the first row's: 16 distinct bank slots per lane group.
    const int l31 = lane & 31, h = lane >> 5;
    const int prow = l31 >> 4, pcol = (l31 < 16) ? l31 : ((l31 - 16 + 14) & 15);
    int aq[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) aq[tm] = (4 * wave + 2 * tm + prow) * PW + pcol;
    int woff[TN][2];                                // [tn][ks]
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) woff[tn][ks] = (tn * 32 + l31) * ROWB + (((2 * ks + h) ^ swz(tn * 32 + l31)) << 4);
    constexpr int kTermX[6] = {2, 1, 0, 1, 0, 0}, kTermW[6] = {0, 1, 2, 0, 1, 0};      // the kept plane products, smallest first
    f32x16 acc[2][TN];                              // [tm (pixel block)][tn (filter block)]
    bf16x8 xf[2][3][2];                             // [ks][plane][tm]: this tap's pixel fragments
    bf16x8 xn[3][2];                                // [plane][tm]: the NEXT tap's first 16-channel slice, read one stage ahead
    auto x_frag = [&](int pbuf, int tap, int ks, int p, int tm) {
        const unsigned char* A = patch0 + pbuf * PATCH;
        const int q = aq[tm] + (tap / 3) * PW + (tap % 3);
        return *reinterpret_cast<const bf16x8*>(A + p * PLANE_P + q * ROWB + (((2 * ks + h) ^ swz(q)) << 4));
    };
    const uint32_t dkey = a.do_drop ? pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid) : 0u;
    for (int it = 0; it < myitems; ++it) {
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[tm][tn][e] = 0.f;
        const int slot0 = it * NH;                   // (NH = 1: the patch buffer alternates with the item; NH = 2: half hsel is buffer hsel)
#pragma unroll
        for (int s = 0; s < NSTG; ++s) {
            const int hsel = s / 9, tap = s % 9;     // (static after unrolling; the filter buffer of global stage NSTG it + s is s mod NF)
            const int fb = s % NF;
            const int pbuf = (slot0 + hsel) & 1;
            wait_lgkm0();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // this tap's first slice: prefetched by the previous stage, or (first tap of a half: the patch only became visible with this
            // barrier) read now
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) xf[0][p][tm] = (tap == 0) ? x_frag(pbuf, 0, 0, p, tm) : xn[p][tm];
            const unsigned char* B = filt0 + fb * FSTG;
            bf16x8 wf[2][3][TN];                     // [ks][plane][tn]
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) wf[0][p][tn] = *reinterpret_cast<const bf16x8*>(B + p * PLANE_F + woff[tn][0]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) xf[1][p][tm] = x_frag(pbuf, tap, 1, p, tm);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) wf[1][p][tn] = *reinterpret_cast<const bf16x8*>(B + p * PLANE_F + woff[tn][1]);
            }
            if (tap != 8) {                          // the next tap's first slice: same patch, no barrier in between
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm) xn[p][tm] = x_frag(pbuf, tap + 1, 0, p, tm);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int trm = 0; trm < 6; ++trm)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xf[ks][kTermX[trm]][tm], wf[ks][kTermW[trm]][tn], acc[tm][tn], 0, 0, 0);
        }
        // ---------------- epilogue (conv_common.h: conv_epilogue, for the tiled pixel order): phases of pure loads / arithmetic, then stores
        int kb, tile, n, oh0, ow0;
        item_origin(it, kb, tile, n, oh0, ow0);
        // MFMA row (i & 3) + 8 (i >> 2) + 4 h of block tm -> linear output pixel (recomputed where it is used: no index arrays kept live)
        auto mlin_of = [&](int tm, int i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
            const int orow = oh0 + 4 * wave + 2 * tm + (r >> 4);
            const int ocol = ow0 + ((r < 16) ? r : ((r - 16 + 14) & 15));
            return (n * a.OH + orow) * a.OW + ocol;
        };
        int ncol[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) ncol[tn] = kb * KW + tn * 32 + l31;
#define X3D_FOR                                                      \
    _Pragma("unroll") for (int tm = 0; tm < 2; ++tm)                 \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)            \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)
        if (a.do_drop) {
            X3D_FOR {
                const uint32_t idx = (uint32_t)((size_t)mlin_of(tm, i) * a.K + ncol[tn]);
                acc[tm][tn][i] = pnp_drop_keep(idx, dkey, a.drop_thresh) ? acc[tm][tn][i] / a.drop_keep : 0.f;
            }
        }
        if (a.res_add) {
            const __amdgpu_buffer_rsrc_t rr = make_rsrc(a.res_add, (unsigned)((size_t)a.M * a.K * 4));
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {          // (one pixel block at a time: all loads, then all adds)
                f32x16 rv[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int i = 0; i < 16; ++i) rv[tn][i] = bload1(rr, (unsigned)(mlin_of(tm, i) * a.K + ncol[tn]) * 4u);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[tm][tn][i] += rv[tn][i];
            }
        }
        if (a.stat_ws) {
            const int part = tile * 4 + wave;
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const float shift = a.stat_shift ? a.stat_shift[ncol[tn]] : 0.f;
                float ssum = 0.f, ssq = 0.f;
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const float d = acc[tm][tn][i] - shift;
                        ssum += d;
                        ssq = fmaf(d, d, ssq);
                    }
                const float s_ = ssum + __shfl_xor(ssum, 32, 64);      // the two half-waves hold the same filter, disjoint pixels
                const float q_ = ssq + __shfl_xor(ssq, 32, 64);
                if (h == 0) {
                    a.stat_ws[((size_t)part * 2 + 0) * a.K + ncol[tn]] = s_;
                    a.stat_ws[((size_t)part * 2 + 1) * a.K + ncol[tn]] = q_;
                }
            }
        }
        const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.y, (unsigned)((size_t)a.M * a.K * 4));
        X3D_FOR bstore1(ry, (unsigned)(mlin_of(tm, i) * a.K + ncol[tn]) * 4u, acc[tm][tn][i]);
#undef X3D_FOR
    }
}

// the filter image: [K / KW][C / 32][tap][plane][KW filters][32 channels] bf16 (KW = 64; 32 for a 32-filter layer).  FLIP = false: w is [3][3][C][K] (forward); true: w is the
// FORWARD filter [3][3][K][C] of a data gradient computed as a convolution of dy (C = the forward's K): taps reversed, roles transposed
template <bool FLIP>
__global__ void __launch_bounds__(256) x3d_filter_kernel(const float* __restrict__ w, unsigned short* __restrict__ w3, int C, int K, int KW) {
    const int NH = C >> 5, KB = K / KW;
    const int total = KB * NH * 9 * KW * 32;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int i = e & 31, o = (e >> 5) % KW;
    int r = (e >> 5) / KW;
    const int tap = r % 9; r /= 9;
    const int hf = r % NH, kb = r / NH;
    const int ic = hf * 32 + i, oc = kb * KW + o;
    const float v = FLIP ? w[((size_t)(8 - tap) * K + oc) * C + ic] : w[((size_t)tap * C + ic) * K + oc];
    const __bf16 b0 = (__bf16)v;
    const float r1 = v - (float)b0;
    const __bf16 b1 = (__bf16)r1;
    const float r2 = r1 - (float)b1;
    const __bf16 b2 = (__bf16)r2;
    const size_t base = ((((size_t)kb * NH + hf) * 9 + tap) * 3) * ((size_t)KW * 32) + (size_t)o * 32 + i;
    w3[base] = __builtin_bit_cast(unsigned short, b0);
    w3[base + (size_t)KW * 32] = __builtin_bit_cast(unsigned short, b1);
    w3[base + 2 * (size_t)KW * 32] = __builtin_bit_cast(unsigned short, b2);
}

std::atomic<int> g_x3d_mode{-1};

int x3d_mode() {
    int m = g_x3d_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = getenv("PNP_X3_DIRECT") ? atoi(getenv("PNP_X3_DIRECT")) : 1;
        g_x3d_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

// mode 1: layers with at least one (tile, filter block) item per CU — a persistent launch of fewer leaves CUs idle and the old route wins;
// mode 2: wherever the shapes allow (tests)
bool dims_ok(int R, int S, int stride, int dil, int C, int K, int OH, int OW, long long M) {
    if (!(R == 3 && S == 3 && stride == 1 && dil == 1 && (C == 32 || C == 64) && (K == 32 || K == 64 || K == 128) && (OH % TH) == 0 && (OW % TW) == 0 &&
          M * (long long)K < (1ll << 30)))
        return false;
    return x3d_mode() >= 2 || (M / (TH * TW)) * (K == 32 ? 1 : K / 64) >= 256;
}

}  // namespace

namespace pnpconv {

bool x3d_chosen(const pnp_conv_geom* g) {
    return x3d_mode() > 0 && g->dtype == PNP_DTYPE_F32 && (g->pad_mode == PNP_PAD_ZERO || (g->pad_t == 0 && g->pad_l == 0)) && dims_ok(g->R, g->S, g->stride, g->dil, g->C, g->K, g->OH, g->OW, (long long)g->N * g->OH * g->OW);
}

bool x3d_chosen(const ConvArgs& a) {
    return x3d_mode() > 0 && a.dtype == PNP_DTYPE_F32 && (a.pad_mode == PNP_PAD_ZERO || (a.pad_t == 0 && a.pad_l == 0)) && a.ups == 1 && a.o_s == 0 && a.y_h == nullptr && a.ep_scale == nullptr &&
           dims_ok(a.R, a.S, a.stride, a.dil, a.C, a.K, a.OH, a.OW, a.M);
}

// one partial per consumer wave: 64 pixels (4 rows x 16) — M / 64 of them
int x3d_stats_parts(const pnp_conv_geom* g) { return (int)(((long long)g->N * g->OH * g->OW) / 64); }

size_t x3d_filter_bytes(int C, int K) { return (size_t)9 * C * K * 6; }

int launch_x3_direct(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    const size_t fbytes = x3d_filter_bytes(a.C, a.K);
    if (!ws || ws_bytes < fbytes) {
        pnp_set_error("launch_x3_direct: workspace too small (%zu < %zu)", ws_bytes, fbytes);
        return PNP_EWORKSPACE;
    }
    unsigned short* w3 = (unsigned short*)ws;
    const int cls = prof_class(kind);
    const int KW = a.K == 32 ? 32 : 64;
    {
        const int total = 9 * a.C * a.K;
        PnpProfScope ps(cls, st, 0.0, 4.0 * total + 6.0 * total, "x3d_filter_kernel<%s>", flip_transpose ? "true" : "false");
        if (flip_transpose) hipLaunchKernelGGL((x3d_filter_kernel<true>), dim3((unsigned)pnp_cdiv(total, 256)), dim3(256), 0, st, a.w, w3, a.C, a.K, KW);
        else hipLaunchKernelGGL((x3d_filter_kernel<false>), dim3((unsigned)pnp_cdiv(total, 256)), dim3(256), 0, st, a.w, w3, a.C, a.K, KW);
        PNP_CHECK_LAUNCH("x3d_filter_kernel");
    }
    X3dArgs x{};
    x.x = a.x; x.w3 = w3; x.y = a.y;
    x.N = a.N; x.H = a.H; x.W = a.W; x.C = a.C; x.K = a.K; x.OH = a.OH; x.OW = a.OW; x.pad_t = a.pad_t; x.pad_l = a.pad_l; x.M = a.M;
    x.do_drop = a.do_drop; x.drop_thresh = a.drop_thresh; x.drop_key = a.drop_key; x.drop_keep = a.drop_keep; x.sp = a.sp; x.drop_sid = a.drop_sid;
    x.res_add = a.res_add; x.stat_ws = a.stat_ws; x.stat_shift = a.stat_shift;
    const long long nitems = (long long)a.N * (a.OH / TH) * (a.OW / TW) * (a.K / KW);
    const dim3 grid((unsigned)(nitems > 256 ? 256 : nitems));
    // flops = the bf16 MFMA flops the kernel EXECUTES (six plane products per fp32 multiply-add): its roof is the dense bf16 peak
    const double fl = 6.0 * 2.0 * (double)a.M * 9.0 * a.C * a.K;
    const double by = 4.0 * ((double)a.N * a.H * a.W * a.C + (double)a.M * a.K) + 6.0 * 9.0 * a.C * a.K;
    PnpProfScope ps(cls, st, fl, by, "conv_x3_direct_kernel<%d, %d, %d>", a.C / 32, KW, kind);
#define X3D_LAUNCH(NH_, KW_, KIND_) hipLaunchKernelGGL((conv_x3_direct_kernel<NH_, KW_, KIND_>), grid, dim3(512), 0, st, x)
    if (a.C == 32) {
        if (KW == 32) { if (kind == 0) X3D_LAUNCH(1, 32, 0); else X3D_LAUNCH(1, 32, 1); }
        else { if (kind == 0) X3D_LAUNCH(1, 64, 0); else X3D_LAUNCH(1, 64, 1); }
    } else {
        if (KW == 32) { if (kind == 0) X3D_LAUNCH(2, 32, 0); else X3D_LAUNCH(2, 32, 1); }
        else { if (kind == 0) X3D_LAUNCH(2, 64, 0); else X3D_LAUNCH(2, 64, 1); }
    }
#undef X3D_LAUNCH
    PNP_CHECK_LAUNCH("conv_x3_direct_kernel");
    return PNP_OK;
}

}  // namespace pnpconv

extern "C" int32_t pnp_conv2d_x3_direct(int32_t mode) {
    const int prev = x3d_mode();
    if (mode >= 0) g_x3d_mode.store(mode > 2 ? 2 : mode, std::memory_order_relaxed);
    return prev;
}
