// conv_small.hip — 3x3 stride-1 convolutions with exactly 16 output channels on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
// The 16-channel layers at 256^2 (group_1 of the segmenter, source_segmenter.py:93-96, and the CT copy adapt_1; their data gradients;
// the first conv of group_2 seen from its data gradient) do not fit the 32x32 MFMA tiles of conv_igemm.hip: N = 16 fills half a tile,
// and the reduction (9 taps x 16 channels = 144) is 4.5 stages deep, so a workgroup spends its life in prologue and epilogue.  Round 1
// moved them to the vector ALUs (conv_fwd_narrow_kernel, wgrad_direct_kernel: 37 / 21 TF/s, i.e. 1.0 / 0.6 TB/s effective).  The 16x16x4
// MFMA has the right shape: N = 16 is one tile, the filters (144 x 16 floats) live in 36 VGPRs per lane for the whole kernel, and the
// input patch of an 8x32-pixel output tile is staged ONCE in LDS and read by all 9 taps (im2col-free, no re-fetch per tap).
//
//   conv_n16_kernel<C>      forward / data gradient, C = 16 or 32 input channels: per 16-pixel tile and tap ONE ds_read_b128 feeds 4 MFMAs
//                           (lane (pixel p, quarter g) reads channels 4g..4g+3; MFMA j contracts channels {4g+j}: a k-permutation, legal
//                           because the filter registers use the same pairing).  Pixel stride C+4 floats: 16-byte slot 5p+g (C = 16) /
//                           9p+g (C = 32) is a bijection over each 16-lane group -> conflict-free.  Epilogue = conv_igemm's (dropout,
//                           fused inference BN + shortcut + leaky-ReLU, residual add).
//   wgrad_n16_kernel<C>     filter gradient: dW[(tap, c)][k] = sum_p x[p + tap][c] * dy[p][k] with M = 16 channels of one tap, N = 16
//                           filters, K = 4 pixels per MFMA; x from the same LDS patch, dy straight from memory (64 lanes = 4 pixels x 16
//                           filters = 256 contiguous bytes); per-workgroup partials are summed by splitk_reduce_many_kernel (fixed order).
#include "conv_common.h"

using namespace pnpconv;

namespace {

constexpr int TH = 8, TW = 32;             // output tile of a workgroup; wave w owns rows 2w, 2w+1 = four 16-pixel MFMA tiles
constexpr int PH = TH + 2, PW = TW + 2;    // input patch (3x3, dilation 1)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// input patch of the tile at (n, oh0, ow0) -> LDS [PH*PW pixels][C + 4]; pixels outside the image are zeros (TF zero padding)
template <int C, int LD>
__device__ __forceinline__ void load_patch(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx, float* __restrict__ lds, int n, int oh0, int ow0, int t) {
    constexpr int CQ = C / 4;
    for (int i = t; i < PH * PW * CQ; i += 256) {
        const int pix = i / CQ, cq = i - pix * CQ;
        const int py = pix / PW, px = pix - py * PW;
        const int ih = oh0 - a.pad_t + py, iw = ow0 - a.pad_l + px;
        const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const f32x4 v = bload4(rx, ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * C + 4 * cq) * 4) : OOB);
        *reinterpret_cast<f32x4*>(lds + pix * LD + 4 * cq) = v;
    }
}

// ---- forward / data gradient: y[n, oh, ow, 0..15] -------------------------------------------------------------------------------------
// grid = (column-tile groups, row tiles, images); a workgroup walks `tiles_per_wg` consecutive 32-pixel column tiles of its row band so that
// the filter registers (and the launch) are paid once per several tiles.  KIND only names the launch for the profiler (0 fwd, 1 dgrad).
template <int C, int KIND>
__global__ void __launch_bounds__(256) conv_n16_kernel(ConvArgs a, int tiles_w, int tiles_per_wg) {
    constexpr int LD = C + 4, NH = C / 16;
    __shared__ __attribute__((aligned(16))) float lds[PH * PW * LD];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.z, oh0 = blockIdx.y * TH;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);

    // filters: lane (g, k = p) holds w[tap][16h + 4g + j][k] for every (tap, h, j)
    float wreg[9][NH][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int j = 0; j < 4; ++j) wreg[tap][h][j] = a.w[(tap * C + 16 * h + 4 * g + j) * 16 + p];

    const int jt0 = blockIdx.x * tiles_per_wg;
    for (int jt = jt0; jt < jt0 + tiles_per_wg && jt < tiles_w; ++jt) {
        const int ow0 = jt * TW;
        __syncthreads();                                   // the previous tile's fragment reads are done
        load_patch<C, LD>(a, rx, lds, n, oh0, ow0, t);
        __syncthreads();
        f32x4 acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int r = tap / 3, s = tap % 3;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                  // tile q: row 2*wave + q/2, columns 16*(q%2) .. +15
                const int pix = (2 * wave + (q >> 1) + r) * PW + 16 * (q & 1) + p + s;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(lds + pix * LD + 16 * h + 4 * g);
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[q] = mfma16(av[j], wreg[tap][h][j], acc[q]);
                }
            }
        }
        // epilogue: lane holds rows (pixels) 4g+i, column (filter) p of each tile.  Phases like conv_epilogue (conv_common.h): everything
        // that reads memory first, then stores only — interleaved, every store was fenced with vmcnt(0) against the next value's loads
        float ov[4][4];
        size_t oidx[4][4];
        bool ook[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int oh = oh0 + 2 * wave + (q >> 1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ow = ow0 + 16 * (q & 1) + 4 * g + i;
                ook[q][i] = oh < a.OH && ow < a.OW;
                const int m = ook[q][i] ? (n * a.OH + oh) * a.OW + ow : 0;
                oidx[q][i] = (size_t)m * 16 + p;
                float v = acc[q][i];
                if (a.do_drop) v = pnp_drop_keep((uint32_t)oidx[q][i], pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid), a.drop_thresh) ? v / a.drop_keep : 0.f;
                if (a.res_add) v += a.res_add[oidx[q][i]];
                if (a.ep_scale) v = bn_epilogue(a, v, m, p);
                ov[q][i] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ook[q][i]) a.y[oidx[q][i]] = ov[q][i];
    }
}

// ---- filter gradient ---------------------------------------------------------------------------------------------------------------------
// a.x = x [N,H,W,C], a.w = dy [N,OH,OW,K], a.y = partials [gridDim.x][9*C][K]; blockIdx.y = 16-filter group.  A workgroup walks the tiles t = blockIdx.x, +gridDim.x, ...
// of the linearised (image, row band, column tile) list; its four waves each own 2 rows of a tile and are summed through LDS at the end.
template <int C>
__global__ void __launch_bounds__(256) wgrad_n16_kernel(ConvArgs a, int tiles_w, int tiles_h, int ntiles) {
    // pixel stride of the patch: the A fragments are ds_read_b32 of 16 consecutive channels x 4 consecutive pixels: 16g + p (C = 16) and
    // 48g + p (C = 32) hit 64 distinct banks
    constexpr int LD = (C == 16) ? 16 : 48, NC = C / 16, NT = 9 * NC;      // NT accumulator tiles (tap, 16-channel group) of 16 x 16
    constexpr int LDSF = PH * PW * LD > NT * 256 ? PH * PW * LD : NT * 256;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.w, a.w_bytes);
    const int kg0 = blockIdx.y * 16;         // this workgroup's 16 filters (K = 16, 32, 64: the input patch is re-staged per filter group)
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int jt = tile % tiles_w, rest = tile / tiles_w;
        const int it = rest % tiles_h, n = rest / tiles_h;
        const int oh0 = it * TH, ow0 = jt * TW;
        __syncthreads();
        load_patch<C, LD>(a, rx, lds, n, oh0, ow0, t);
        __syncthreads();
        // 16 groups of 4 consecutive pixels per wave: row 2*wave + q/8, columns 4*(q%8) .. +3; lane quarter g = pixel within the group
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int row = 2 * wave + (q >> 3), col = 4 * (q & 7) + g;
            const int oh = oh0 + row, ow = ow0 + col;
            // B operand: dy[pixel g of the group][filter p]; pixels outside the image contribute zeros
            const bool ok = (oh < a.OH) & (ow < a.OW);
            const float dyv = bload1(rdy, ok ? (unsigned)((((n * a.OH + oh) * a.OW + ow) * a.K + kg0 + p) * 4) : OOB);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int pix = (row + tap / 3) * PW + col + tap % 3;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float xv = lds[pix * LD + 16 * c + p];      // A operand: x[pixel g + tap shift][channel 16c + p]
                    acc[tap * NC + c] = mfma16(xv, dyv, acc[tap * NC + c]);
                }
            }
        }
    }
    // the four waves' accumulators are summed through LDS one wave after the other (fixed order), then one partial per workgroup:
    // lane (g, p) of tile (tap, c) holds rows m' = tap*C + 16c + 4g + k (k = 0..3), column p
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                f32x4* slot = reinterpret_cast<f32x4*>(lds + (i * 64 + lane) * 4);
                if (w == 0) *slot = acc[i];
                else {
                    f32x4 v = *slot;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += acc[i][k];
                    *slot = v;
                }
            }
        }
    }
    __syncthreads();
    float* out = a.y + (size_t)blockIdx.x * (9 * C * a.K) + kg0;
    for (int e = t; e < NT * 64; e += 256) {
        const int i = e >> 6, l = e & 63;
        const f32x4 s = *reinterpret_cast<const f32x4*>(lds + (i * 64 + l) * 4);
        const int tap = i / NC, c = i - tap * NC, lp = l & 15, lg = l >> 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(size_t)(tap * C + 16 * c + 4 * lg + k) * a.K + lp] = s[k];
    }
}

// ---- the first layer: 3 input channels (source_segmenter.py:93, adversarial.py:132) -----------------------------------------------------
// Reduction 27 = 9 taps x 3 channels: 7 MFMA k-steps of 4 (the 28th product is a zero filter row).  The patch keeps 4 floats per pixel
// (the image's 3 + a zero); k index kk = 4*step + g -> (tap, channel) = (kk / 3, kk % 3), i.e. each lane quarter reads its own tap: one
// ds_read_b32 per MFMA at a per-lane offset fixed for the whole kernel.  28 MFMAs per 64 pixels: the kernel is a streaming pass (12.6 MB
// in, 67 MB out at 256^2, B = 16), which is the point — the 32x32-tile kernel spent 0.13 ms on it (7 TF/s).
__device__ __forceinline__ void load_patch3(const ConvArgs& a, __amdgpu_buffer_rsrc_t rx, float* __restrict__ lds, int n, int oh0, int ow0, int t) {
    for (int i = t; i < PH * PW; i += 256) {
        const int py = i / PW, px = i - py * PW;
        const int ih = oh0 - a.pad_t + py, iw = ow0 - a.pad_l + px;
        const bool ok = ((unsigned)ih < (unsigned)a.H) & ((unsigned)iw < (unsigned)a.W);
        const unsigned off = ok ? (unsigned)((((n * a.H + ih) * a.W + iw) * 3) * 4) : OOB;
        f32x4 v;
        v[0] = bload1(rx, off);
        v[1] = bload1(rx, ok ? off + 4u : OOB);
        v[2] = bload1(rx, ok ? off + 8u : OOB);
        v[3] = 0.f;
        *reinterpret_cast<f32x4*>(lds + i * 4) = v;
    }
}

template <int KIND>
__global__ void __launch_bounds__(256) conv_c3n16_kernel(ConvArgs a, int tiles_w, int tiles_per_wg) {
    __shared__ __attribute__((aligned(16))) float lds[PH * PW * 4];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 15, g = lane >> 4;
    const int n = blockIdx.z, oh0 = blockIdx.y * TH;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    float wreg[7];
    int aoff[7];                       // LDS float offset of (tap, channel) of this lane's k index in step i
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int kk = 4 * i + g;
        const int tap = kk < 27 ? kk / 3 : 0, c = kk < 27 ? kk - 3 * (kk / 3) : 3;      // kk = 27: the zero column of pixel 0, zero filter
        wreg[i] = kk < 27 ? a.w[kk * 16 + p] : 0.f;
        aoff[i] = ((tap / 3) * PW + tap % 3) * 4 + c;
    }
    const int jt0 = blockIdx.x * tiles_per_wg;
    for (int jt = jt0; jt < jt0 + tiles_per_wg && jt < tiles_w; ++jt) {
        const int ow0 = jt * TW;
        __syncthreads();
        load_patch3(a, rx, lds, n, oh0, ow0, t);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int base = ((2 * wave + (q >> 1)) * PW + 16 * (q & 1) + p) * 4;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 7; ++i) acc = mfma16(lds[base + aoff[i]], wreg[i], acc);
            const int oh = oh0 + 2 * wave + (q >> 1);
            float ov[4];
            size_t oidx[4];
            bool ook[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {           // (loads first, then stores only: see conv_n16_kernel)
                const int ow = ow0 + 16 * (q & 1) + 4 * g + i;
                ook[i] = oh < a.OH && ow < a.OW;
                const int m = ook[i] ? (n * a.OH + oh) * a.OW + ow : 0;
                oidx[i] = (size_t)m * 16 + p;
                float v = acc[i];
                if (a.do_drop) v = pnp_drop_keep((uint32_t)oidx[i], pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid), a.drop_thresh) ? v / a.drop_keep : 0.f;
                if (a.res_add) v += a.res_add[oidx[i]];
                if (a.ep_scale) v = bn_epilogue(a, v, m, p);
                ov[i] = v;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ook[i]) a.y[oidx[i]] = ov[i];
        }
    }
}

// filter gradient of the first layer: dW[27][16]; rows m' = (tap, channel) = 16T + p for T = 0, 1 (rows 27..31 are padding, never written)
__global__ void __launch_bounds__(256) wgrad_c3n16_kernel(ConvArgs a, int tiles_w, int tiles_h, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[PH * PW * 4 > 2 * 256 ? PH * PW * 4 : 2 * 256];
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int p = lane & 15, g = lane >> 4;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const __amdgpu_buffer_rsrc_t rdy = make_rsrc(a.w, a.w_bytes);
    int aoff[2];
#pragma unroll
    for (int T = 0; T < 2; ++T) {
        const int mm = 16 * T + p;
        const int tap = mm < 27 ? mm / 3 : 0, c = mm < 27 ? mm - 3 * (mm / 3) : 3;
        aoff[T] = ((tap / 3) * PW + tap % 3) * 4 + c;
    }
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int jt = tile % tiles_w, rest = tile / tiles_w;
        const int it = rest % tiles_h, n = rest / tiles_h;
        const int oh0 = it * TH, ow0 = jt * TW;
        __syncthreads();
        load_patch3(a, rx, lds, n, oh0, ow0, t);
        __syncthreads();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int row = 2 * wave + (q >> 3), col = 4 * (q & 7) + g;
            const int oh = oh0 + row, ow = ow0 + col;
            const bool ok = (oh < a.OH) & (ow < a.OW);
            const float dyv = bload1(rdy, ok ? (unsigned)((((n * a.OH + oh) * a.OW + ow) * 16 + p) * 4) : OOB);
            const int base = (row * PW + col) * 4;
            acc[0] = mfma16(lds[base + aoff[0]], dyv, acc[0]);
            acc[1] = mfma16(lds[base + aoff[1]], dyv, acc[1]);
        }
    }
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                f32x4* slot = reinterpret_cast<f32x4*>(lds + (i * 64 + lane) * 4);
                if (w == 0) *slot = acc[i];
                else {
                    f32x4 v = *slot;
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] += acc[i][k];
                    *slot = v;
                }
            }
        }
    }
    __syncthreads();
    float* out = a.y + (size_t)blockIdx.x * (27 * 16);
    if (t < 128) {
        const int i = t >> 6, l = t & 63;
        const f32x4 s = *reinterpret_cast<const f32x4*>(lds + (i * 64 + l) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int mm = 16 * i + 4 * (l >> 4) + k;
            if (mm < 27) out[mm * 16 + (l & 15)] = s[k];
        }
    }
}

}  // namespace

namespace pnpconv {

bool n16_geom_ok(const pnp_conv_geom* g) {
    static const int off = getenv("PNP_CONV_NON16") ? 1 : 0;
    if (off || g->K != 16 || (g->C != 16 && g->C != 32 && g->C != 3) || g->R != 3 || g->S != 3 || g->stride != 1 || g->dil != 1) return false;
    // dtype: these kernels compute in fp32 whatever the geometry allows — PNP_DTYPE_BF16 PERMITS bf16 operands (conv_bf16.hip's header), and
    // on 16-channel layers the fp32 16x16x4 tiles are both exact and faster than the 32-wide bf16 tiles (B = 16: 16->16 0.072 vs 0.131 ms)
    static const int f32only = getenv("PNP_N16_F32ONLY") ? 1 : 0;                                 // A/B: round 2's routing
    if (g->pad_mode != PNP_PAD_ZERO || g->pad_t > 1 || g->pad_l > 1 || (f32only && g->dtype != PNP_DTYPE_F32)) return false;
    if (g->OH != g->H + 2 * g->pad_t - 2 || g->OW != g->W + 2 * g->pad_l - 2) return false;
    return (long long)g->N * g->OH * g->OW >= 8192 && g->N <= 65535;
}

int launch_n16_fwd(const ConvArgs& a, int kind, hipStream_t st) {
    const int tiles_w = pnp_cdiv(a.OW, TW), tiles_h = pnp_cdiv(a.OH, TH);
    // ~8 workgroup slots per CU x 256 CUs: walk several column tiles per workgroup once there are more tiles than that
    int tpw = 1;
    while (tpw < tiles_w && (long long)pnp_cdiv(tiles_w, tpw) * tiles_h * a.N > 4096) tpw *= 2;
    dim3 grid((unsigned)pnp_cdiv(tiles_w, tpw), (unsigned)tiles_h, (unsigned)a.N);
    PnpProfScope ps(kind == 0 ? PNP_PROF_CONV_FWD : PNP_PROF_CONV_DGRAD, st, conv_flops(a), conv_bytes(a), "%s<%d, %d>",
                    a.C == 3 ? "conv_c3n16_kernel" : "conv_n16_kernel", a.C, kind);
    if (a.C == 3 && kind == 0) hipLaunchKernelGGL((conv_c3n16_kernel<0>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    else if (a.C == 3) hipLaunchKernelGGL((conv_c3n16_kernel<1>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    else if (a.C == 16 && kind == 0) hipLaunchKernelGGL((conv_n16_kernel<16, 0>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    else if (a.C == 16) hipLaunchKernelGGL((conv_n16_kernel<16, 1>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    else if (kind == 0) hipLaunchKernelGGL((conv_n16_kernel<32, 0>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    else hipLaunchKernelGGL((conv_n16_kernel<32, 1>), grid, dim3(256), 0, st, a, tiles_w, tpw);
    PNP_CHECK_LAUNCH("conv_n16_kernel");
    return PNP_OK;
}

// filter gradient: also 32 / 64 filters (one workgroup column per 16 of them) when the input has 16 or 32 channels — group_2's and
// group_3's first convolutions, whose 9*C-row gradients are too small for the 128-row tiles of conv_wgrad_kernel (20-30 TF/s there)
bool n16_wgrad_ok(const pnp_conv_geom* g) {
    if (n16_geom_ok(g)) return true;
    static const int off = getenv("PNP_CONV_NON16") ? 1 : 0;
    static const int maxk = getenv("PNP_N16W_MAXK") ? atoi(getenv("PNP_N16W_MAXK")) : 64;       // A/B against the ring kernel's 128x64 tiles
    if (off || g->K > maxk || (g->K != 32 && g->K != 64) || (g->C != 16 && g->C != 32) || g->R != 3 || g->S != 3 || g->stride != 1 || g->dil != 1) return false;
    static const int f32only = getenv("PNP_N16_F32ONLY") ? 1 : 0;
    if (g->pad_mode != PNP_PAD_ZERO || g->pad_t > 1 || g->pad_l > 1 || (f32only && g->dtype != PNP_DTYPE_F32)) return false;
    if (g->OH != g->H + 2 * g->pad_t - 2 || g->OW != g->W + 2 * g->pad_l - 2) return false;
    // 64 filters over >= 2^19 pixels (cls_1's 32 -> 64 at 256^2): the ring kernel's 128x64 tiles win there (measured B = 16: 0.728 ms here,
    // 0.583 ms on conv_wgrad_ring_kernel); on the 64^2 layer of the same shape this kernel wins 0.054 vs 0.106
    if (g->K == 64 && (long long)g->N * g->OH * g->OW >= (1ll << 19)) return false;
    return (long long)g->N * g->OH * g->OW >= 8192;
}

int n16_wgrad_blocks(const pnp_conv_geom* g) {
    const long long ntiles = (long long)pnp_cdiv(g->OW, TW) * pnp_cdiv(g->OH, TH) * g->N;
    long long nb = ntiles < 1024 ? ntiles : 1024;
    const long long cap = (8ll << 20) / ((long long)9 * g->C * g->K);          // partials of at most 32 MB
    if (nb > cap) nb = cap;
    return (int)(nb < 1 ? 1 : nb);
}

// partials: [n16_wgrad_blocks][9*C][16] floats in `part`
int launch_n16_wgrad(const ConvArgs& a, float* part, hipStream_t st) {
    const int tiles_w = pnp_cdiv(a.OW, TW), tiles_h = pnp_cdiv(a.OH, TH);
    const long long ntiles = (long long)tiles_w * tiles_h * a.N;
    long long nb = ntiles < 1024 ? ntiles : 1024;
    const long long cap = (8ll << 20) / ((long long)9 * a.C * a.K);
    if (nb > cap) nb = cap;
    const int nblk = (int)(nb < 1 ? 1 : nb);
    ConvArgs b = a;
    b.y = part;
    PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, conv_flops(a), conv_bytes(a), "%s<%d>", a.C == 3 ? "wgrad_c3n16_kernel" : "wgrad_n16_kernel", a.C);
    if (a.C == 3) hipLaunchKernelGGL(wgrad_c3n16_kernel, dim3((unsigned)nblk), dim3(256), 0, st, b, tiles_w, tiles_h, (int)ntiles);
    else if (a.C == 16) hipLaunchKernelGGL((wgrad_n16_kernel<16>), dim3((unsigned)nblk, (unsigned)(a.K / 16)), dim3(256), 0, st, b, tiles_w, tiles_h, (int)ntiles);
    else hipLaunchKernelGGL((wgrad_n16_kernel<32>), dim3((unsigned)nblk, (unsigned)(a.K / 16)), dim3(256), 0, st, b, tiles_w, tiles_h, (int)ntiles);
    PNP_CHECK_LAUNCH("wgrad_n16_kernel");
    return PNP_OK;
}

}  // namespace pnpconv
