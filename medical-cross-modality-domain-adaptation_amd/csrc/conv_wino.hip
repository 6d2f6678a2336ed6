// conv_wino.hip — Winograd F(2x2, 3x3) route for the wide stride-1 3x3 convolutions (forward, data gradient, filter gradient) on fp32
// matrix cores.
//
// Replaces, for the layers the planner picks, the direct implicit GEMM of conv_igemm.hip behind the same entry points
// (tf.nn.conv2d / tf.nn.atrous_conv2d at /root/reference/layers.py:18,24,67,73,86,92 and their TF-autodiff data gradients): the
// segmenter's 256- and 512-channel groups (source_segmenter.py:140-200: g5-g10, dilation 2 in g8) and the critics' widest stride-1 layers
// (adversarial.py:337-400) are 84 % of the step's FLOPs and run at 0.91-0.92 of the fp32 MFMA peak — the matrix pipe is the roof, so the
// remaining lever is fewer multiplications.  The minimal filtering algorithm F(2x2, 3x3) (Lavin & Gray 2016) computes a 2x2 output tile
// from a 4x4 input patch with 16 multiplications per (channel, filter) pair instead of 36: 2.25x fewer MFMA flops.
//
//   V[pos][t][c] = (B^T d B)[pos]     wino_in_kernel      4x4 patch of tile t, channel c   (adds only)
//   U[pos][c][k] = (G g G^T)[pos]     wino_filter_kernel  (data gradient: of the flipped, transposed filter — no separate flip launch)
//   M[pos]       = V[pos] x U[pos]    wino_gemm_kernel    16 independent GEMMs [T x C] x [C x K] in ONE launch, v_mfma_f32_32x32x2_f32,
//                                                         the 128x128 tile / 4-slice register pipeline of conv_taps_kernel
//   y tile       = A^T M A            wino_out_kernel     + the whole convolution epilogue: dropout, residual add, BN statistics partials,
//                                                         fused inference BN + shortcut + leaky-ReLU (conv_common.h: conv_epilogue's order)
// (the filter gradient is the transposed algorithm on the same V: second half of this file.)
// pos = 4 i + j indexes the 16 points of the transformed 4x4 tile.  V and M pass through HBM (workspace): 4x the input / output size
// each, which is why the route only pays where the contraction is deep — the planner takes it when C K / (C + K) >= 85 (128->256 up).
// A dilation-d SAME convolution is d x d independent dense convolutions of the sub-images (a + d u, b + d v): the tile enumeration
// walks (image, a, b, tile row, tile column), everything else is unchanged (oracle/tf_ops.py::conv3x3_winograd_np restates the index
// arithmetic; tests/test_host.py holds it to the direct convolution).
//
// Arithmetic: every product and sum in fp32; the transforms add a rounding of ~1 ulp per element in front of and behind the
// contraction (measured against float64: same 1e-6 level as the direct kernel, tests/test_gpu_wino.py), inside north_star's 1e-4.
#include <atomic>
#include "conv_common.h"
#include "conv_mma.h"

using namespace pnpconv;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// Hi x Wi: the input image, Ho x Wo = Hi + 2 pad - 2 dil: the output; ps = pad / dil in {0, 1, 2} (VALID on a pre-padded image, SAME, the
// data gradient of a VALID convolution); Hsi / Hso ...: extents of one dilation phase's sub-image; tiles enumerate the OUTPUT
struct WinoGeom {
    int N, Hi, Wi, Ho, Wo, dil, ps, Hsi, Wsi, Hso, Wso, th, tw, T;
};

// tile id -> (image, phase a, phase b, tile row, tile column): t = (((n d + a) d + b) th + ti) tw + tj
__device__ __forceinline__ void tile_of(const WinoGeom& g, int t, int& n, int& a, int& b, int& ti, int& tj) {
    tj = t % g.tw;
    t /= g.tw;
    ti = t % g.th;
    t /= g.th;
    b = t % g.dil;
    t /= g.dil;
    a = t % g.dil;
    n = t / g.dil;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// input transform: one thread = one tile x 4 channels.  16 loads of 16 B (buffer descriptor: a patch pixel outside the image gets an
// out-of-range offset and reads 0), 32 + 32 vector adds per channel quad, 16 stores of 16 B (one per transform point, C contiguous).
struct WinoInArgs {
    const float* x;
    float* V;
    WinoGeom g;
    int C;
    unsigned x_bytes;
};

__global__ void __launch_bounds__(NT) wino_in_kernel(WinoInArgs a) {
    const int C4 = a.C >> 2;
    const size_t nvec = (size_t)a.g.T * C4;
    const size_t plane = (size_t)a.g.T * a.C;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const size_t gs = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += gs) {
        const int t = (int)(i / C4);
        const int c = (int)(i - (size_t)t * C4) * 4;
        int n, pa, pb, ti, tj;
        tile_of(a.g, t, n, pa, pb, ti, tj);
        f32x4 d[4][4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int u = 2 * ti - a.g.ps + p;
            const bool uok = (unsigned)u < (unsigned)a.g.Hsi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int v = 2 * tj - a.g.ps + q;
                const bool ok = uok & ((unsigned)v < (unsigned)a.g.Wsi);
                const unsigned off = (unsigned)(((n * a.g.Hi + pa + a.g.dil * u) * a.g.Wi + pb + a.g.dil * v) * a.C + c) * 4u;    // (host: < 2^30 elements)
                d[p][q] = bload4(rx, ok ? off : OOB);
            }
        }
        // rows (B^T d), then columns ((B^T d) B)
        f32x4 r[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[0][q] = d[0][q] - d[2][q];
            r[1][q] = d[1][q] + d[2][q];
            r[2][q] = d[2][q] - d[1][q];
            r[3][q] = d[1][q] - d[3][q];
        }
        float* out = a.V + (size_t)t * a.C + c;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            st4(out + (size_t)(4 * p + 0) * plane, r[p][0] - r[p][2]);
            st4(out + (size_t)(4 * p + 1) * plane, r[p][1] + r[p][2]);
            st4(out + (size_t)(4 * p + 2) * plane, r[p][2] - r[p][1]);
            st4(out + (size_t)(4 * p + 3) * plane, r[p][1] - r[p][3]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// filter transform: U[pos][row][col] = (G g G^T)[pos], rows = reduction channels, cols = output channels of the GEMM.
//   TRANS false (forward):        g[r][s] = w[r][s][row][col]
//   TRANS true  (data gradient):  g[r][s] = w[2-r][2-s][col][row]   (the flipped, transposed filter; 32x32 tiles through LDS so that both
//                                 the read along w's last axis and the write along U's last axis are coalesced)
template <bool TRANS>
__global__ void __launch_bounds__(NT) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int rows, int cols) {
    __shared__ float tile[TRANS ? 9 : 1][32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const size_t plane = (size_t)rows * cols;
    if constexpr (TRANS) {
        // w is [tap][cols][rows] here
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            for (int j = ty; j < 32; j += 8) {
                const int sc = c0 + j, sr = r0 + tx;
                tile[tap][j][tx] = (sc < cols && sr < rows) ? w[((size_t)tap * cols + sc) * rows + sr] : 0.f;
            }
        __syncthreads();
    }
    for (int j = ty; j < 32; j += 8) {
        const int row = r0 + j, col = c0 + tx;
        const bool ok = row < rows && col < cols;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if constexpr (TRANS) g[r][s] = tile[(2 - r) * 3 + (2 - s)][tx][j];
                else g[r][s] = ok ? w[((size_t)(r * 3 + s) * rows + row) * cols + col] : 0.f;
            }
        float t[4][3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            t[0][s] = g[0][s];
            t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
            t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
            t[3][s] = g[2][s];
        }
        if (ok) {
            float* out = U + (size_t)row * cols + col;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                out[(size_t)(4 * i + 0) * plane] = t[i][0];
                out[(size_t)(4 * i + 1) * plane] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
                out[(size_t)(4 * i + 2) * plane] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
                out[(size_t)(4 * i + 3) * plane] = t[i][2];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// the 16 GEMMs M[pos] = V[pos] x U[pos] in one launch.  Workgroup = one 128 x 128 tile of one transform point; stage loop, LDS layout,
// fragment pipeline and store schedule are conv_taps_kernel's with a single tap (conv_igemm.hip): A rows need no validity masks beyond
// the ragged last tile (out-of-range buffer offset = 0), B rows are U's rows.  KIND only names the symbol (forward / data gradient).
struct WinoGemmArgs {
    const float* V;
    const float* U;
    float* Mm;
    int T, C, K;
    int nblk_m, nblk_n, gn, xcd_swizzle;
};

template <int BM, int BN, int WM, int WN, int KIND>
__global__ void __launch_bounds__(NTHREADS, 2) wino_gemm_kernel(WinoGemmArgs g) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BK + 4, LDB = BN + 4;
    constexpr int ASZ = BM * LDA, BSZ = BK * LDB;
    constexpr int NR = BM / 32;
    constexpr int C4 = BN / 4, RPB = NTHREADS / C4, NPB = BK / RPB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = g.nblk_m * g.nblk_n;
    const int pos = blockIdx.x / nblk;
    int bid = blockIdx.x - pos * nblk;
    if (g.xcd_swizzle) bid = xcd_remap(bid, nblk);
    int mt, nt;
    tile_coords(bid, g.nblk_m, g.nblk_n, g.gn, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    const float* Ap = g.V + (size_t)pos * g.T * g.C;
    const float* Bp = g.U + (size_t)pos * g.C * g.K;
    float* Op = g.Mm + (size_t)pos * g.T * g.K;

    const int kg = t & 7, mrow = t >> 3;
    unsigned abase[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int m = m0 + mrow + 32 * i;
        abase[i] = (m < g.T) ? (unsigned)((m * g.C + 4 * kg) * 4) : OOB2;
    }
    const int bcol = t % C4, brow = t / C4;
    unsigned boff[NPB];
#pragma unroll
    for (int i = 0; i < NPB; ++i)
        boff[i] = (n0 + 4 * bcol < g.K) ? (unsigned)(((brow + RPB * i) * g.K + n0 + 4 * bcol) * 4) : OOB2;

    const __amdgpu_buffer_rsrc_t rx = make_rsrc(Ap, (unsigned)((size_t)g.T * g.C * 4));
    const __amdgpu_buffer_rsrc_t rw = make_rsrc(Bp, (unsigned)((size_t)g.C * g.K * 4));
    f32x4 areg[NR], breg[NPB];
    auto gload = [&](int cc) {
        const int sa = cc * (BK * 4);
        const int sb = (cc * BK * g.K) * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) areg[i] = bload4s(rx, abase[i], sa);
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
    const int ncc = g.C / BK;

    gload(0);
    lstore(lds, lds + 2 * ASZ);
    __syncthreads();
    Frag<TM, TN, true, LDA, LDB> f0, f1;
    f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
    for (int cc = 0; cc < ncc; ++cc) {
        const int cur = cc & 1;
        const float* As = lds + cur * ASZ;
        const float* Bs = lds + 2 * ASZ + cur * BSZ;
        float* An = lds + (cur ^ 1) * ASZ;
        float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
        const int nxt = (cc + 1 < ncc) ? cc + 1 : cc;        // past the end: the last group again (valid addresses, a buffer nobody reads)
        // ---- slice 0
        f1.load(As, Bs, 1, wm0, wn0, lane);
        gload(nxt);
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
        f0.load(An, Bn, 0, wm0, wn0, lane);
    }

    // plain [T][K] rows of this transform point: conv_epilogue with every feature off
    ConvArgs e{};
    e.M = g.T;
    e.K = g.K;
    e.nsplit = 1;
    conv_epilogue<TM, TN>(e, acc, Op, m0, n0, wm0, wn0, lane, 0);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// output transform + the convolution epilogue.  One thread = one tile x 4 output channels: 16 loads of 16 B, 24 + 12 vector adds per
// channel quad, then per output pixel of the 2x2 tile (those inside the image) the epilogue in conv_epilogue's order:
// dropout (counter hash on the flat output index) -> residual add -> BN statistics partials -> fused inference BN (+ shortcut, channel
// zero-padded) + leaky-ReLU -> one 16-B store.  Workgroup (x, y) = tile slab x, channel slice y (256 channel quads): statistics partials
// of slab x go to stat_ws[(x*2 + q)*K + k] (reduced over the workgroup's tiles through LDS, fixed order: deterministic).
struct WinoOutArgs {
    const float* Mm;
    float* y;
    WinoGeom g;
    int K;
    int tiles_per_block;
    int do_drop;
    uint32_t drop_thresh, drop_key;
    float drop_scale;
    const pnp_step_params* sp;
    uint32_t drop_sid;
    const float* res_add;
    float* stat_ws;
    const float* stat_shift;
    const float* ep_scale;
    const float* ep_shift;
    const float* ep_res;
    int ep_cs;
    float ep_alpha;
};

__global__ void __launch_bounds__(NT) wino_out_kernel(WinoOutArgs a) {
    __shared__ float red[NT * 8];
    const int K4 = a.K >> 2;
    const int kq0 = blockIdx.y * NT;                               // first channel quad of this slice
    const int K4s = (K4 - kq0) < NT ? (K4 - kq0) : NT;             // channel quads in this slice
    const int rpi = NT / K4s;                                      // tiles per iteration
    const int tid = threadIdx.x;
    const int cg = tid % K4s, rsub = tid / K4s;
    const bool active = rsub < rpi;
    const int k = (kq0 + cg) * 4;
    const int t0 = blockIdx.x * a.tiles_per_block;
    int t1 = t0 + a.tiles_per_block;
    if (t1 > a.g.T) t1 = a.g.T;
    const size_t plane = (size_t)a.g.T * a.K;
    const uint32_t dkey = a.do_drop ? pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid) : 0u;
    const bool stats = a.stat_ws != nullptr;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (active) {
        f32x4 shift = {0, 0, 0, 0}, sc = {0, 0, 0, 0}, sh = {0, 0, 0, 0};
        if (stats && a.stat_shift) shift = ld4(a.stat_shift + k);
        if (a.ep_scale) {
            sc = ld4(a.ep_scale + k);
            sh = ld4(a.ep_shift + k);
        }
        const int cpad = (a.K - a.ep_cs) >> 1;
        for (int t = t0 + rsub; t < t1; t += rpi) {
            int n, pa, pb, ti, tj;
            tile_of(a.g, t, n, pa, pb, ti, tj);
            const float* src = a.Mm + (size_t)t * a.K + k;
            f32x4 m[4][4];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int q = 0; q < 4; ++q) m[p][q] = ld4(src + (size_t)(4 * p + q) * plane);
            // rows (A^T m), then columns ((A^T m) A)
            f32x4 r[2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[0][q] = m[0][q] + m[1][q] + m[2][q];
                r[1][q] = m[1][q] - m[2][q] - m[3][q];
            }
            f32x4 o[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                o[p][0] = r[p][0] + r[p][1] + r[p][2];
                o[p][1] = r[p][1] - r[p][2] - r[p][3];
            }
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int u = 2 * ti + p, v = 2 * tj + q;
                    if (u >= a.g.Hso || v >= a.g.Wso) continue;
                    const size_t row = (size_t)(n * a.g.Ho + pa + a.g.dil * u) * a.g.Wo + pb + a.g.dil * v;
                    const size_t idx = row * a.K + k;
                    f32x4 val = o[p][q];
                    if (a.do_drop) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            val[e] = pnp_drop_keep((uint32_t)(idx + e), dkey, a.drop_thresh) ? val[e] * a.drop_scale : 0.f;
                    }
                    if (a.res_add) val += ld4(a.res_add + idx);
                    if (stats) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float d = val[e] - shift[e];
                            s0[e] += d;
                            s1[e] = fmaf(d, d, s1[e]);
                        }
                    }
                    if (a.ep_scale) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) val[e] = fmaf(val[e], sc[e], sh[e]);
                        if (a.ep_res) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int cs = k + e - cpad;
                                if ((unsigned)cs < (unsigned)a.ep_cs) val[e] += a.ep_res[row * a.ep_cs + cs];
                            }
                        }
                        if (a.ep_alpha >= 0.f) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) val[e] = val[e] < 0.f ? val[e] * a.ep_alpha : val[e];
                        }
                    }
                    st4(a.y + idx, val);
                }
        }
    }
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[tid * 8 + e] = s0[e];
        red[tid * 8 + 4 + e] = s1[e];
    }
    __syncthreads();
    if (tid < K4s) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int j = 0; j < rpi; ++j) {
            const int tt = j * K4s + tid;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] += red[tt * 8 + e];
                a1[e] += red[tt * 8 + 4 + e];
            }
        }
        st4(a.stat_ws + ((size_t)blockIdx.x * 2 + 0) * a.K + k, a0);
        st4(a.stat_ws + ((size_t)blockIdx.x * 2 + 1) * a.K + k, a1);
    }
}


// =================================================================================================================================
// Filter gradient on the route: dW = G^T [ sum over tiles of (B^T d B) (.) (A y A^T) ] G  — the transposition of F(2x2, 3x3): the SAME
// input transform V as the forward, the 2x2 tile of dy spread to the 4x4 transform points (A = (A^T)^T: z0 = y0, z1 = y0 + y1,
// z2 = y0 - y1, z3 = -y1 along each axis), 16 GEMMs S[pos] = V[pos]^T x Y[pos] ([C x T] x [T x K], reduction over the tiles, split
// across workgroups like every filter gradient here) and a 4x4 -> 3x3 transform that also sums the split partials.
struct WinoDyArgs {
    const float* dy;
    float* Y;
    WinoGeom g;
    int K;
};

__global__ void __launch_bounds__(NT) wino_dy_kernel(WinoDyArgs a) {
    const int K4 = a.K >> 2;
    const size_t nvec = (size_t)a.g.T * K4;
    const size_t plane = (size_t)a.g.T * a.K;
    const size_t gs = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += gs) {
        const int t = (int)(i / K4);
        const int k = (int)(i - (size_t)t * K4) * 4;
        int n, pa, pb, ti, tj;
        tile_of(a.g, t, n, pa, pb, ti, tj);
        f32x4 y[2][2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int u = 2 * ti + p, v = 2 * tj + q;
                const f32x4 zero = {0, 0, 0, 0};
                y[p][q] = (u < a.g.Hso && v < a.g.Wso) ? ld4(a.dy + ((size_t)(n * a.g.Ho + pa + a.g.dil * u) * a.g.Wo + pb + a.g.dil * v) * a.K + k) : zero;
            }
        f32x4 z[4][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            z[0][q] = y[0][q];
            z[1][q] = y[0][q] + y[1][q];
            z[2][q] = y[0][q] - y[1][q];
            z[3][q] = -y[1][q];
        }
        float* out = a.Y + (size_t)t * a.K + k;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            st4(out + (size_t)(4 * p + 0) * plane, z[p][0]);
            st4(out + (size_t)(4 * p + 1) * plane, z[p][0] + z[p][1]);
            st4(out + (size_t)(4 * p + 2) * plane, z[p][0] - z[p][1]);
            st4(out + (size_t)(4 * p + 3) * plane, -z[p][1]);
        }
    }
}

// S[z][pos] = V[pos][rows of split z]^T x Y[pos][rows of split z].  Workgroup = one 128 x 128 tile of (C x K) of one transform point and
// one reduction split; both operands are [tile row][channel] with the reduction index as the ROW, so both LDS tiles are [32][128 + 4]
// like the B tile of the convolutions (fragments by ds_read_b32: Frag<.., A_MMAJOR = false>, the layout of conv_wgrad_kernel).
struct WinoWgradGemmArgs {
    const float* V;
    const float* Y;
    float* S;
    int T, C, K;
    int nblk_m, nblk_n, nsplit, chunks_per_split, xcd_swizzle;
};

template <int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(NTHREADS, 2) wino_wgrad_gemm_kernel(WinoWgradGemmArgs g) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    constexpr int ASZ = BK * LDA, BSZ = BK * LDB;
    constexpr int A4 = BM / 4, ARPB = NTHREADS / A4, ANPB = BK / ARPB;        // A tile: 32 rows x BM/4 float4 columns
    constexpr int B4 = BN / 4, BRPB = NTHREADS / B4, BNPB = BK / BRPB;
    __shared__ __attribute__((aligned(16))) float lds[2 * (ASZ + BSZ)];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int nblk = g.nblk_m * g.nblk_n;
    // all (channel, filter) tiles of one (transform point, split) re-read the same rows of V and Y: consecutive logical ids = one XCD's L2
    const int lid = g.xcd_swizzle ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int pz = lid / nblk;
    const int bid = lid - pz * nblk;
    const int pos = pz / g.nsplit, z = pz - pos * g.nsplit;
    const int mt = bid / g.nblk_n, nt = bid - mt * g.nblk_n;
    const int m0 = mt * BM, n0 = nt * BN;
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);

    const float* Ap = g.V + (size_t)pos * g.T * g.C;
    const float* Bp = g.Y + (size_t)pos * g.T * g.K;
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(Ap, (unsigned)((size_t)g.T * g.C * 4));
    const __amdgpu_buffer_rsrc_t rb = make_rsrc(Bp, (unsigned)((size_t)g.T * g.K * 4));

    const int acol = t % A4, arow = t / A4;
    const int bcol = t % B4, brow = t / B4;
    const bool aok = m0 + 4 * acol < g.C, bok = n0 + 4 * bcol < g.K;
    const int r_begin = z * g.chunks_per_split * BK;
    int r_end = r_begin + g.chunks_per_split * BK;
    if (r_end > g.T) r_end = g.T;
    const int nchunks = (r_end - r_begin + BK - 1) / BK;

    f32x4 areg[ANPB], breg[BNPB];
    auto gload = [&](int r0) {               // rows r0 .. r0 + 31 of this split (past r_end: out-of-range offset = zeros)
#pragma unroll
        for (int i = 0; i < ANPB; ++i) {
            const int r = r0 + arow + ARPB * i;
            areg[i] = bload4s(ra, (aok & (r < r_end)) ? (unsigned)((r * g.C + m0 + 4 * acol) * 4) : OOB2, 0);
        }
#pragma unroll
        for (int i = 0; i < BNPB; ++i) {
            const int r = r0 + brow + BRPB * i;
            breg[i] = bload4s(rb, (bok & (r < r_end)) ? (unsigned)((r * g.K + n0 + 4 * bcol) * 4) : OOB2, 0);
        }
    };
    auto lstore = [&](float* An, float* Bn) {
#pragma unroll
        for (int i = 0; i < ANPB; ++i) *reinterpret_cast<f32x4*>(An + (arow + ARPB * i) * LDA + 4 * acol) = areg[i];
#pragma unroll
        for (int i = 0; i < BNPB; ++i) *reinterpret_cast<f32x4*>(Bn + (brow + BRPB * i) * LDB + 4 * bcol) = breg[i];
    };

    Acc<TM, TN> acc;
    acc.zero();
    if (nchunks > 0) {
        gload(r_begin);
        lstore(lds, lds + 2 * ASZ);
        __syncthreads();
        Frag<TM, TN, false, LDA, LDB> f0, f1;
        f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
        for (int c = 0; c < nchunks; ++c) {
            const int cur = c & 1;
            const float* As = lds + cur * ASZ;
            const float* Bs = lds + 2 * ASZ + cur * BSZ;
            float* An = lds + (cur ^ 1) * ASZ;
            float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
            f1.load(As, Bs, 1, wm0, wn0, lane);
            gload(r_begin + (c + 1) * BK);
            f0.mma(acc);
            __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
            for (int i = 0; i < 4 * TM * TN; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x006, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
            PNP_SCHED_FENCE();
            PNP_SLICE(f0.load(As, Bs, 2, wm0, wn0, lane), f1.mma(acc), 4 * TM * TN, (4 * TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            PNP_SLICE2(f1.load(As, Bs, 3, wm0, wn0, lane), f0.mma(acc), lstore(An, Bn), 4 * TM * TN, (4 * TM + 4 * TN + 4 * TM * TN - 1) / (4 * TM * TN))
            PNP_LAST_SLICE(f1.mma(acc), lstore(An, Bn), 4 * TM * TN)
            __syncthreads();
            f0.load(An, Bn, 0, wm0, wn0, lane);
        }
    }
    // S[z][pos] rows = channels, columns = filters: wgrad_epilogue with Kred = C
    ConvArgs e{};
    e.Kred = g.C;
    e.K = g.K;
    wgrad_epilogue<TM, TN>(e, acc, g.S + ((size_t)z * 16 + pos) * g.C * g.K, m0, n0, wm0, wn0, lane);
}

// dW[r][s][c][k] (+)= sum_ij G[i][r] G[j][s] sum_z S[z][4i+j][c][k]: one thread = one channel x 4 filters
__global__ void __launch_bounds__(NT) wino_wgrad_out_kernel(const float* __restrict__ S, float* __restrict__ dw, int C, int K, int nsplit,
                                                            int accumulate) {
    const int K4 = K >> 2;
    const size_t nvec = (size_t)C * K4;
    const size_t plane = (size_t)C * K;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= nvec) return;
    const size_t off = i * 4;                 // (c, k) -> c*K + k
    f32x4 s[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        f32x4 v = ld4(S + (size_t)p * plane + off);
        for (int z = 1; z < nsplit; ++z) v += ld4(S + ((size_t)z * 16 + p) * plane + off);
        s[p] = v;
    }
    f32x4 t[3][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[0][j] = s[j] + 0.5f * (s[4 + j] + s[8 + j]);
        t[1][j] = 0.5f * (s[4 + j] - s[8 + j]);
        t[2][j] = 0.5f * (s[4 + j] + s[8 + j]) + s[12 + j];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        f32x4 o[3];
        o[0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
        o[1] = 0.5f * (t[r][1] - t[r][2]);
        o[2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float* dst = dw + (size_t)(r * 3 + q) * plane + off;
            if (accumulate) o[q] += ld4(dst);
            st4(dst, o[q]);
        }
    }
}

// ------------------------------------------------------------ host side ----------------------------------------------------------
#ifndef PNP_WINOGRAD_DEFAULT
#define PNP_WINOGRAD_DEFAULT 1
#endif
#ifndef PNP_WINOGRAD_WGRAD_DEFAULT
#define PNP_WINOGRAD_WGRAD_DEFAULT 1
#endif
std::atomic<int> g_wino_wgrad_mode{-1};    // the filter gradient's own switch (PNP_WINOGRAD_WGRAD): 0 never / 1 planner / 2 wherever eligible
int wino_wgrad_mode() {
    int m = g_wino_wgrad_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = getenv("PNP_WINOGRAD_WGRAD") ? atoi(getenv("PNP_WINOGRAD_WGRAD")) : PNP_WINOGRAD_WGRAD_DEFAULT;
        if (m < 0) m = 0;
        g_wino_wgrad_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
// reduction split of the filter gradient's GEMMs: enough workgroups for >= 1 dispatch round (512 slots), >= 8 stages each
int wgrad_split(int T, int C, int K, int* chunks_per_split) {
    const int nblk = pnp_cdiv(C, 128) * pnp_cdiv(K, 128) * 16;
    const int nchunks = pnp_cdiv(T, BK);
    int ns = pnp_cdiv(512, nblk);
    if (ns > nchunks / 8) ns = nchunks / 8;
    if (ns < 1) ns = 1;
    *chunks_per_split = pnp_cdiv(nchunks, ns);
    return pnp_cdiv(nchunks, *chunks_per_split);
}

std::atomic<int> g_wino_mode{-1};          // -1: not read yet (environment PNP_WINOGRAD, else the compiled-in default)
int wino_mode() {
    int m = g_wino_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = getenv("PNP_WINOGRAD") ? atoi(getenv("PNP_WINOGRAD")) : PNP_WINOGRAD_DEFAULT;
        if (m < 0) m = 0;
        g_wino_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}

WinoGeom make_wgeom(int N, int Hi, int Wi, int Ho, int Wo, int dil, int pad) {
    WinoGeom g{};
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.dil = dil; g.ps = pad / dil;
    g.Hsi = Hi / dil; g.Wsi = Wi / dil; g.Hso = Ho / dil; g.Wso = Wo / dil;
    g.th = (g.Hso + 1) / 2; g.tw = (g.Wso + 1) / 2;
    g.T = N * dil * dil * g.th * g.tw;
    return g;
}
WinoGeom make_wgeom(const pnp_conv_geom* g) { return make_wgeom(g->N, g->H, g->W, g->OH, g->OW, g->dil, g->pad_t); }

inline size_t al256(size_t b) { return (b + 255) & ~(size_t)255; }

// tile slabs of the output transform (= BN statistics partial rows): ~1024 workgroups per channel slice
void out_plan(int T, int K, int* tpb, int* nblk) {
    const int K4 = K >> 2;
    const int K4s = K4 < NT ? K4 : NT;
    const int rpi = NT / K4s;
    int iters = pnp_cdiv(T, (long long)rpi * 1024);
    if (iters < 1) iters = 1;
    *tpb = rpi * iters;
    *nblk = pnp_cdiv(T, *tpb);
}

}  // namespace

namespace pnpconv {

bool wino_eligible(const pnp_conv_geom* g) {
    if (!g || g->dtype != PNP_DTYPE_F32 || g->R != 3 || g->S != 3 || g->stride != 1 || g->pad_mode != PNP_PAD_ZERO) return false;
    // padding 0 (VALID on a mirror-padded image: g10), dil (SAME) or 2 dil (the data gradient of a VALID convolution), same on both axes
    if (g->dil < 1 || g->dil > 2 || g->pad_t != g->pad_l || (g->pad_t % g->dil) != 0 || g->pad_t > 2 * g->dil) return false;
    if (g->OH != g->H + 2 * g->pad_t - 2 * g->dil || g->OW != g->W + 2 * g->pad_l - 2 * g->dil) return false;
    if ((g->H % g->dil) != 0 || (g->W % g->dil) != 0 || (g->OH % g->dil) != 0 || (g->OW % g->dil) != 0) return false;
    if ((g->C % 32) != 0 || (g->K % 4) != 0 || g->K < 32) return false;       // (K <= 16: vector-ALU kernels)
    const WinoGeom w = make_wgeom(g);
    const long long lim = 1ll << 29;        // 2 GiB per transform-point plane: 32-bit buffer offsets with the OOB2 sentinel
    return (long long)w.T * g->C < lim && (long long)w.T * g->K < lim && (long long)g->C * g->K < lim;
}

bool wino_chosen(const pnp_conv_geom* g) {
    const int mode = wino_mode();
    if (mode <= 0 || !wino_eligible(g)) return false;
    if (mode >= 2) return true;
    // the transforms move 20 M (C + K) bytes through HBM that the direct kernel does not; the contraction saves 10 M C K flops.  Measured
    // at B = 16 (tools/bench_conv.py WINO=2 against 0, profiles/r04_conv_layers_wino_B16.txt): 512->512 0.546 -> 0.336 ms, 256->256
    // 0.156 -> 0.121, 128->256 0.083 -> 0.078 (C K / (C + K) = 85: break-even), 128->128 0.047 -> 0.050, 64->128@128^2 0.287 -> 0.429
    static const double thr = getenv("PNP_WINOGRAD_MIN") ? atof(getenv("PNP_WINOGRAD_MIN")) : 85.0;
    const WinoGeom w = make_wgeom(g);
    return (double)g->C * g->K / ((double)g->C + g->K) >= thr && w.T >= 512;
}

size_t wino_workspace_bytes(const pnp_conv_geom* g) {
    const WinoGeom w = make_wgeom(g);
    return al256((size_t)16 * g->C * g->K * 4) + al256((size_t)16 * w.T * g->C * 4) + al256((size_t)16 * w.T * g->K * 4);
}

int wino_stats_parts(const pnp_conv_geom* g) {
    const WinoGeom w = make_wgeom(g);
    int tpb, nblk;
    out_plan(w.T, g->K, &tpb, &nblk);
    return nblk;
}

// a: the convolution's arguments as make_args built them (kind 1: of the data gradient AS a convolution of dy: a.C = the forward's K,
// a.K = its C) with every epilogue field honoured; flip_transpose: a.w is the FORWARD filter [3][3][a.K][a.C]
int launch_wino(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    const WinoGeom w = make_wgeom(a.N, a.H, a.W, a.OH, a.OW, a.dil, a.pad_t);
    const size_t ub = al256((size_t)16 * a.C * a.K * 4), vb = al256((size_t)16 * w.T * a.C * 4), mb = al256((size_t)16 * w.T * a.K * 4);
    if (!ws || ws_bytes < ub + vb + mb) {
        pnp_set_error("launch_wino: workspace too small (%zu < %zu)", ws_bytes, ub + vb + mb);
        return PNP_EWORKSPACE;
    }
    PNP_REQUIRE(a.y_h == nullptr && a.o_s == 0 && a.ups == 1, "launch_wino: unsupported epilogue");
    float* U = (float*)ws;
    float* V = (float*)((char*)ws + ub);
    float* Mm = (float*)((char*)ws + ub + vb);
    const int cls = prof_class(kind);
    {
        dim3 grid((unsigned)pnp_cdiv(a.K, 32), (unsigned)pnp_cdiv(a.C, 32));
        if (flip_transpose) hipLaunchKernelGGL(wino_filter_kernel<true>, grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
        else hipLaunchKernelGGL(wino_filter_kernel<false>, grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
        PNP_CHECK_LAUNCH("wino_filter_kernel");
    }
    {
        WinoInArgs ia{};
        ia.x = a.x; ia.V = V; ia.g = w; ia.C = a.C; ia.x_bytes = a.x_bytes;
        const size_t nvec = (size_t)w.T * (a.C / 4);
        long long nb = (long long)((nvec + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(cls, st, 0.0, 4.0 * ((double)a.N * a.H * a.W * a.C + 16.0 * w.T * a.C), "wino_in_kernel");
        hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)nb), dim3(NT), 0, st, ia);
        PNP_CHECK_LAUNCH("wino_in_kernel");
    }
    {
        WinoGemmArgs ga{};
        ga.V = V; ga.U = U; ga.Mm = Mm; ga.T = w.T; ga.C = a.C; ga.K = a.K;
        ga.nblk_m = pnp_cdiv(w.T, 128); ga.nblk_n = pnp_cdiv(a.K, 128);
        ga.gn = a.gn; ga.xcd_swizzle = a.xcd_swizzle;
        dim3 grid((unsigned)(ga.nblk_m * ga.nblk_n * 16));
        const double fl = 2.0 * 16.0 * (double)w.T * a.C * a.K;
        const double by = 4.0 * 16.0 * ((double)w.T * a.C + (double)a.C * a.K + (double)w.T * a.K);
        PnpProfScope ps(cls, st, fl, by, "wino_gemm_kernel<128, 128, 2, 2, %d>", kind);
        if (kind == 0) hipLaunchKernelGGL((wino_gemm_kernel<128, 128, 2, 2, 0>), grid, dim3(NTHREADS), 0, st, ga);
        else hipLaunchKernelGGL((wino_gemm_kernel<128, 128, 2, 2, 1>), grid, dim3(NTHREADS), 0, st, ga);
        PNP_CHECK_LAUNCH("wino_gemm_kernel");
    }
    {
        WinoOutArgs oa{};
        oa.Mm = Mm; oa.y = a.y; oa.g = w; oa.K = a.K;
        int nblk;
        out_plan(w.T, a.K, &oa.tiles_per_block, &nblk);
        oa.do_drop = a.do_drop; oa.drop_thresh = a.drop_thresh; oa.drop_key = a.drop_key; oa.drop_scale = a.drop_scale;
        oa.sp = a.sp; oa.drop_sid = a.drop_sid;
        oa.res_add = a.res_add; oa.stat_ws = a.stat_ws; oa.stat_shift = a.stat_shift;
        oa.ep_scale = a.ep_scale; oa.ep_shift = a.ep_shift; oa.ep_res = a.ep_res; oa.ep_cs = a.ep_cs; oa.ep_alpha = a.ep_alpha;
        dim3 grid((unsigned)nblk, (unsigned)pnp_cdiv(a.K / 4, NT));
        PnpProfScope ps(cls, st, 0.0, 4.0 * (16.0 * w.T * a.K + (double)a.M * a.K), "wino_out_kernel");
        hipLaunchKernelGGL(wino_out_kernel, grid, dim3(NT), 0, st, oa);
        PNP_CHECK_LAUNCH("wino_out_kernel");
    }
    return PNP_OK;
}

bool wino_wgrad_chosen(const pnp_conv_geom* g) {
    const int mode = wino_wgrad_mode();
    if (mode <= 0 || wino_mode() <= 0 || !wino_eligible(g)) return false;        // PNP_WINOGRAD=0 switches the whole route off
    if (mode >= 2) return true;
    // measured at B = 16 (profiles/r04_conv_layers_wino_wgrad_B16.txt, direct ring kernel -> route): 512->512 0.610 -> 0.350 ms, g10 3.022 ->
    // 1.516, 256->512 0.331 -> 0.217, 256->256 0.193 -> 0.142 (@64^2: 0.615 -> 0.425), 128->256@32^2 0.120 -> 0.131: two transforms in front of
    // the contraction instead of one, so the break-even sits higher than the forward's
    static const double thr = getenv("PNP_WINOGRAD_WGRAD_MIN") ? atof(getenv("PNP_WINOGRAD_WGRAD_MIN")) : 120.0;
    const WinoGeom w = make_wgeom(g);
    return (double)g->C * g->K / ((double)g->C + g->K) >= thr && w.T >= 512;
}

size_t wino_wgrad_workspace_bytes(const pnp_conv_geom* g) {
    const WinoGeom w = make_wgeom(g);
    int cps;
    const int ns = wgrad_split(w.T, g->C, g->K, &cps);
    return al256((size_t)16 * w.T * g->C * 4) + al256((size_t)16 * w.T * g->K * 4) + al256((size_t)ns * 16 * g->C * g->K * 4);
}

// a: make_args(x, dy, dw, g) of the FORWARD geometry (a.x = x, a.w = dy, a.y unused); dw [3][3][C][K]
int launch_wino_wgrad(const ConvArgs& a, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const WinoGeom w = make_wgeom(a.N, a.H, a.W, a.OH, a.OW, a.dil, a.pad_t);
    int cps;
    const int ns = wgrad_split(w.T, a.C, a.K, &cps);
    const size_t vb = al256((size_t)16 * w.T * a.C * 4), yb = al256((size_t)16 * w.T * a.K * 4), sb = al256((size_t)ns * 16 * a.C * a.K * 4);
    if (!ws || ws_bytes < vb + yb + sb) {
        pnp_set_error("launch_wino_wgrad: workspace too small (%zu < %zu)", ws_bytes, vb + yb + sb);
        return PNP_EWORKSPACE;
    }
    float* V = (float*)ws;
    float* Y = (float*)((char*)ws + vb);
    float* S = (float*)((char*)ws + vb + yb);
    {
        WinoInArgs ia{};
        ia.x = a.x; ia.V = V; ia.g = w; ia.C = a.C; ia.x_bytes = a.x_bytes;
        long long nb = (long long)(((size_t)w.T * (a.C / 4) + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)a.N * a.H * a.W * a.C + 16.0 * w.T * a.C), "wino_in_kernel");
        hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)nb), dim3(NT), 0, st, ia);
        PNP_CHECK_LAUNCH("wino_in_kernel");
    }
    {
        WinoDyArgs da{};
        da.dy = a.w; da.Y = Y; da.g = w; da.K = a.K;
        long long nb = (long long)(((size_t)w.T * (a.K / 4) + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)a.M * a.K + 16.0 * w.T * a.K), "wino_dy_kernel");
        hipLaunchKernelGGL(wino_dy_kernel, dim3((unsigned)nb), dim3(NT), 0, st, da);
        PNP_CHECK_LAUNCH("wino_dy_kernel");
    }
    {
        WinoWgradGemmArgs ga{};
        ga.V = V; ga.Y = Y; ga.S = S; ga.T = w.T; ga.C = a.C; ga.K = a.K;
        ga.nblk_m = pnp_cdiv(a.C, 128); ga.nblk_n = pnp_cdiv(a.K, 128);
        ga.nsplit = ns; ga.chunks_per_split = cps; ga.xcd_swizzle = a.xcd_swizzle;
        dim3 grid((unsigned)(ga.nblk_m * ga.nblk_n * 16 * ns));
        const double fl = 2.0 * 16.0 * (double)w.T * a.C * a.K;
        const double by = 4.0 * 16.0 * ((double)w.T * a.C + (double)w.T * a.K + (double)ns * a.C * a.K);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, fl, by, "wino_wgrad_gemm_kernel<128, 128, 2, 2>");
        hipLaunchKernelGGL((wino_wgrad_gemm_kernel<128, 128, 2, 2>), grid, dim3(NTHREADS), 0, st, ga);
        PNP_CHECK_LAUNCH("wino_wgrad_gemm_kernel");
    }
    {
        const size_t nvec = (size_t)a.C * (a.K / 4);
        hipLaunchKernelGGL(wino_wgrad_out_kernel, dim3((unsigned)((nvec + NT - 1) / NT)), dim3(NT), 0, st, (const float*)S, dw, a.C, a.K, ns, accumulate);
        PNP_CHECK_LAUNCH("wino_wgrad_out_kernel");
    }
    return PNP_OK;
}

}  // namespace pnpconv

// route policy at run time (tests, A/B measurements): mode 0 never / 1 where the cost model says it pays / 2 wherever the geometry
// allows; mode < 0 only reads.  Returns the previous mode.  The workspace / parts queries follow the mode in force when they are called.
extern "C" int32_t pnp_conv2d_wino_wgrad_mode(int32_t mode) {
    const int prev = wino_wgrad_mode();
    if (mode >= 0) g_wino_wgrad_mode.store(mode > 2 ? 2 : mode, std::memory_order_relaxed);
    return prev;
}
extern "C" int32_t pnp_conv2d_wino_mode(int32_t mode) {
    const int prev = wino_mode();
    if (mode >= 0) g_wino_mode.store(mode > 2 ? 2 : mode, std::memory_order_relaxed);
    return prev;
}
