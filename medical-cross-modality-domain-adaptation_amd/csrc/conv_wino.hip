// conv_wino.hip — Winograd F(2x2, 3x3) and F(4x4, 3x3) routes for the wide stride-1 3x3 convolutions (forward, data gradient, filter
// gradient) on fp32 matrix cores.
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
//
// F(4x4, 3x3) (round 5; every transform kernel is a template on the output tile edge M = 2 / 4): a 4x4 output tile from a 6x6 patch with 36
// multiplications instead of 144 — 4x fewer MFMA flops than the direct sum (F(2x2): 2.25x), V and M are 2.25x the input / output (F(2x2):
// 4x), the GEMM kernels are the same with 36 transform points.  The price is rounding: the transforms are no longer pure additions.  On the
// interpolation points (0, 1, -1, 1/2, -2, inf) the float32 error against the float64 convolution is 3e-6..5e-6 of max|y| on the 256- /
// 512-channel layers — the direct fp32 kernel's level (2e-6..3e-6), Lavin & Gray's classic (0, +-1, +-2) lands at 1e-5 — forward, data and
// filter gradient alike (tools/wino_f43_study.py -> profiles/r05_wino_f43_tolerance.txt; oracle/tf_ops.py WINO4_* are the matrices).
// B^T and A^T are exact in binary on these points; G has thirds and fifteenths (rounded once, as float32 constants).
#include <atomic>
#include <map>
#include <mutex>
#include "conv_common.h"
#include "conv_mma.h"

using namespace pnpconv;

namespace {

constexpr int NT = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }

// ---- split-bf16 ("x3") operands of the GEMMs (conv_wino_x3.hip): v = hi + mid + lo EXACTLY, hi = bf16(v), mid = bf16(v - hi),
// lo = bf16(v - hi - mid) (round-to-nearest-even: v_cvt_pk_bf16_f32; both differences are exact in fp32).  Operand layout ("stage-major":
// what ONE stage of the GEMM reads of one plane is contiguous): [pos][C / X3_SW][plane][rows][X3_SW]; pstride = rows * X3_SW elements between planes.
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
    hi = (__bf16)v;
    const float r = v - (float)hi;
    mid = (__bf16)r;
    lo = (__bf16)(r - (float)mid);
}
__device__ __forceinline__ void st4x3(__bf16* p, size_t pstride, f32x4 v) {
    bf16x4 h, m, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        __bf16 a, b, c;
        split3(v[e], a, b, c);
        h[e] = a; m[e] = b; l[e] = c;
    }
    *reinterpret_cast<bf16x4*>(p) = h;
    *reinterpret_cast<bf16x4*>(p + pstride) = m;
    *reinterpret_cast<bf16x4*>(p + 2 * pstride) = l;
}

// Hi x Wi: the input image, Ho x Wo = Hi + 2 pad - 2 dil: the output; ps = pad / dil in {0, 1, 2} (VALID on a pre-padded image, SAME, the
// data gradient of a VALID convolution); Hsi / Hso ...: extents of one dilation phase's sub-image; tiles enumerate the OUTPUT
struct WinoGeom {
    int N, Hi, Wi, Ho, Wo, dil, ps, Hsi, Wsi, Hso, Wso, th, tw, T;
    int m;        // output tile edge: 2 (F(2x2, 3x3)) or 4 (F(4x4, 3x3))
};

// ---- the 1-D transforms of F(4, 3) on the points (0, 1, -1, 1/2, -2, inf) (oracle/tf_ops.py: WINO4_BT / WINO4_G / WINO4_AT) --------------
// B^T d (6 -> 6): every coefficient exact in binary
template <typename V>
__device__ __forceinline__ void w4_bt(const V& d0, const V& d1, const V& d2, const V& d3, const V& d4, const V& d5, V* o) {
    o[0] = (d0 + d4) + (1.5f * (d3 - d1) - 2.f * d2);
    o[1] = (d4 - d1) + (0.5f * d2 + 2.5f * d3);
    o[2] = (d4 + d1) + (0.5f * d3 - 2.5f * d2);
    o[3] = (d4 - d2) + 2.f * (d3 - d1);
    o[4] = (d4 - d2) + 0.5f * (d1 - d3);
    o[5] = (d1 + d5) + (1.5f * (d4 - d2) - 2.f * d3);
}
// A^T m (6 -> 4)
template <typename V>
__device__ __forceinline__ void w4_at(const V& m0, const V& m1, const V& m2, const V& m3, const V& m4, const V& m5, V* o) {
    const V s = m1 + m2, d = m1 - m2;
    o[0] = (m0 + s) + (m3 + m4);
    o[1] = d + (0.5f * m3 - 2.f * m4);
    o[2] = s + (0.25f * m3 + 4.f * m4);
    o[3] = (d + m5) + (0.125f * m3 - 8.f * m4);
}
// A y (4 -> 6): the transposition of A^T (filter gradient: the dy tile spread to the transform points)
template <typename V>
__device__ __forceinline__ void w4_a(const V& y0, const V& y1, const V& y2, const V& y3, V* z) {
    const V e = y0 + y2, o = y1 + y3;
    z[0] = y0;
    z[1] = e + o;
    z[2] = e - o;
    z[3] = (y0 + 0.5f * y1) + (0.25f * y2 + 0.125f * y3);
    z[4] = (y0 - 2.f * y1) + (4.f * y2 - 8.f * y3);
    z[5] = y3;
}
constexpr float W4_3 = 1.f / 3.f, W4_15 = 1.f / 15.f;
// G g (3 -> 6)
template <typename V>
__device__ __forceinline__ void w4_g(const V& g0, const V& g1, const V& g2, V* t) {
    t[0] = g0;
    t[1] = W4_3 * ((g0 + g2) + g1);
    t[2] = W4_3 * (g1 - (g0 + g2));
    t[3] = -W4_15 * ((16.f * g0 + 4.f * g2) + 8.f * g1);
    t[4] = W4_15 * ((g0 + 4.f * g2) - 2.f * g1);
    t[5] = g2;
}
// G^T s (6 -> 3): the transposition of G (filter gradient: transform points -> taps)
template <typename V>
__device__ __forceinline__ void w4_gt(const V& s0, const V& s1, const V& s2, const V& s3, const V& s4, const V& s5, V* o) {
    o[0] = s0 + (W4_3 * (s1 - s2) + W4_15 * (s4 - 16.f * s3));
    o[1] = W4_3 * (s1 + s2) - W4_15 * (8.f * s3 + 2.f * s4);
    o[2] = s5 + (W4_3 * (s1 - s2) + (4.f * W4_15) * (s4 - s3));
}

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
    int xcd;
};

template <int M, bool X3>
__global__ void __launch_bounds__(NT) wino_in_kernel(WinoInArgs a) {
    constexpr int P = M + 2;                 // patch edge
    const int C4 = a.C >> 2;
    const size_t nvec = (size_t)a.g.T * C4;
    const size_t plane = (size_t)a.g.T * a.C;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const size_t gs = (size_t)gridDim.x * NT;
    // neighbouring tiles share 2 of their P patch rows / columns: consecutive logical blocks on ONE XCD, so that the overlap is an L2 hit
    // (hardware order = block id modulo 8 XCDs: the 8 L2s each fetched the halo — 60 MB read for a 33.5 MB input, r05 counters)
    const int lb = a.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    for (size_t i = (size_t)lb * NT + threadIdx.x; i < nvec; i += gs) {
        const int t = (int)(i / C4);
        const int c = (int)(i - (size_t)t * C4) * 4;
        int n, pa, pb, ti, tj;
        tile_of(a.g, t, n, pa, pb, ti, tj);
        f32x4 d[P][P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int u = M * ti - a.g.ps + p;
            const bool uok = (unsigned)u < (unsigned)a.g.Hsi;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                const int v = M * tj - a.g.ps + q;
                const bool ok = uok & ((unsigned)v < (unsigned)a.g.Wsi);
                const unsigned off = (unsigned)(((n * a.g.Hi + pa + a.g.dil * u) * a.g.Wi + pb + a.g.dil * v) * a.C + c) * 4u;    // (host: < 2^30 elements)
                d[p][q] = bload4(rx, ok ? off : OOB);
            }
        }
        // rows (B^T d), then columns ((B^T d) B)
        f32x4 r[P][P];
        float* out = a.V + (size_t)t * a.C + c;
        // X3: V3 [pos][C / 32][plane][T][32] bf16 — transform point `pos` starts 3 * plane elements behind the previous one
        __bf16* out3 = reinterpret_cast<__bf16*>(a.V) + ((size_t)(c / X3_SW) * 3 * a.g.T + t) * X3_SW + (c % X3_SW);
        auto put = [&](int pos, f32x4 v) {
            if constexpr (X3) st4x3(out3 + (size_t)pos * 3 * plane, (size_t)a.g.T * X3_SW, v);
            else st4(out + (size_t)pos * plane, v);
        };
        if constexpr (M == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                r[0][q] = d[0][q] - d[2][q];
                r[1][q] = d[1][q] + d[2][q];
                r[2][q] = d[2][q] - d[1][q];
                r[3][q] = d[1][q] - d[3][q];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                put(4 * p + 0, r[p][0] - r[p][2]);
                put(4 * p + 1, r[p][1] + r[p][2]);
                put(4 * p + 2, r[p][2] - r[p][1]);
                put(4 * p + 3, r[p][1] - r[p][3]);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f32x4 o[6];
                w4_bt(d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q], o);
#pragma unroll
                for (int p = 0; p < 6; ++p) r[p][q] = o[p];
            }
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                f32x4 o[6];
                w4_bt(r[p][0], r[p][1], r[p][2], r[p][3], r[p][4], r[p][5], o);
#pragma unroll
                for (int q = 0; q < 6; ++q) put(6 * p + q, o[q]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// filter transform: U[pos][row][col] = (G g G^T)[pos], rows = reduction channels, cols = output channels of the GEMM.
//   TRANS false (forward):        g[r][s] = w[r][s][row][col]                      (w's fast axis = col)
//   TRANS true  (data gradient):  g[r][s] = w[2-r][2-s][col][row]   (the flipped, transposed filter: w's fast axis = row)
//   X3 false: U  [pos][row][col] fp32 (fast axis = col);  X3 true: U3 [pos][row / 32][plane][col][row % 32] bf16 split operands (fast
//   axis = row: the B operand of conv_wino_x3.hip has the reduction index contiguous).
// Where the source's and the destination's fast axes differ (TRANS != X3) the nine taps of a 32 x 32 tile go through LDS so that both the
// read and the write are coalesced; no separate flip / transpose launch in any of the four cases.
template <bool TRANS, int M, bool X3>
__global__ void __launch_bounds__(NT) wino_filter_kernel(const float* __restrict__ w, float* __restrict__ U, int rows, int cols) {
    constexpr int P = M + 2;
    constexpr bool VIA_LDS = TRANS != X3;
    __shared__ float tile[VIA_LDS ? 9 : 1][32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const size_t plane = (size_t)rows * cols;
    // element (row, col) of memory tap `tap`
    auto src = [&](int tap, int row, int col) {
        return TRANS ? w[((size_t)tap * cols + col) * rows + row] : w[((size_t)tap * rows + row) * cols + col];
    };
    if constexpr (VIA_LDS) {
        // tx walks the SOURCE's fast axis; tile[tap][j][tx]
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
            for (int j = ty; j < 32; j += 8) {
                const int sr = TRANS ? r0 + tx : r0 + j, sc = TRANS ? c0 + j : c0 + tx;
                tile[tap][j][tx] = (sc < cols && sr < rows) ? src(tap, sr, sc) : 0.f;
            }
        __syncthreads();
    }
    for (int j = ty; j < 32; j += 8) {
        // tx walks the DESTINATION's fast axis
        const int row = X3 ? r0 + tx : r0 + j, col = X3 ? c0 + j : c0 + tx;
        const bool ok = row < rows && col < cols;
        float g[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int s_ = 0; s_ < 3; ++s_) {
                const int tap = TRANS ? 8 - (r * 3 + s_) : r * 3 + s_;
                if constexpr (VIA_LDS) g[r][s_] = tile[tap][tx][j];
                else g[r][s_] = ok ? src(tap, row, col) : 0.f;
            }
        float t[P][3];
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) {
            if constexpr (M == 2) {
                t[0][s_] = g[0][s_];
                t[1][s_] = 0.5f * (g[0][s_] + g[1][s_] + g[2][s_]);
                t[2][s_] = 0.5f * (g[0][s_] - g[1][s_] + g[2][s_]);
                t[3][s_] = g[2][s_];
            } else {
                float o[6];
                w4_g(g[0][s_], g[1][s_], g[2][s_], o);
#pragma unroll
                for (int i = 0; i < 6; ++i) t[i][s_] = o[i];
            }
        }
        if (ok) {
            float* out = U + (size_t)row * cols + col;
            __bf16* out3 = reinterpret_cast<__bf16*>(U) + ((size_t)(row / X3_SW) * 3 * cols + col) * X3_SW + (row % X3_SW);
            auto put = [&](int pos, float v) {
                if constexpr (X3) {
                    __bf16 hi, mid, lo;
                    split3(v, hi, mid, lo);
                    __bf16* q = out3 + (size_t)pos * 3 * plane;
                    q[0] = hi;
                    q[(size_t)cols * X3_SW] = mid;
                    q[(size_t)cols * 2 * X3_SW] = lo;
                } else {
                    out[(size_t)pos * plane] = v;
                }
            };
#pragma unroll
            for (int i = 0; i < P; ++i) {
                if constexpr (M == 2) {
                    put(4 * i + 0, t[i][0]);
                    put(4 * i + 1, 0.5f * (t[i][0] + t[i][1] + t[i][2]));
                    put(4 * i + 2, 0.5f * (t[i][0] - t[i][1] + t[i][2]));
                    put(4 * i + 3, t[i][2]);
                } else {
                    float o[6];
                    w4_g(t[i][0], t[i][1], t[i][2], o);
#pragma unroll
                    for (int q = 0; q < 6; ++q) put(6 * i + q, o[q]);
                }
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
    int npos;        // transform points (16 / 36)
    int whole;       // V and U of ALL points fit 31-bit byte offsets: one buffer descriptor per operand for the launch (needed for > 1 tile per workgroup)
    // tail split of a persistent launch (GemmPlan): every XCD owns P tiles; its 64 workgroups take the first F = 64 * floor(P / 64) as whole
    // tiles, the remaining R = P - F tiles are cut into s pieces of the reduction each (R s <= 64: one piece per workgroup), piece 0 goes to
    // M like a whole tile, pieces 1 .. s-1 to Px[((xcd R + tail tile) (s - 1) + piece - 1)][BM][BN]; the output transform sums them
    int tail_F, tail_R, tail_s;
    float* Px;
};

// The plan of one GEMM launch of the route, shared by the workspace query, the launch and the output transform.
struct GemmPlan {
    int grid;                 // workgroups
    int P, F, R, s;           // per XCD: tiles, whole-round tiles, tail tiles, pieces per tail tile (s == 1: no split)
    size_t px_bytes;          // partial products of the pieces 1 .. s-1
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
    const int ntiles = nblk * g.npos;
    // Which tiles this workgroup works through.  Hardware places workgroup b on XCD b % 8; with xcd_swizzle every XCD owns a CONTIGUOUS
    // range of the logical tile ids (xcd_remap's partition), i.e. whole transform points: U[pos] and V[pos] come into ONE L2 (round 4's
    // mapping gave every XCD one pixel tile of every point and each of the 8 L2s read all of U — counter figure 380 MB per 512->512 launch
    // against 113 MB of operands, profiles/r05_pmc_counters_layer512_before_xcd.json).  Round 5, second half: the launch is PERSISTENT when
    // there are more tiles than workgroup slots (grid = 512 = 2 per CU): workgroup j of an XCD takes tiles j, j + 64, j + 128 ... of its
    // XCD's range and carries the software pipeline ACROSS tiles — the last stage of a tile fetches the first stage of the next one, so a
    // workgroup leaves its main loop only for the stores of a finished tile (a 16-stage tile used to pay ~2 stages of launch + first-load
    // latency + epilogue, during which the co-resident workgroup alone cannot keep the matrix pipe busy).
    int base = 0, cnt = ntiles, first = (int)blockIdx.x, step = (int)gridDim.x;
    if (g.xcd_swizzle) {
        const int NX = 8, x = (int)blockIdx.x % NX, q = ntiles / NX, r = ntiles % NX;
        base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
        cnt = q + (x < r ? 1 : 0);
        first = (int)blockIdx.x / NX;
        step = ((int)gridDim.x - x + NX - 1) / NX;
    }
    const int wm0 = (wave / WN) * (BM / WM), wn0 = (wave % WN) * (BN / WN);
    const int kg = t & 7, mrow = t >> 3;
    const int bcol = t % C4, brow = t / C4;
    // ONE descriptor per operand for the whole launch when all transform points fit 31-bit byte offsets (g.whole: the host checks), the
    // point's plane offset inside the per-row offsets; else (one tile per workgroup only) a descriptor per point
    const size_t planeV = (size_t)g.T * g.C, planeU = (size_t)g.C * g.K;

    // row / column byte offsets of tile `gid` (out of range: OOB2 — reads return 0)
    auto setup = [&](int gid, unsigned* ab, unsigned* bo, int& pos, int& m0, int& n0) {
        pos = gid / nblk;
        int mt, nt;
        tile_coords(gid - pos * nblk, g.nblk_m, g.nblk_n, g.gn, mt, nt);
        m0 = mt * BM;
        n0 = nt * BN;
        const unsigned pv = g.whole ? (unsigned)(pos * planeV) : 0u, pu = g.whole ? (unsigned)(pos * planeU) : 0u;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int m = m0 + mrow + 32 * i;
            ab[i] = (m < g.T) ? (pv + (unsigned)(m * g.C + 4 * kg)) * 4u : OOB2;
        }
#pragma unroll
        for (int i = 0; i < NPB; ++i)
            bo[i] = (n0 + 4 * bcol < g.K) ? (pu + (unsigned)((brow + RPB * i) * g.K + n0 + 4 * bcol)) * 4u : OOB2;
    };

    unsigned abase[NR], boff[NPB], abase_n[NR], boff_n[NPB];
    int pos, m0, n0, pos_n = 0, m0_n = 0, n0_n = 0;
    if (first >= cnt) return;
    const int ncc = g.C / BK;
    // the items of this workgroup: whole tiles first, local, local + step, ... (below `whole_end`), then at most one item of the tail
    // (tail_s > 1: piece `first / tail_R` of tail tile `first % tail_R`, ncc / tail_s stages; else the whole tail tile `first`)
    const int split = (g.tail_s > 1) ? g.tail_s : 1;
    const int whole_end = (split > 1) ? g.tail_F : cnt;
    const bool has_tail = split > 1 && first < g.tail_R * split;
    const int tail_tile = has_tail ? g.tail_F + first % g.tail_R : 0, tail_piece = has_tail ? first / g.tail_R : 0;
    const int cpp = ncc / split;                       // stages of a piece
    setup(base + first, abase, boff, pos, m0, n0);
    const __amdgpu_buffer_rsrc_t rx = g.whole ? make_rsrc(g.V, (unsigned)(planeV * g.npos * 4)) : make_rsrc(g.V + (size_t)pos * planeV, (unsigned)(planeV * 4));
    const __amdgpu_buffer_rsrc_t rw = g.whole ? make_rsrc(g.U, (unsigned)(planeU * g.npos * 4)) : make_rsrc(g.U + (size_t)pos * planeU, (unsigned)(planeU * 4));

    f32x4 areg[NR], breg[NPB];
    auto lstore = [&](float* An, float* Bn) {
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4*>(An + (mrow + 32 * i) * LDA + 4 * kg) = areg[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) *reinterpret_cast<f32x4*>(Bn + (brow + RPB * i) * LDB + 4 * bcol) = breg[i];
    };

    Acc<TM, TN> acc;
    acc.zero();
    ConvArgs e{};            // plain [T][K] rows of a transform point: conv_epilogue with every feature off
    e.M = g.T;
    e.K = g.K;
    e.nsplit = 1;

#pragma unroll
    for (int i = 0; i < NR; ++i) areg[i] = bload4s(rx, abase[i], 0);
#pragma unroll
    for (int i = 0; i < NPB; ++i) breg[i] = bload4s(rw, boff[i], 0);
    lstore(lds, lds + 2 * ASZ);
    __syncthreads();
    Frag<TM, TN, true, LDA, LDB> f0, f1;
    f0.load(lds, lds + 2 * ASZ, 0, wm0, wn0, lane);
    int sidx = 0;            // stages done by this workgroup: the LDS double buffer alternates across tile boundaries too
    int local = first;
    bool in_tail = false;    // the current item is this workgroup's piece of a tail tile
    for (;;) {
        // the item after this one: the next whole tile, else the tail item, else nothing
        const int nl = local + step;
        const bool next_whole = !in_tail && nl < whole_end;
        const bool next_tail = !in_tail && !next_whole && has_tail;
        const bool more = next_whole || next_tail;
        int cc0_n = 0;
        if (more) {
            setup(base + (next_whole ? nl : tail_tile), abase_n, boff_n, pos_n, m0_n, n0_n);
            cc0_n = next_tail ? tail_piece * cpp : 0;
        } else {
#pragma unroll
            for (int i = 0; i < NR; ++i) abase_n[i] = OOB2;
#pragma unroll
            for (int i = 0; i < NPB; ++i) boff_n[i] = OOB2;
        }
        const int cc_begin = in_tail ? tail_piece * cpp : 0, cc_end = in_tail ? cc_begin + cpp : ncc;
        for (int cc = cc_begin; cc < cc_end; ++cc, ++sidx) {
            const int cur = sidx & 1;
            const float* As = lds + cur * ASZ;
            const float* Bs = lds + 2 * ASZ + cur * BSZ;
            float* An = lds + (cur ^ 1) * ASZ;
            float* Bn = lds + 2 * ASZ + (cur ^ 1) * BSZ;
            // the stage fetched under this one: the next reduction group of this item, or — under its last stage — the first group of the
            // NEXT item (no next item: out-of-range offsets, the loads return zeros into a buffer nobody reads)
            const bool lastst = cc + 1 >= cc_end;
            const int cn = lastst ? cc0_n : cc + 1;
            const int sa = cn * (BK * 4);
            const int sb = (cn * BK * g.K) * 4;
            // ---- slice 0
            f1.load(As, Bs, 1, wm0, wn0, lane);
#pragma unroll
            for (int i = 0; i < NR; ++i) areg[i] = bload4s(rx, lastst ? abase_n[i] : abase[i], sa);
#pragma unroll
            for (int i = 0; i < NPB; ++i) breg[i] = bload4s(rw, lastst ? boff_n[i] : boff[i], sb);
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
        if (in_tail && tail_piece > 0) {
            // a later piece of a split tile: its partial product as a dense [BM][BN] tile of Px (rows past T are not written, columns past K
            // are never read)
            const int x = g.xcd_swizzle ? (int)blockIdx.x % 8 : 0;
            ConvArgs ep{};
            ep.M = (g.T - m0 < BM) ? g.T - m0 : BM;
            ep.K = BN;
            ep.nsplit = 1;
            float* dst = g.Px + ((size_t)(x * g.tail_R + (tail_tile - g.tail_F)) * (split - 1) + (tail_piece - 1)) * (BM * BN);
            conv_epilogue<TM, TN>(ep, acc, dst, 0, 0, wm0, wn0, lane, 0);
        } else {
            conv_epilogue<TM, TN>(e, acc, g.Mm + (size_t)pos * g.T * g.K, m0, n0, wm0, wn0, lane, 0);
        }
        if (!more) break;
        acc.zero();
#pragma unroll
        for (int i = 0; i < NR; ++i) abase[i] = abase_n[i];
#pragma unroll
        for (int i = 0; i < NPB; ++i) boff[i] = boff_n[i];
        pos = pos_n; m0 = m0_n; n0 = n0_n;
        if (next_tail) in_tail = true;
        else local = nl;
    }
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
    float drop_keep;
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
    // tail split of the GEMM launch that produced Mm (GemmPlan; tail_s <= 1: none): tile (row block, filter block) of point pos has logical
    // id pos nblk + tile_index; ids are dealt to the 8 XCDs in contiguous ranges of tail_P; within a range the ids >= tail_F were cut into
    // tail_s pieces whose partial products 1 .. tail_s-1 sit in Px as dense [128][tail_bn] tiles
    const float* Px;
    int tail_P, tail_F, tail_R, tail_s, tail_bn, nblk_m, nblk_n, gn;
};

// TS: the GEMM launch that produced Mm used the tail split (fp32 pipe, PNP_WINO_TAILSPLIT=1: off by default) — its partial products are
// summed here; a template parameter so that the default instance does not carry that code's registers (232 VGPRs + 81 spilled SGPRs with
// it in the common path: VERDICT r5 weak #4)
template <int M, bool TS = false>
__global__ void __launch_bounds__(NT) wino_out_kernel(WinoOutArgs a) {
    constexpr int P = M + 2;
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
            f32x4 m[P][P];
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int q = 0; q < P; ++q) m[p][q] = ld4(src + (size_t)(P * p + q) * plane);
            if constexpr (TS) {
                // which of this (row block, filter block)'s P^2 tiles were split: inverse of tile_coords, then the id walks the XCD ranges
                const int mt = t >> 7, nt = k / a.tail_bn;
                int bid;
                if (a.gn <= 0 || a.gn >= a.nblk_n) bid = mt * a.nblk_n + nt;
                else {
                    const int sc = nt / a.gn, left = a.nblk_n - sc * a.gn;
                    const int width = left < a.gn ? left : a.gn;
                    bid = sc * (a.nblk_m * a.gn) + mt * width + (nt - sc * a.gn);
                }
                const int nblk = a.nblk_m * a.nblk_n;
                int x = bid / a.tail_P, local = bid - x * a.tail_P;
                const size_t tsz = (size_t)128 * a.tail_bn;
                const size_t inner = (size_t)(t & 127) * a.tail_bn + (k - nt * a.tail_bn);
#pragma unroll
                for (int p = 0; p < P; ++p)
#pragma unroll
                    for (int q = 0; q < P; ++q) {
                        if (local >= a.tail_F) {
                            const float* px = a.Px + (size_t)(x * a.tail_R + local - a.tail_F) * (a.tail_s - 1) * tsz + inner;
                            for (int pc = 0; pc < a.tail_s - 1; ++pc) m[p][q] += ld4(px + pc * tsz);      // fixed order: deterministic
                        }
                        local += nblk;
                        if (local >= a.tail_P) {
                            local -= a.tail_P;
                            ++x;
                        }
                    }
            }
            // rows (A^T m), then columns ((A^T m) A)
            f32x4 o[M][M];
            if constexpr (M == 2) {
                f32x4 r[2][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    r[0][q] = m[0][q] + m[1][q] + m[2][q];
                    r[1][q] = m[1][q] - m[2][q] - m[3][q];
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    o[p][0] = r[p][0] + r[p][1] + r[p][2];
                    o[p][1] = r[p][1] - r[p][2] - r[p][3];
                }
            } else {
                f32x4 r[4][6];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    f32x4 c4[4];
                    w4_at(m[0][q], m[1][q], m[2][q], m[3][q], m[4][q], m[5][q], c4);
#pragma unroll
                    for (int p = 0; p < 4; ++p) r[p][q] = c4[p];
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) w4_at(r[p][0], r[p][1], r[p][2], r[p][3], r[p][4], r[p][5], o[p]);
            }
#pragma unroll
            for (int p = 0; p < M; ++p)
#pragma unroll
                for (int q = 0; q < M; ++q) {
                    const int u = M * ti + p, v = M * tj + q;
                    if (u >= a.g.Hso || v >= a.g.Wso) continue;
                    const size_t row = (size_t)(n * a.g.Ho + pa + a.g.dil * u) * a.g.Wo + pb + a.g.dil * v;
                    const size_t idx = row * a.K + k;
                    f32x4 val = o[p][q];
                    if (a.do_drop) {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            val[e] = pnp_drop_keep((uint32_t)(idx + e), dkey, a.drop_thresh) ? val[e] / a.drop_keep : 0.f;
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

template <int M>
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
        f32x4 y[M][M];
#pragma unroll
        for (int p = 0; p < M; ++p)
#pragma unroll
            for (int q = 0; q < M; ++q) {
                const int u = M * ti + p, v = M * tj + q;
                const f32x4 zero = {0, 0, 0, 0};
                y[p][q] = (u < a.g.Hso && v < a.g.Wso) ? ld4(a.dy + ((size_t)(n * a.g.Ho + pa + a.g.dil * u) * a.g.Wo + pb + a.g.dil * v) * a.K + k) : zero;
            }
        float* out = a.Y + (size_t)t * a.K + k;
        if constexpr (M == 2) {
            f32x4 z[4][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                z[0][q] = y[0][q];
                z[1][q] = y[0][q] + y[1][q];
                z[2][q] = y[0][q] - y[1][q];
                z[3][q] = -y[1][q];
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                st4(out + (size_t)(4 * p + 0) * plane, z[p][0]);
                st4(out + (size_t)(4 * p + 1) * plane, z[p][0] + z[p][1]);
                st4(out + (size_t)(4 * p + 2) * plane, z[p][0] - z[p][1]);
                st4(out + (size_t)(4 * p + 3) * plane, -z[p][1]);
            }
        } else {
            f32x4 z[6][4];          // rows (A y), then columns ((A y) A^T)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 c6[6];
                w4_a(y[0][q], y[1][q], y[2][q], y[3][q], c6);
#pragma unroll
                for (int p = 0; p < 6; ++p) z[p][q] = c6[p];
            }
#pragma unroll
            for (int p = 0; p < 6; ++p) {
                f32x4 o[6];
                w4_a(z[p][0], z[p][1], z[p][2], z[p][3], o);
#pragma unroll
                for (int q = 0; q < 6; ++q) st4(out + (size_t)(6 * p + q) * plane, o[q]);
            }
        }
    }
}

// ---- the filter gradient on the split-bf16 GEMM (conv_wino_x3.hip, sym 4 / 5) ------------------------------------------------------
// S[pos] = V[pos]^T x Y[pos] reduces over the TILES, so both operands need the tile index contiguous:
//     V3t [pos][Tp / 32][plane][C][32 tiles],  Y3t [pos][Tp / 32][plane][K][32 tiles]      (Tp = T rounded up to 64, zero-filled)
// — the stage-major layout of the forward's operands with rows = channels (filters).  A workgroup transforms 32 tiles x 32 channels
// (thread = tile, channel quad: the same loads and arithmetic as wino_in_kernel / wino_dy_kernel) and transposes them through LDS, PG
// transform points per round, so that every store instruction writes whole 64-byte rows (8 lanes x 4 tiles x 2 bytes).
template <int M>
__device__ __forceinline__ void bt_2d(const f32x4 (&d)[M + 2][M + 2], f32x4 (&v)[(M + 2) * (M + 2)]) {
    constexpr int P = M + 2;
    f32x4 r[P][P];
    if constexpr (M == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            r[0][q] = d[0][q] - d[2][q];
            r[1][q] = d[1][q] + d[2][q];
            r[2][q] = d[2][q] - d[1][q];
            r[3][q] = d[1][q] - d[3][q];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[4 * p + 0] = r[p][0] - r[p][2];
            v[4 * p + 1] = r[p][1] + r[p][2];
            v[4 * p + 2] = r[p][2] - r[p][1];
            v[4 * p + 3] = r[p][1] - r[p][3];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            f32x4 o[6];
            w4_bt(d[0][q], d[1][q], d[2][q], d[3][q], d[4][q], d[5][q], o);
#pragma unroll
            for (int p = 0; p < 6; ++p) r[p][q] = o[p];
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            f32x4 o[6];
            w4_bt(r[p][0], r[p][1], r[p][2], r[p][3], r[p][4], r[p][5], o);
#pragma unroll
            for (int q = 0; q < 6; ++q) v[6 * p + q] = o[q];
        }
    }
}
template <int M>
__device__ __forceinline__ void a_2d(const f32x4 (&y)[M][M], f32x4 (&v)[(M + 2) * (M + 2)]) {
    if constexpr (M == 2) {
        f32x4 z[4][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            z[0][q] = y[0][q];
            z[1][q] = y[0][q] + y[1][q];
            z[2][q] = y[0][q] - y[1][q];
            z[3][q] = -y[1][q];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            v[4 * p + 0] = z[p][0];
            v[4 * p + 1] = z[p][0] + z[p][1];
            v[4 * p + 2] = z[p][0] - z[p][1];
            v[4 * p + 3] = -z[p][1];
        }
    } else {
        f32x4 z[6][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 c6[6];
            w4_a(y[0][q], y[1][q], y[2][q], y[3][q], c6);
#pragma unroll
            for (int p = 0; p < 6; ++p) z[p][q] = c6[p];
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            f32x4 o[6];
            w4_a(z[p][0], z[p][1], z[p][2], z[p][3], o);
#pragma unroll
            for (int q = 0; q < 6; ++q) v[6 * p + q] = o[q];
        }
    }
}
// v[pos] (4 channels ch0 + 4 c4 ..) of tile tl = tid >> 3  ->  out[pos][tblk][plane][channel][tl]
template <int NP, int PG>
__device__ __forceinline__ void emit_t(const f32x4 (&v)[NP], float (*tile)[32][34], __bf16* __restrict__ out, size_t pos_stride, int tblk, int rows, int ch0,
                                       int tid) {
    const int c4 = tid & 7, tl = tid >> 3;          // producer view: channel quad, tile
    const int cl = tid >> 3, tq = tid & 7;          // consumer view: channel, tile quad
#pragma unroll
    for (int r0 = 0; r0 < NP; r0 += PG) {
#pragma unroll
        for (int p = 0; p < PG; ++p)
#pragma unroll
            for (int e = 0; e < 4; ++e) tile[p][c4 * 4 + e][tl] = v[r0 + p][e];        // (row stride 34: the 64 lanes of a wave hit 64 banks)
        __syncthreads();
        if (ch0 + cl < rows) {
#pragma unroll
            for (int p = 0; p < PG; ++p) {
                const float* src = &tile[p][cl][4 * tq];
                const f32x4 w = {src[0], src[1], src[2], src[3]};
                const int tg = tblk * 32 + 4 * tq;          // first of this thread's four tiles: stage group tg / X3_SW, position tg % X3_SW
                st4x3(out + (size_t)(r0 + p) * pos_stride + ((size_t)(tg / X3_SW) * 3 * rows + ch0 + cl) * X3_SW + (tg % X3_SW), (size_t)rows * X3_SW, w);
            }
        }
        __syncthreads();
    }
}
struct WinoTArgs {
    const float* src;      // x (wino_in_t) or dy (wino_dy_t)
    __bf16* out;
    WinoGeom g;
    int rows;              // C or K
    unsigned src_bytes;
    int ntb;               // Tp / 32
};
template <int M>
__global__ void __launch_bounds__(NT) wino_in_t_kernel(WinoTArgs a) {
    constexpr int P = M + 2, NP = P * P, PG = M == 4 ? 12 : 8;
    __shared__ float tile[PG][32][34];
    const int tid = threadIdx.x, c4 = tid & 7, tl = tid >> 3;
    const int t = blockIdx.x * 32 + tl, c = blockIdx.y * 32 + c4 * 4;
    const bool tok = t < a.g.T;
    const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.src, a.src_bytes);
    int n, pa, pb, ti, tj;
    tile_of(a.g, tok ? t : 0, n, pa, pb, ti, tj);
    f32x4 d[P][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const int u = M * ti - a.g.ps + p;
        const bool uok = tok & ((unsigned)u < (unsigned)a.g.Hsi);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int v_ = M * tj - a.g.ps + q;
            const bool ok = uok & ((unsigned)v_ < (unsigned)a.g.Wsi);
            const unsigned off = (unsigned)(((n * a.g.Hi + pa + a.g.dil * u) * a.g.Wi + pb + a.g.dil * v_) * a.rows + c) * 4u;
            d[p][q] = bload4(rx, ok ? off : OOB);
        }
    }
    f32x4 v[NP];
    bt_2d<M>(d, v);
    emit_t<NP, PG>(v, tile, a.out, (size_t)a.ntb * 3 * a.rows * 32, (int)blockIdx.x, a.rows, (int)blockIdx.y * 32, tid);
}
template <int M>
__global__ void __launch_bounds__(NT) wino_dy_t_kernel(WinoTArgs a) {
    constexpr int P = M + 2, NP = P * P, PG = M == 4 ? 12 : 8;
    __shared__ float tile[PG][32][34];
    const int tid = threadIdx.x, c4 = tid & 7, tl = tid >> 3;
    const int t = blockIdx.x * 32 + tl, k = blockIdx.y * 32 + c4 * 4;
    const bool tok = t < a.g.T && k < a.rows;
    int n, pa, pb, ti, tj;
    tile_of(a.g, t < a.g.T ? t : 0, n, pa, pb, ti, tj);
    f32x4 y[M][M];
#pragma unroll
    for (int p = 0; p < M; ++p)
#pragma unroll
        for (int q = 0; q < M; ++q) {
            const int u = M * ti + p, v_ = M * tj + q;
            const f32x4 zero = {0, 0, 0, 0};
            y[p][q] = (tok && u < a.g.Hso && v_ < a.g.Wso) ? ld4(a.src + ((size_t)(n * a.g.Ho + pa + a.g.dil * u) * a.g.Wo + pb + a.g.dil * v_) * a.rows + k) : zero;
        }
    f32x4 v[NP];
    a_2d<M>(y, v);
    emit_t<NP, PG>(v, tile, a.out, (size_t)a.ntb * 3 * a.rows * 32, (int)blockIdx.x, a.rows, (int)blockIdx.y * 32, tid);
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
    int npos;      // transform points: 16 (F(2x2)) or 36 (F(4x4))
};

// TILE (2 / 4) only names the symbol
template <int BM, int BN, int WM, int WN, int TILE>
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
    wgrad_epilogue<TM, TN>(e, acc, g.S + ((size_t)z * g.npos + pos) * g.C * g.K, m0, n0, wm0, wn0, lane);
}

// S[0] += sum_{z >= 1} S[z], one thread per float4 of the npos x C x K products: the many-way splits of the narrow layers (128->128 at
// 128^2: 14 splits of a single 128 x 128 tile per point) summed at full width — wino_wgrad_out_kernel's own loop over the splits runs
// on C K / 4 threads only (16 workgroups for that layer: 117 us at 0.3 TB/s, profiles/r05_wino_per_kernel_layers_F4x4_B16.txt)
__global__ void __launch_bounds__(NT) wino_splitsum_kernel(float* __restrict__ S, size_t nvec, int nsplit) {
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= nvec) return;
    f32x4 v = ld4(S + i * 4);
    for (int z = 1; z < nsplit; ++z) v += ld4(S + ((size_t)z * nvec + i) * 4);       // fixed order: deterministic
    st4(S + i * 4, v);
}

// dW[r][s][c][k] (+)= sum_ij G[i][r] G[j][s] sum_z S[z][P i + j][c][k]: one thread = one channel x 4 filters
template <int M>
__global__ void __launch_bounds__(NT) wino_wgrad_out_kernel(const float* __restrict__ S, float* __restrict__ dw, int C, int K, int nsplit,
                                                            int accumulate) {
    constexpr int P = M + 2, NP = P * P;
    const int K4 = K >> 2;
    const size_t nvec = (size_t)C * K4;
    const size_t plane = (size_t)C * K;
    const size_t i = (size_t)blockIdx.x * NT + threadIdx.x;
    if (i >= nvec) return;
    const size_t off = i * 4;                 // (c, k) -> c*K + k
    f32x4 s[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        f32x4 v = ld4(S + (size_t)p * plane + off);
        for (int z = 1; z < nsplit; ++z) v += ld4(S + ((size_t)z * NP + p) * plane + off);
        s[p] = v;
    }
    f32x4 t[3][P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        if constexpr (M == 2) {
            t[0][j] = s[j] + 0.5f * (s[4 + j] + s[8 + j]);
            t[1][j] = 0.5f * (s[4 + j] - s[8 + j]);
            t[2][j] = 0.5f * (s[4 + j] + s[8 + j]) + s[12 + j];
        } else {
            f32x4 o[3];
            w4_gt(s[j], s[6 + j], s[12 + j], s[18 + j], s[24 + j], s[30 + j], o);
            t[0][j] = o[0];
            t[1][j] = o[1];
            t[2][j] = o[2];
        }
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        f32x4 o[3];
        if constexpr (M == 2) {
            o[0] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
            o[1] = 0.5f * (t[r][1] - t[r][2]);
            o[2] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
        } else {
            w4_gt(t[r][0], t[r][1], t[r][2], t[r][3], t[r][4], t[r][5], o);
        }
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
#ifndef PNP_WINOGRAD_TILE_DEFAULT
#define PNP_WINOGRAD_TILE_DEFAULT 4
#endif
int env_int(const char* name, int dflt) { return getenv(name) ? atoi(getenv(name)) : dflt; }
double env_dbl(const char* name, double dflt) { return getenv(name) ? atof(getenv(name)) : dflt; }

int env_xcd_in() {
    static const int v = env_int("PNP_WINO_XCD_IN", 1);
    return v;
}
std::atomic<int> g_wino_wgrad_mode{-1};    // the filter gradient's own switch (PNP_WINOGRAD_WGRAD): 0 never / 1 planner / 2 wherever eligible
int wino_wgrad_mode() {
    int m = g_wino_wgrad_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = env_int("PNP_WINOGRAD_WGRAD", PNP_WINOGRAD_WGRAD_DEFAULT);
        if (m < 0) m = 0;
        g_wino_wgrad_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
std::atomic<int> g_wino_mode{-1};          // -1: not read yet (environment PNP_WINOGRAD, else the compiled-in default)
int wino_mode() {
    int m = g_wino_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        m = env_int("PNP_WINOGRAD", PNP_WINOGRAD_DEFAULT);
        if (m < 0) m = 0;
        g_wino_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
std::atomic<int> g_wino_tile{-1};          // largest output tile the route may use (PNP_WINOGRAD_TILE): 2 = F(2x2, 3x3) only, 4 = F(4x4, 3x3) first
int wino_tile_max() {
    int m = g_wino_tile.load(std::memory_order_relaxed);
    if (m < 0) {
        m = env_int("PNP_WINOGRAD_TILE", PNP_WINOGRAD_TILE_DEFAULT) >= 4 ? 4 : 2;
        g_wino_tile.store(m, std::memory_order_relaxed);
    }
    return m;
}

// arithmetic of the route's forward / data-gradient GEMMs (PNP_WINOGRAD_X3 / pnp_conv2d_wino_x3): 0 = fp32 matrix pipe (wino_gemm_kernel);
// 1 = split-bf16 operands on the bf16 matrix pipe with chunked accumulation (conv_wino_x3.hip) where it pays: reductions over >= 256
// channels (PNP_WINOGRAD_X3_CMIN; measured at B = 16, profiles/r06_x3_layers_B16.txt: 512->512 163 -> 108 us, 256->256 96 -> 39, g10's data
// gradient 1 554 (F(2x2)) -> 637, cls3 256->256 @64^2 157 -> 126; 128-channel reductions 167 -> 182 and 64-channel ones 219 -> 340 lose: two
// or four 32-channel stages do not amortise the pipeline fill); 2 = wherever the shapes allow (C % 64 == 0)
#ifndef PNP_WINOGRAD_X3_DEFAULT
#define PNP_WINOGRAD_X3_DEFAULT 1
#endif
std::atomic<int> g_wino_x3{-1};
int wino_x3_mode() {
    int m = g_wino_x3.load(std::memory_order_relaxed);
    if (m < 0) {
        m = env_int("PNP_WINOGRAD_X3", PNP_WINOGRAD_X3_DEFAULT);
        m = m < 0 ? 0 : (m > 2 ? 2 : m);
        g_wino_x3.store(m, std::memory_order_relaxed);
    }
    return m;
}
inline bool use_x3(int T, int C, int K) {
    static const int cmin = env_int("PNP_WINOGRAD_X3_CMIN", 256);
    const int m = wino_x3_mode();
    return m != 0 && wino_x3_dims_ok(T, C, K) && (m >= 2 || C >= cmin);
}

// ---- transformed-filter cache (pnp_conv2d_wino_filter_bind / pnp_weights_changed) -----------------------------------------------------
// U = G g G^T only changes when the filter does: the caller lends one buffer per (filter, pass) and tells the library which weights a
// kernel or a host-side load has written; a launch whose filter has a valid entry skips the transform kernel (frozen layers — the whole
// source segmenter and the shared half in the adaptation phase, adversarial.py:839-882 — never pay it again; a trained layer pays it once
// per update instead of once per pass).  Entries are keyed by (filter address, pass); an entry is valid for the tile / shape / stream it
// was filled for.  A launch recorded into a hipGraph never reads or fills an entry (a replay could not see a later invalidation): it
// transforms into its workspace like an un-cached launch.
struct UEntry {
    float* U;
    size_t bytes;
    bool valid;
    int tile, C, K;
    hipStream_t st;
    int x3;          // what the buffer holds: fp32 U (0) or the split-bf16 planes U3 (1)
};
std::mutex g_umx;
std::map<std::pair<const void*, int>, UEntry> g_ucache;
std::atomic<long long> g_ucache_hits{0}, g_ucache_fills{0};

// the buffer to transform into / read from for this launch, and whether the transform kernel has to run
float* filter_slot(const float* w, int kind, int tile, int C, int K, int x3, size_t need, float* ws_u, hipStream_t st, bool* run) {
    *run = true;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return ws_u;
    }
    std::lock_guard<std::mutex> lk(g_umx);
    auto it = g_ucache.find({(const void*)w, kind});
    if (it == g_ucache.end() || it->second.bytes < need) return ws_u;
    UEntry& e = it->second;
    if (e.valid && e.tile == tile && e.C == C && e.K == K && e.st == st && e.x3 == x3) {
        *run = false;
        g_ucache_hits.fetch_add(1, std::memory_order_relaxed);
    } else if (e.valid && e.st != st) {
        // a valid entry filled on ANOTHER stream: that stream's GEMM may still be reading it — do not overwrite it (and do not let two
        // streams refill it alternately): this launch transforms into its own workspace
        return ws_u;
    } else {
        e.valid = true; e.tile = tile; e.C = C; e.K = K; e.st = st; e.x3 = x3;
        g_ucache_fills.fetch_add(1, std::memory_order_relaxed);
    }
    return e.U;
}

// reduction split of the filter gradient's GEMMs.  256 CUs hold two workgroups each; a CU that gets r workgroups of s stages runs them in
// pairs (2 (s + 3) per pair: ~3 stages of prologue + epilogue per workgroup) and a last, lone one at ~1.7x the speed of a paired one:
// pick the split that minimises floor(r / 2) x 2 (s + 3) + (r odd) x 1.18 (s + 3), r = ceil(workgroups / 256), with >= 8 stages per
// workgroup.  512->512 at B = 16, F(4x4): 36 x 16 = 576 workgroups of 32 stages (r = 3; r05 counters: matrix pipe busy 0.58) -> two-way
// split, r = 5; F(2x2): 256 workgroups of 128 stages -> two-way split (r = 2), as measured in round 4
int wgrad_split(int T, int C, int K, int npos, int* chunks_per_split) {
    const int nblk = pnp_cdiv(C, 128) * pnp_cdiv(K, 128) * npos;
    const int nchunks = pnp_cdiv(T, BK);
    static const int fixed = env_int("PNP_WINO_WGRAD_SPLIT", 0);
    int best = 1;
    double best_cost = 1e30;
    for (int ns = 1; ns <= 16; ++ns) {
        const int cps = pnp_cdiv(nchunks, ns);
        if (ns > 1 && cps < 8) break;
        if (pnp_cdiv(nchunks, cps) != ns) continue;             // (the same number of splits as a smaller ns)
        const int r = pnp_cdiv((long long)nblk * ns, 256);
        // + every split's partials: npos C K floats written here and read by the 6x6 -> 3x3 kernel at ~5 TB/s, in the same unit (one
        // stage of a paired workgroup's half = 1.7 us)
        const double cost = ((r / 2) * 2.0 + (r & 1) * 1.18) * (cps + 3.0) + ns * 9.4e-7 * npos * (double)C * K;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
    }
    if (fixed > 0 && pnp_cdiv(nchunks, pnp_cdiv(nchunks, fixed)) == fixed) best = fixed;
    *chunks_per_split = pnp_cdiv(nchunks, best);
    return pnp_cdiv(nchunks, *chunks_per_split);
}

WinoGeom make_wgeom(int N, int Hi, int Wi, int Ho, int Wo, int dil, int pad, int m) {
    WinoGeom g{};
    g.N = N; g.Hi = Hi; g.Wi = Wi; g.Ho = Ho; g.Wo = Wo; g.dil = dil; g.ps = pad / dil; g.m = m;
    g.Hsi = Hi / dil; g.Wsi = Wi / dil; g.Hso = Ho / dil; g.Wso = Wo / dil;
    g.th = (g.Hso + m - 1) / m; g.tw = (g.Wso + m - 1) / m;
    g.T = N * dil * dil * g.th * g.tw;
    return g;
}
WinoGeom make_wgeom(const pnp_conv_geom* g, int m) { return make_wgeom(g->N, g->H, g->W, g->OH, g->OW, g->dil, g->pad_t, m); }

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

// can F(m x m, 3x3) run this stride-1 3x3 geometry at all
bool eligible_dims(int dtype, int R, int S, int stride, int pad_mode, int dil, int pad_t, int pad_l, int H, int W, int OH, int OW, int N, int C, int K,
                   int m) {
    if (dtype != PNP_DTYPE_F32 || R != 3 || S != 3 || stride != 1 || pad_mode != PNP_PAD_ZERO) return false;
    // padding 0 (VALID on a mirror-padded image: g10), dil (SAME) or 2 dil (the data gradient of a VALID convolution), same on both axes
    if (dil < 1 || dil > 2 || pad_t != pad_l || (pad_t % dil) != 0 || pad_t > 2 * dil) return false;
    if (OH != H + 2 * pad_t - 2 * dil || OW != W + 2 * pad_l - 2 * dil) return false;
    if ((H % dil) != 0 || (W % dil) != 0 || (OH % dil) != 0 || (OW % dil) != 0) return false;
    if ((C % 32) != 0 || (K % 4) != 0 || K < 32) return false;       // (K <= 16: vector-ALU kernels)
    const WinoGeom w = make_wgeom(N, H, W, OH, OW, dil, pad_t, m);
    const long long lim = 1ll << 29;        // 2 GiB per transform-point plane: 32-bit buffer offsets with the OOB2 sentinel
    return (long long)w.T * C < lim && (long long)w.T * K < lim && (long long)C * K < lim;
}

// The planner: which output tile (0 = the direct kernels, 2, 4) a layer gets.  wgrad: the filter gradient's own thresholds.
// The transforms move ~(2 + 2 P^2/M^2) 4 (C + K) bytes per output pixel through HBM that the direct kernel does not (P = M + 2: V and M are
// P^2/M^2 times the input / output, written once and read once), the contraction saves 2 (9 - P^2/M^2) C K flops per pixel: break-even in
// C K / (C + K).  F(2x2), measured at B = 16 (profiles/r04_conv_layers_wino_B16.txt): 512->512 0.546 -> 0.336 ms, 256->256 0.156 -> 0.121,
// 128->256 0.083 -> 0.078 (C K / (C + K) = 85: break-even), 128->128 0.047 -> 0.050; filter gradient (two transforms in front of the
// contraction): break-even 120.  F(4x4) moves 0.66x the bytes and saves 1.35x the flops: break-even ~ 0.5x F(2x2)'s (profiles/r05_*).
int plan_tile(int dtype, int R, int S, int stride, int pad_mode, int dil, int pad_t, int pad_l, int H, int W, int OH, int OW, int N, int C, int K,
              bool wgrad) {
    const int mode = wgrad ? wino_wgrad_mode() : wino_mode();
    if (mode <= 0 || wino_mode() <= 0) return 0;              // PNP_WINOGRAD=0 switches the whole route off
    static const double thr2 = env_dbl("PNP_WINOGRAD_MIN", 85.0), thr2w = env_dbl("PNP_WINOGRAD_WGRAD_MIN", 120.0);
    static const double thr4 = env_dbl("PNP_WINOGRAD4_MIN", 60.0), thr4w = env_dbl("PNP_WINOGRAD4_WGRAD_MIN", 60.0);
    static const int tmin2 = env_int("PNP_WINOGRAD_TMIN", 512), tmin4 = env_int("PNP_WINOGRAD4_TMIN", 128);
    static const int wgmin = env_int("PNP_WINOGRAD_WGMIN", 256), c4max = env_int("PNP_WINOGRAD4_CMAX", 1024);
    static const double thr4lo = env_dbl("PNP_WINOGRAD4_LOW", 32.0);
    const double ck = (double)C * K / ((double)C + K);
    for (int m = wino_tile_max(); m >= 2; m -= 2) {
        if (!eligible_dims(dtype, R, S, stride, pad_mode, dil, pad_t, pad_l, H, W, OH, OW, N, C, K, m)) continue;
        // F(4x4)'s rounding error grows with the reduction length (the GEMMs accumulate in the transform domain, where the output transform's
        // 8 x 8 coefficients meet cancelling sums): 512-channel reductions 3e-6..8e-6 of max|ref|, group_10's data gradient (2 560 channels)
        // 1.7e-5 (teacher-forced, profiles/r05_pytest_gpu.log) — too close to the 2e-5 adoption bar: reductions over > 1 024 channels stay
        // on F(2x2) (1e-6), at 1.80 instead of 1.39 ms for that one launch per generator step
        // (the split-bf16 GEMM accumulates in 96-channel chunks: 1.7e-6 at 2 560 channels — no cap)
        const WinoGeom w = make_wgeom(N, H, W, OH, OW, dil, pad_t, m);
        if (m == 4 && !wgrad && C > c4max && !use_x3(w.T, C, K)) continue;
        if (mode >= 2) return m;
        if (w.T < (m == 4 ? tmin4 : tmin2)) continue;
        const long long nwg = (long long)(m + 2) * (m + 2) * pnp_cdiv(w.T, 128) * pnp_cdiv(K, K <= 64 ? 64 : 128);
        const double thr = m == 4 ? (wgrad ? thr4w : thr4) : (wgrad ? thr2w : thr2);
        if (ck < thr) {
            // F(4x4) below its break-even, measured at B = 16 with the 128 x 64 GEMM tile (profiles/r05_wino_thresholds.txt, part 3):
            //  * forward / data gradient of the 64-channel layers only on the large maps (>= 4 096 GEMM workgroups): cls1 64->64 @256^2
            //    0.679 -> 0.618 / 0.665 -> 0.625 ms, cls2 64->128 @128^2 0.298 -> 0.266 / 0.300 -> 0.254; g3 64->64 @64^2 and g4 64->128 @32^2 lose
            //  * filter gradient down to 64 -> 64 unless the maps are so large that transforming x and dy costs more than the contraction saves
            //    (T > 32 768 tiles: cls1 64->64 @256^2 0.825 -> 0.933): cls2 64->128 0.387 -> 0.291, g3 64->64 0.114 -> 0.079, g4 64->128 0.065 -> 0.055
            if (m != 4 || ck < thr4lo) continue;
            if (wgrad ? w.T > 32768 : nwg < 4096) continue;
        }
        // forward / data gradient: the GEMMs' workgroups (P^2 points x 128-tile blocks x filter blocks) must cover the chip once (256 CUs)
        // — below that the direct kernel (or F(2x2): 4x the tiles) is faster: profiles/r05_wino_thresholds.txt, B = 2 and 4 per GPU.
        // Near the break-even (C K / (C + K) < 85: 128 -> 128) two rounds: g4 128->128 @32^2 0.050 / 0.058 ms against the direct 0.048 / 0.052.
        // (filter gradient: its workgroup count does not depend on the tile count; the reduction is split to fill the chip)
        if (!wgrad && nwg < (ck < 85.0 ? 2 * wgmin : wgmin)) continue;
        return m;
    }
    return 0;
}
int plan_tile(const pnp_conv_geom* g, bool wgrad) {
    if (!g) return 0;
    return plan_tile(g->dtype, g->R, g->S, g->stride, g->pad_mode, g->dil, g->pad_t, g->pad_l, g->H, g->W, g->OH, g->OW, g->N, g->C, g->K, wgrad);
}
int plan_tile(const ConvArgs& a, bool wgrad) {
    return plan_tile(a.dtype, a.R, a.S, a.stride, a.pad_mode, a.dil, a.pad_t, a.pad_l, a.H, a.W, a.OH, a.OW, a.N, a.C, a.K, wgrad);
}

}  // namespace

namespace pnpconv {

bool wino_eligible(const pnp_conv_geom* g) {
    return g && eligible_dims(g->dtype, g->R, g->S, g->stride, g->pad_mode, g->dil, g->pad_t, g->pad_l, g->H, g->W, g->OH, g->OW, g->N, g->C, g->K, 2);
}

int wino_tile(const pnp_conv_geom* g) { return plan_tile(g, false); }
bool wino_chosen(const pnp_conv_geom* g) { return plan_tile(g, false) != 0; }

// One plan per GEMM launch of forward / data gradient, shared by the workspace query, the launch and the output transform.
// Persistent (grid = 512 = 2 workgroups per CU, 64 per XCD) when there are more tiles than slots and every transform point fits 31-bit
// offsets.  Tail split: an XCD's P tiles are F = 64 floor(P / 64) whole rounds + R tail tiles; without a split R of the 64 workgroups
// run one tile more than the rest (512->512 at B = 16: P = 144 = 2 rounds + 16, i.e. a makespan of 3 tiles for 2.25 tiles of work per
// workgroup); cutting each tail tile into s pieces of the reduction (s R <= 64, >= 2 stages per piece) hands every workgroup 1 / s of a tile
// instead.  Pieces 1 .. s-1 leave their partial products in Px, summed by the output transform in a fixed order.
static GemmPlan gemm_plan(int T, int C, int K, int NP, bool xcd) {
    // (the tail split is OFF by default: measured within one run, round 5: tools/experiments/README.md — the GEMMs gain 4-16 % (512->512 163 ->
    // 156 us, 256->256 59.7 -> 49.9, 256->512 data gradient 106.5 -> 88.4) but the output transform pays for finding and summing the pieces
    // (512->512 20.0 -> 26.0 us, g10 95 -> 113): joint step 248.6 -> 249.5 slices/s, inside the noise.  Kept, parity-tested, as the measured
    // answer to "balance the 2.25 tiles per workgroup": a lone workgroup on a CU already runs its extra tile at nearly twice the speed)
    static const int persist = env_int("PNP_WINO_PERSIST", 1), split_on = env_int("PNP_WINO_TAILSPLIT", 0);
    GemmPlan p{};
    const int bn = K <= 64 ? 64 : 128;
    const long long ntiles = (long long)NP * pnp_cdiv(T, 128) * pnp_cdiv(K, bn);
    const bool whole = (double)NP * T * C * 4.0 < 2147483648.0 && (double)NP * C * K * 4.0 < 2147483648.0;
    p.grid = (int)ntiles;
    p.P = (int)ntiles; p.F = (int)ntiles; p.R = 0; p.s = 1; p.px_bytes = 0;
    if (!(persist && whole && ntiles > 512)) return p;
    p.grid = 512;
    if (!(xcd && split_on) || (ntiles % 8) != 0) return p;
    const int P = (int)(ntiles / 8), F = (P / 64) * 64, R = P - F, ncc = C / BK;
    p.P = P; p.F = F; p.R = R;
    if (R == 0) return p;
    for (int c = 8; c >= 2; c >>= 1)
        if (c * R <= 64 && (ncc % c) == 0 && ncc / c >= 2) {
            p.s = c;
            break;
        }
    if (p.s > 1) p.px_bytes = al256((size_t)8 * R * (p.s - 1) * 128 * bn * 4);
    return p;
}

static size_t fwd_ws_bytes(const WinoGeom& w, int C, int K) {
    const size_t np = (size_t)(w.m + 2) * (w.m + 2);
    if (use_x3(w.T, C, K)) return al256(np * C * K * 6) + al256(np * w.T * C * 6) + al256(np * w.T * K * 4);
    return al256(np * C * K * 4) + al256(np * w.T * C * 4) + al256(np * w.T * K * 4) + gemm_plan(w.T, C, K, (int)np, true).px_bytes;
}

size_t wino_workspace_bytes(const pnp_conv_geom* g) {
    const int m = plan_tile(g, false);
    return fwd_ws_bytes(make_wgeom(g, m ? m : 2), g->C, g->K);
}

int wino_stats_parts(const pnp_conv_geom* g) {
    if (x3d_chosen(g)) return x3d_stats_parts(g);            // (conv_x3_direct.hip takes the layer from launch_wino)
    const int m = plan_tile(g, false);
    const WinoGeom w = make_wgeom(g, m ? m : 2);
    int tpb, nblk;
    out_plan(w.T, g->K, &tpb, &nblk);
    return nblk;
}

template <int M>
static int launch_wino_m(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int NP = (M + 2) * (M + 2);
    const WinoGeom w = make_wgeom(a.N, a.H, a.W, a.OH, a.OW, a.dil, a.pad_t, M);
    const bool x3 = use_x3(w.T, a.C, a.K);
    const size_t eb = x3 ? 6 : 4;          // bytes per transformed operand value
    const size_t ub = al256((size_t)NP * a.C * a.K * eb), vb = al256((size_t)NP * w.T * a.C * eb), mb = al256((size_t)NP * w.T * a.K * 4);
    static const int xcd_mode = env_int("PNP_WINO_XCD", 2);
    const bool xcd = a.xcd_swizzle && xcd_mode;
    GemmPlan plan = gemm_plan(w.T, a.C, a.K, NP, xcd);
    if (x3) { plan.s = 1; plan.px_bytes = 0; }
    if (!ws || ws_bytes < ub + vb + mb + plan.px_bytes) {
        pnp_set_error("launch_wino: workspace too small (%zu < %zu)", ws_bytes, ub + vb + mb + plan.px_bytes);
        return PNP_EWORKSPACE;
    }
    PNP_REQUIRE(a.y_h == nullptr && a.o_s == 0 && a.ups == 1, "launch_wino: unsupported epilogue");
    bool run_filter;
    float* U = filter_slot(a.w, flip_transpose ? 1 : 0, M, a.C, a.K, x3 ? 1 : 0, (size_t)NP * a.C * a.K * eb, (float*)ws, st, &run_filter);
    float* V = (float*)((char*)ws + ub);
    float* Mm = (float*)((char*)ws + ub + vb);
    float* Px = (float*)((char*)ws + ub + vb + mb);
    const int cls = prof_class(kind);
    if (run_filter) {
        dim3 grid((unsigned)pnp_cdiv(a.K, 32), (unsigned)pnp_cdiv(a.C, 32));
        PnpProfScope ps(cls, st, 0.0, (36.0 + eb * NP) * a.C * a.K, "wino_filter_kernel<%s, %d, %s>", flip_transpose ? "true" : "false", M, x3 ? "true" : "false");
        if (x3) {
            if (flip_transpose) hipLaunchKernelGGL((wino_filter_kernel<true, M, true>), grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
            else hipLaunchKernelGGL((wino_filter_kernel<false, M, true>), grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
        } else if (flip_transpose) hipLaunchKernelGGL((wino_filter_kernel<true, M, false>), grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
        else hipLaunchKernelGGL((wino_filter_kernel<false, M, false>), grid, dim3(NT), 0, st, a.w, U, a.C, a.K);
        PNP_CHECK_LAUNCH("wino_filter_kernel");
    }
    {
        WinoInArgs ia{};
        ia.x = a.x; ia.V = V; ia.g = w; ia.C = a.C; ia.x_bytes = a.x_bytes; ia.xcd = a.xcd_swizzle ? env_xcd_in() : 0;
        const size_t nvec = (size_t)w.T * (a.C / 4);
        long long nb = (long long)((nvec + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(cls, st, 0.0, 4.0 * (double)a.N * a.H * a.W * a.C + (double)eb * NP * w.T * a.C, "wino_in_kernel<%d, %s>", M, x3 ? "true" : "false");
        if (x3) hipLaunchKernelGGL((wino_in_kernel<M, true>), dim3((unsigned)nb), dim3(NT), 0, st, ia);
        else hipLaunchKernelGGL((wino_in_kernel<M, false>), dim3((unsigned)nb), dim3(NT), 0, st, ia);
        PNP_CHECK_LAUNCH("wino_in_kernel");
    }
    if (x3) {
        static const int wino_gn3 = env_int("PNP_WINO_GN", -1);
        const int rc = launch_wino_gemm_x3((const unsigned short*)V, (const unsigned short*)U, Mm, w.T, a.C, a.K, NP, (M == 4 ? 2 : 0) + (kind != 0), wino_gn3 >= 0 ? wino_gn3 : a.gn,
                                           xcd ? 1 : 0, 1, a.C / 32, st);
        if (rc != PNP_OK) return rc;
    } else {
        WinoGemmArgs ga{};
        ga.V = V; ga.U = U; ga.Mm = Mm; ga.T = w.T; ga.C = a.C; ga.K = a.K;
        const bool narrow = a.K <= 64;                 // 64 filters: a 128 x 64 tile (a 128-wide one would be half empty)
        ga.nblk_m = pnp_cdiv(w.T, 128); ga.nblk_n = pnp_cdiv(a.K, narrow ? 64 : 128);
        static const int wino_gn = env_int("PNP_WINO_GN", -1);          // filter-tile group width of the route's GEMM (< 0: the convolutions' PNP_CONV_GN)
        ga.gn = wino_gn >= 0 ? wino_gn : a.gn; ga.xcd_swizzle = xcd ? 2 : 0;
        ga.npos = NP;
        ga.whole = ((double)NP * w.T * a.C * 4.0 < 2147483648.0 && (double)NP * a.C * a.K * 4.0 < 2147483648.0) ? 1 : 0;
        ga.tail_F = plan.F; ga.tail_R = plan.R; ga.tail_s = plan.s; ga.Px = Px;
        dim3 grid((unsigned)plan.grid);
        const double fl = 2.0 * NP * (double)w.T * a.C * a.K;
        const double by = 4.0 * NP * ((double)w.T * a.C + (double)a.C * a.K + (double)w.T * a.K);
        constexpr int KB = M == 4 ? 2 : 0;            // symbol: <.., 0 / 1> F(2x2) forward / data gradient, <.., 2 / 3> F(4x4)
        PnpProfScope ps(cls, st, fl, by, "wino_gemm_kernel<128, %d, 2, 2, %d>", narrow ? 64 : 128, KB + (kind != 0));
        if (narrow) {
            if (kind == 0) hipLaunchKernelGGL((wino_gemm_kernel<128, 64, 2, 2, KB>), grid, dim3(NTHREADS), 0, st, ga);
            else hipLaunchKernelGGL((wino_gemm_kernel<128, 64, 2, 2, KB + 1>), grid, dim3(NTHREADS), 0, st, ga);
        } else if (kind == 0) hipLaunchKernelGGL((wino_gemm_kernel<128, 128, 2, 2, KB>), grid, dim3(NTHREADS), 0, st, ga);
        else hipLaunchKernelGGL((wino_gemm_kernel<128, 128, 2, 2, KB + 1>), grid, dim3(NTHREADS), 0, st, ga);
        PNP_CHECK_LAUNCH("wino_gemm_kernel");
    }
    {
        WinoOutArgs oa{};
        oa.Mm = Mm; oa.y = a.y; oa.g = w; oa.K = a.K;
        int nblk;
        out_plan(w.T, a.K, &oa.tiles_per_block, &nblk);
        oa.do_drop = a.do_drop; oa.drop_thresh = a.drop_thresh; oa.drop_key = a.drop_key; oa.drop_keep = a.drop_keep;
        oa.sp = a.sp; oa.drop_sid = a.drop_sid;
        oa.res_add = a.res_add; oa.stat_ws = a.stat_ws; oa.stat_shift = a.stat_shift;
        oa.ep_scale = a.ep_scale; oa.ep_shift = a.ep_shift; oa.ep_res = a.ep_res; oa.ep_cs = a.ep_cs; oa.ep_alpha = a.ep_alpha;
        oa.Px = Px; oa.tail_P = plan.P; oa.tail_F = plan.F; oa.tail_R = plan.R; oa.tail_s = plan.s; oa.tail_bn = a.K <= 64 ? 64 : 128;
        static const int wino_gn_o = env_int("PNP_WINO_GN", -1);
        oa.nblk_m = pnp_cdiv(w.T, 128); oa.nblk_n = pnp_cdiv(a.K, oa.tail_bn); oa.gn = wino_gn_o >= 0 ? wino_gn_o : a.gn;
        dim3 grid((unsigned)nblk, (unsigned)pnp_cdiv(a.K / 4, NT));
        PnpProfScope ps(cls, st, 0.0, 4.0 * ((double)NP * w.T * a.K + (double)a.M * a.K), "wino_out_kernel<%d>", M);
        if (oa.tail_s > 1) hipLaunchKernelGGL((wino_out_kernel<M, true>), grid, dim3(NT), 0, st, oa);
        else hipLaunchKernelGGL((wino_out_kernel<M, false>), grid, dim3(NT), 0, st, oa);
        PNP_CHECK_LAUNCH("wino_out_kernel");
    }
    return PNP_OK;
}

// a: the convolution's arguments as make_args built them (kind 1: of the data gradient AS a convolution of dy: a.C = the forward's K,
// a.K = its C) with every epilogue field honoured; flip_transpose: a.w is the FORWARD filter [3][3][a.K][a.C].  The tile is planned here
// from the same fields the workspace / parts queries see (one decision per call)
int launch_wino(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st) {
    // the narrow layers (32 / 64 input channels, <= 128 filters) run as DIRECT split-bf16 convolutions where that is on: no transforms,
    // no 2.25x tensors (conv_x3_direct.hip; the workspace this route asked for is more than its filter image needs)
    if (x3d_chosen(a)) return launch_x3_direct(a, kind, flip_transpose, ws, ws_bytes, st);
    const int m = plan_tile(a, false);
    PNP_REQUIRE(m == 2 || m == 4, "launch_wino: the planner does not route this layer (policy changed between the query and the launch?)");
    return m == 4 ? launch_wino_m<4>(a, kind, flip_transpose, ws, ws_bytes, st) : launch_wino_m<2>(a, kind, flip_transpose, ws, ws_bytes, st);
}

bool wino_wgrad_chosen(const pnp_conv_geom* g) { return plan_tile(g, true) != 0; }
int wino_wgrad_tile(const pnp_conv_geom* g) { return plan_tile(g, true); }

// the filter gradient's GEMMs on the split-bf16 kernel (PNP_WINOGRAD_X3 on, PNP_WINOGRAD_X3_WGRAD != 0).  Mode 1: where measured to pay at
// B = 16 (profiles/r06_x3_wgrad_layers_B16.txt) — reductions over 256 .. 2 048 tiles, i.e. the 32^2 maps: 512->512 0.239 -> 0.200 ms, g10
// 0.94 -> 0.71, 256->256 0.089 -> 0.074; on the large maps the two TRANSPOSING transforms (6 bytes per value, through LDS) cost more than
// the GEMM gains: cls3 256->256 @64^2 (4 096 tiles) 0.255 -> 0.267, cls2 128->128 @128^2 (16 384 tiles) 0.36 -> 0.48.  Mode 2: everywhere.
inline int x3_tp(int T) { return (T + 63) & ~63; }
inline bool use_x3_wgrad(int T, int C, int K) {
    static const int on = env_int("PNP_WINOGRAD_X3_WGRAD", 1), tmin = env_int("PNP_WINOGRAD_X3_WGRAD_TMIN", 256), tmax = env_int("PNP_WINOGRAD_X3_WGRAD_TMAX", 2048);
    const int m = wino_x3_mode();
    if (!on || m == 0 || !wino_x3_dims_ok(C, x3_tp(T), K) || !wino_x3_dims_ok(K, x3_tp(T), C)) return false;
    return m >= 2 || (T >= tmin && T <= tmax);
}
// reduction split of the x3 filter-gradient GEMMs: one 8-wave workgroup per CU (256 slots), a workgroup of s stages costs ~s + 3 (pipeline
// fill + epilogue); every split writes npos C K floats that the 6x6 -> 3x3 kernel reads back.  Returns the splits; *sps = stages per split (even)
int wgrad_split_x3(int T, int C, int K, int npos, int* sps) {
    const int nchunks = x3_tp(T) / 64;
    const long long tiles = (long long)pnp_cdiv(C, 128) * pnp_cdiv(K, K <= 64 ? 64 : 128) * npos;
    int best = 1;
    double best_cost = 1e30;
    for (int ns = 1; ns <= 32 && ns <= nchunks; ++ns) {
        const int cps = pnp_cdiv(nchunks, ns);
        if (pnp_cdiv(nchunks, cps) != ns) continue;
        if (ns > 1 && cps < 2) break;
        const int rounds = pnp_cdiv(tiles * ns, 256);
        const double cost = rounds * (2.0 * cps + 3.0) + ns * 1.5e-6 * npos * (double)C * K;
        if (cost < best_cost - 1e-9) { best_cost = cost; best = ns; }
    }
    const int cps = pnp_cdiv(nchunks, best);
    *sps = 2 * cps;
    return pnp_cdiv(nchunks, cps);
}

static size_t wgrad_ws_bytes(const WinoGeom& w, int C, int K) {
    const int np = (w.m + 2) * (w.m + 2);
    if (use_x3_wgrad(w.T, C, K)) {
        int sps;
        const int ns = wgrad_split_x3(w.T, C, K, np, &sps);
        const size_t tp = (size_t)x3_tp(w.T);
        return al256((size_t)np * tp * C * 6) + al256((size_t)np * tp * K * 6) + al256((size_t)ns * np * C * K * 4);
    }
    int cps;
    const int ns = wgrad_split(w.T, C, K, np, &cps);
    return al256((size_t)np * w.T * C * 4) + al256((size_t)np * w.T * K * 4) + al256((size_t)ns * np * C * K * 4);
}

size_t wino_wgrad_workspace_bytes(const pnp_conv_geom* g) {
    const int m = plan_tile(g, true);
    return wgrad_ws_bytes(make_wgeom(g, m ? m : 2), g->C, g->K);
}

template <int M>
static int launch_wino_wgrad_m(const ConvArgs& a, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    constexpr int NP = (M + 2) * (M + 2);
    const WinoGeom w = make_wgeom(a.N, a.H, a.W, a.OH, a.OW, a.dil, a.pad_t, M);
    if (use_x3_wgrad(w.T, a.C, a.K)) {
        int sps;
        const int ns = wgrad_split_x3(w.T, a.C, a.K, NP, &sps);
        const int tp = x3_tp(w.T), ntb = tp / 32;
        const size_t vb = al256((size_t)NP * tp * a.C * 6), yb = al256((size_t)NP * tp * a.K * 6), sb = al256((size_t)ns * NP * a.C * a.K * 4);
        if (!ws || ws_bytes < vb + yb + sb) {
            pnp_set_error("launch_wino_wgrad: workspace too small (%zu < %zu)", ws_bytes, vb + yb + sb);
            return PNP_EWORKSPACE;
        }
        __bf16* V3 = (__bf16*)ws;
        __bf16* Y3 = (__bf16*)((char*)ws + vb);
        float* S = (float*)((char*)ws + vb + yb);
        {
            WinoTArgs ta{};
            ta.src = a.x; ta.out = V3; ta.g = w; ta.rows = a.C; ta.src_bytes = a.x_bytes; ta.ntb = ntb;
            PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * (double)a.N * a.H * a.W * a.C + 6.0 * NP * tp * a.C, "wino_in_t_kernel<%d>", M);
            hipLaunchKernelGGL(wino_in_t_kernel<M>, dim3((unsigned)ntb, (unsigned)(a.C / 32)), dim3(NT), 0, st, ta);
            PNP_CHECK_LAUNCH("wino_in_t_kernel");
        }
        {
            WinoTArgs ta{};
            ta.src = a.w; ta.out = Y3; ta.g = w; ta.rows = a.K; ta.src_bytes = 0; ta.ntb = ntb;
            PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * (double)a.M * a.K + 6.0 * NP * tp * a.K, "wino_dy_t_kernel<%d>", M);
            hipLaunchKernelGGL(wino_dy_t_kernel<M>, dim3((unsigned)ntb, (unsigned)pnp_cdiv(a.K, 32)), dim3(NT), 0, st, ta);
            PNP_CHECK_LAUNCH("wino_dy_t_kernel");
        }
        const int rc = launch_wino_gemm_x3((const unsigned short*)V3, (const unsigned short*)Y3, S, a.C, tp, a.K, NP, M == 4 ? 5 : 4, 0, a.xcd_swizzle ? 1 : 0, ns, sps, st);
        if (rc != PNP_OK) return rc;
        int ns_out = ns;
        if (ns > 2 && (size_t)a.C * (a.K / 4) < (size_t)NT * 512) {
            const size_t nv = (size_t)NP * a.C * (a.K / 4);
            PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)ns + 1.0) * NP * a.C * a.K, "wino_splitsum_kernel");
            hipLaunchKernelGGL(wino_splitsum_kernel, dim3((unsigned)((nv + NT - 1) / NT)), dim3(NT), 0, st, S, nv, ns);
            PNP_CHECK_LAUNCH("wino_splitsum_kernel");
            ns_out = 1;
        }
        const size_t nvec = (size_t)a.C * (a.K / 4);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)ns_out * NP + 9.0 * (1 + (accumulate != 0))) * a.C * a.K, "wino_wgrad_out_kernel<%d>", M);
        hipLaunchKernelGGL(wino_wgrad_out_kernel<M>, dim3((unsigned)((nvec + NT - 1) / NT)), dim3(NT), 0, st, (const float*)S, dw, a.C, a.K, ns_out, accumulate);
        PNP_CHECK_LAUNCH("wino_wgrad_out_kernel");
        return PNP_OK;
    }
    int cps;
    const int ns = wgrad_split(w.T, a.C, a.K, NP, &cps);
    const size_t vb = al256((size_t)NP * w.T * a.C * 4), yb = al256((size_t)NP * w.T * a.K * 4), sb = al256((size_t)ns * NP * a.C * a.K * 4);
    if (!ws || ws_bytes < vb + yb + sb) {
        pnp_set_error("launch_wino_wgrad: workspace too small (%zu < %zu)", ws_bytes, vb + yb + sb);
        return PNP_EWORKSPACE;
    }
    float* V = (float*)ws;
    float* Y = (float*)((char*)ws + vb);
    float* S = (float*)((char*)ws + vb + yb);
    {
        WinoInArgs ia{};
        ia.x = a.x; ia.V = V; ia.g = w; ia.C = a.C; ia.x_bytes = a.x_bytes; ia.xcd = a.xcd_swizzle ? env_xcd_in() : 0;
        long long nb = (long long)(((size_t)w.T * (a.C / 4) + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)a.N * a.H * a.W * a.C + (double)NP * w.T * a.C), "wino_in_kernel<%d, false>", M);
        hipLaunchKernelGGL((wino_in_kernel<M, false>), dim3((unsigned)nb), dim3(NT), 0, st, ia);
        PNP_CHECK_LAUNCH("wino_in_kernel");
    }
    {
        WinoDyArgs da{};
        da.dy = a.w; da.Y = Y; da.g = w; da.K = a.K;
        long long nb = (long long)(((size_t)w.T * (a.K / 4) + NT - 1) / NT);
        if (nb > 65536) nb = 65536;
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)a.M * a.K + (double)NP * w.T * a.K), "wino_dy_kernel<%d>", M);
        hipLaunchKernelGGL(wino_dy_kernel<M>, dim3((unsigned)nb), dim3(NT), 0, st, da);
        PNP_CHECK_LAUNCH("wino_dy_kernel");
    }
    {
        WinoWgradGemmArgs ga{};
        ga.V = V; ga.Y = Y; ga.S = S; ga.T = w.T; ga.C = a.C; ga.K = a.K;
        ga.nblk_m = pnp_cdiv(a.C, 128); ga.nblk_n = pnp_cdiv(a.K, 128);
        ga.nsplit = ns; ga.chunks_per_split = cps; ga.xcd_swizzle = a.xcd_swizzle; ga.npos = NP;
        dim3 grid((unsigned)(ga.nblk_m * ga.nblk_n * NP * ns));
        const double fl = 2.0 * NP * (double)w.T * a.C * a.K;
        const double by = 4.0 * NP * ((double)w.T * a.C + (double)w.T * a.K + (double)ns * a.C * a.K);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, fl, by, "wino_wgrad_gemm_kernel<128, 128, 2, 2, %d>", M);
        hipLaunchKernelGGL((wino_wgrad_gemm_kernel<128, 128, 2, 2, M>), grid, dim3(NTHREADS), 0, st, ga);
        PNP_CHECK_LAUNCH("wino_wgrad_gemm_kernel");
    }
    int ns_out = ns;
    if (ns > 2 && (size_t)a.C * (a.K / 4) < (size_t)NT * 512) {       // few (channel, filter) vectors, many splits: sum the splits at full width first
        const size_t nv = (size_t)NP * a.C * (a.K / 4);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)ns + 1.0) * NP * a.C * a.K, "wino_splitsum_kernel");
        hipLaunchKernelGGL(wino_splitsum_kernel, dim3((unsigned)((nv + NT - 1) / NT)), dim3(NT), 0, st, S, nv, ns);
        PNP_CHECK_LAUNCH("wino_splitsum_kernel");
        ns_out = 1;
    }
    {
        const size_t nvec = (size_t)a.C * (a.K / 4);
        PnpProfScope ps(PNP_PROF_CONV_WGRAD, st, 0.0, 4.0 * ((double)ns_out * NP + 9.0 * (1 + (accumulate != 0))) * a.C * a.K, "wino_wgrad_out_kernel<%d>", M);
        hipLaunchKernelGGL(wino_wgrad_out_kernel<M>, dim3((unsigned)((nvec + NT - 1) / NT)), dim3(NT), 0, st, (const float*)S, dw, a.C, a.K, ns_out, accumulate);
        PNP_CHECK_LAUNCH("wino_wgrad_out_kernel");
    }
    return PNP_OK;
}

// a: make_args(x, dy, dw, g) of the FORWARD geometry (a.x = x, a.w = dy, a.y unused); dw [3][3][C][K]
int launch_wino_wgrad(const ConvArgs& a, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const int m = plan_tile(a, true);
    PNP_REQUIRE(m == 2 || m == 4, "launch_wino_wgrad: the planner does not route this layer (policy changed between the query and the launch?)");
    return m == 4 ? launch_wino_wgrad_m<4>(a, dw, accumulate, ws, ws_bytes, st) : launch_wino_wgrad_m<2>(a, dw, accumulate, ws, ws_bytes, st);
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
// largest output tile of the route: 2 = F(2x2, 3x3) only, 4 = F(4x4, 3x3) wherever the planner takes it (else F(2x2), else the direct
// kernels); tile < 2 only reads.  Returns the previous value.
extern "C" int32_t pnp_conv2d_wino_tile(int32_t tile) {
    const int prev = wino_tile_max();
    if (tile >= 2) g_wino_tile.store(tile >= 4 ? 4 : 2, std::memory_order_relaxed);
    return prev;
}

// arithmetic of the route's forward / data-gradient GEMMs: 0 = fp32 MFMA (wino_gemm_kernel), 1 = split-bf16 operands with chunked
// accumulation (wino_gemm_x3_kernel); mode < 0 only reads.  Returns the previous mode.  Workspace queries follow the mode in force.
extern "C" int32_t pnp_conv2d_wino_x3(int32_t mode) {
    const int prev = wino_x3_mode();
    if (mode >= 0) g_wino_x3.store(mode > 2 ? 2 : mode, std::memory_order_relaxed);
    return prev;
}

// ---- transformed-filter cache: the caller's side -----------------------------------------------------------------------------------------
// bytes of one (filter, pass) entry that serves either output tile: 36 C K floats; 0: this filter shape never takes the route
extern "C" size_t pnp_conv2d_wino_filter_bytes(int32_t C, int32_t K) {
    if (C <= 0 || K < 32 || (C % 32) != 0 || (K % 4) != 0) return 0;
    return (size_t)36 * C * K * 6;        // the larger of the two formats: fp32 U (4 bytes per value) / three bf16 planes (6)
}
// lend (U != null) or withdraw (U == null) the buffer of filter w's pass (kind 0: forward, 1: data gradient); w == null withdraws every
// entry.  The buffer must stay allocated until it is withdrawn; a new binding starts invalid.
extern "C" int pnp_conv2d_wino_filter_bind(const float* w, int32_t kind, float* U, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_umx);
    if (!w) {
        g_ucache.clear();
        return PNP_OK;
    }
    PNP_REQUIRE(kind == 0 || kind == 1, "pnp_conv2d_wino_filter_bind: kind must be 0 (forward) or 1 (data gradient)");
    if (!U) {
        g_ucache.erase({(const void*)w, (int)kind});
        return PNP_OK;
    }
    PNP_REQUIRE(bytes > 0 && ((uintptr_t)U & 15) == 0, "pnp_conv2d_wino_filter_bind: empty or misaligned buffer");
    g_ucache[{(const void*)w, (int)kind}] = UEntry{U, bytes, false, 0, 0, 0, nullptr, 0};
    return PNP_OK;
}
// the weights in [lo, hi) were written (optimiser / clip kernel queued, host-side load): their entries are stale.  lo == null: all.
extern "C" void pnp_weights_changed(const void* lo, const void* hi) {
    std::lock_guard<std::mutex> lk(g_umx);
    for (auto& kv : g_ucache)
        if (!lo || ((const char*)kv.first.first >= (const char*)lo && (const char*)kv.first.first < (const char*)hi)) kv.second.valid = false;
}
// (hits, fills) since the last call with reset != 0 — tests and the bench's launch accounting
extern "C" void pnp_conv2d_wino_filter_stats(int64_t* hits, int64_t* fills, int32_t reset) {
    if (hits) *hits = g_ucache_hits.load(std::memory_order_relaxed);
    if (fills) *fills = g_ucache_fills.load(std::memory_order_relaxed);
    if (reset) {
        g_ucache_hits.store(0, std::memory_order_relaxed);
        g_ucache_fills.store(0, std::memory_order_relaxed);
    }
}
