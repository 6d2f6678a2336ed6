// conv_common.h — what the convolution translation units share: the kernel argument block, tile / XCD mapping, accumulators and the
// buffer-descriptor loads (conv_igemm.hip: fp32 MFMA + vector-ALU kernels and ALL host planning; conv_bf16.hip: bf16-operand MFMA kernels).
#pragma once
#include "pnp_common.h"

namespace pnpconv {

constexpr int BK = 32;
constexpr int NTHREADS = 256;
#ifndef PNP_CONV_ABLATE
#define PNP_CONV_ABLATE 0
#endif
constexpr int kAblate = PNP_CONV_ABLATE;

struct ConvArgs {
    const float* x;
    const float* w;
    float* y;
    int N, H, W, C, K, R, S, OH, OW, stride, dil, pad_t, pad_l, pad_mode;
    int ups;    // zero-upsampling factor of the input (dgrad of a strided conv); 1 otherwise
    int M;      // N*OH*OW
    int Kred;   // R*S*C
    int OHW;    // OH*OW
    int nblk_m, nblk_n;
    int gn;                       // filter-tile group width of the workgroup -> tile order (0: plain row-major)
    int nsplit, chunks_per_split;   // wgrad only
    long long split_stride;          // wgrad only: elements between split partials
    uint32_t drop_thresh, drop_key;
    float drop_keep;             // keep_prob: tf.nn.dropout DIVIDES (div(x, keep_prob) * mask) — one correctly rounded fp32 division per kept value
    int do_drop;
    const pnp_step_params* sp;   // step capture: the dropout seed lives in device memory (pnp_common.h); null: drop_key as passed
    uint32_t drop_sid;           // the call site's stream id (only read with sp)
    int xcd_swizzle;
    unsigned x_bytes, w_bytes;   // sizes of the tensors behind a.x / a.w (buffer descriptors)
    int stagger;                 // s_sleep units (64 clk) by which every second dispatch wave of workgroups starts late
    // output scatter of one stride-phase of a strided data gradient (o_s = 0: plain [M][K] rows): GEMM row (n, i, j) is the
    // image pixel (n, o_h0 + i*o_s, o_w0 + j*o_s) of an o_H x o_W image
    int o_s, o_H, o_W, o_h0, o_w0;
    // fused inference-mode batch norm (+ shortcut + leaky-ReLU) behind the convolution (monitoring / frozen-BN forwards):
    //   y = act( drop(acc) * ep_scale[k] + ep_shift[k] + pad_channels(ep_res) ),  act(v) = v > 0 ? v : ep_alpha * v  (ep_alpha < 0: none)
    const float* ep_scale;
    const float* ep_shift;
    const float* ep_res;        // [M][ep_cs] or null
    int ep_cs;                   // shortcut channels, zero-padded (K - ep_cs)/2 on each side
    float ep_alpha;
    int dtype;                   // PNP_DTYPE_F32 / PNP_DTYPE_BF16: arithmetic type of the MFMA operands (tensors in HBM are fp32 either way)
    // batch-norm statistics from the epilogue (training-mode conv -> dropout -> BN): per (pixel tile, wave row) partial sums of
    // (v - stat_shift[k]) and its square over the rows the wave owns, written to stat_ws[(part*2 + q)*K + k]; null: off
    float* stat_ws;
    const float* stat_shift;
    // data gradient: dx = conv + res_add ([M][K] rows like the output; the gradient that reaches the same tensor through a residual
    // shortcut) — added by the workgroups of reduction split 0.  filter gradient: accumulate != 0 adds into dW instead of overwriting
    // (un-split launches; split launches accumulate in the reduce kernel)
    const float* res_add;
    int accumulate;
    // OW and OH*OW powers of two (every layer of the model): log2 of both, so that row -> (image, oh, ow) is two shifts and two masks
    // instead of two integer divisions (~25 VALU each; a 128x64 tile of a 64-channel layer spends ~3 % of its life on them, the
    // scattered epilogue of a stride-phase data gradient far more); -1: divide
    int ow_sh, ohw_sh;
    // bf16-resident data gradient of ONE stride phase (conv_bf16r.hip): the kernel's R x S taps are the phase's sub-filter, read out of
    // the FULL filter shadow [tap][C][K]: kernel tap (tr, ts) is full-filter tap (ph_pa + ph_st (R-1-tr), ph_pb + ph_st (S-1-ts)) of a
    // filter ph_S taps wide.  ph_st = 0: not a phase.
    int ph_st, ph_pa, ph_pb, ph_S;
    // bf16 copy of the output, [M][K] rows (conv_bf16r.hip: the operand of the NEXT convolution, written by the same epilogue); null: none
    unsigned short* y_h;
};

// output row m -> (image n, oh, ow)
__device__ __forceinline__ void split_row(const ConvArgs& a, int m, int& n, int& oh, int& ow) {
    if (a.ohw_sh >= 0) {            // uniform
        n = m >> a.ohw_sh;
        const int rem = m & (a.OHW - 1);
        oh = rem >> a.ow_sh;
        ow = rem & (a.OW - 1);
    } else {
        n = m / a.OHW;
        const int rem = m - n * a.OHW;
        oh = rem / a.OW;
        ow = rem - oh * a.OW;
    }
}

__device__ __forceinline__ float bn_epilogue(const ConvArgs& a, float v, int m, int n) {
    v = fmaf(v, a.ep_scale[n], a.ep_shift[n]);
    if (a.ep_res) {
        const int cs = n - ((a.K - a.ep_cs) >> 1);
        if ((unsigned)cs < (unsigned)a.ep_cs) v += a.ep_res[(size_t)m * a.ep_cs + cs];
    }
    return (a.ep_alpha >= 0.f && v < 0.f) ? v * a.ep_alpha : v;
}

__device__ __forceinline__ size_t out_row(const ConvArgs& a, int m, bool scatter) {
    if (!scatter) return (size_t)m * a.K;
    int n, oh, ow;
    split_row(a, m, n, oh, ow);
    return ((size_t)(n * a.o_H + a.o_h0 + oh * a.o_s) * a.o_W + a.o_w0 + ow * a.o_s) * a.K;
}

// virtual coordinate -> real coordinate; returns false when the tap reads zero.  Branch-free on purpose (selects only):
// a scalar branch here would split the main-loop body into basic blocks and pin the address arithmetic in front of the MFMAs.
// UPS: the input is zero-upsampled by `ups` (dgrad of a strided convolution); sym: tf.pad SYMMETRIC mirror (edge included).
template <bool UPS>
__device__ __forceinline__ bool map_coord(int v, int H, int ups, int sym, int& i) {
    const int vs = v < 0 ? -1 - v : (v >= H ? 2 * H - 1 - v : v);
    const int vv = sym ? vs : v;
    if constexpr (UPS) {
        const int q = vv / ups;
        i = q;
        return (vv >= 0) & (q * ups == vv) & (q < H);
    } else {
        i = vv;
        return (unsigned)vv < (unsigned)H;
    }
}

// bijective XCD-aware remap: consecutive tiles (which share the A rows / filter panel) land on one XCD's L2
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int NX = 8;
    int q = nblk / NX, r = nblk % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Logical workgroup id -> (pixel tile, filter tile).  Plain order is filter-tile fastest; with gn > 0 the filter tiles are walked in
// groups of gn (a "super-column"): all pixel tiles of one group before the next, so the workgroups in flight on an XCD touch at most gn
// filter panels however wide the layer is (512->2560: 20 panels in flight without it).
__device__ __forceinline__ void tile_coords(int bid, int nblk_m, int nblk_n, int gn, int& mt, int& nt) {
    if (gn <= 0 || gn >= nblk_n) { mt = bid / nblk_n; nt = bid - mt * nblk_n; return; }
    const int span = nblk_m * gn;
    const int sc = bid / span, rem = bid - sc * span;
    const int left = nblk_n - sc * gn;
    const int width = left < gn ? left : gn;
    mt = rem / width;
    nt = sc * gn + rem - mt * width;
}

template <int TM, int TN>
struct Acc {
    f32x16 v[TM][TN];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) v[i][j][e] = 0.f;
    }
};

// ---- global loads go through buffer descriptors ---------------------------------------------------
// An out-of-range byte offset returns 0 in hardware, so TF zero padding, ragged tile edges, the k tail and the
// (unused) prefetch past the last stage need neither branches nor selects: the main-loop body is ONE basic block,
// which is what lets the MFMA / ds_read / buffer_load interleave below be scheduled at all.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const float __attribute__((address_space(4))) pnp_cfloat;     // constant address space: uniform reads become s_load
constexpr unsigned OOB = 0xFFFFFF00u;   // beyond any legal offset (host checks tensors are < 2^30 elements)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, off, 0, 0));
}
// a store whose byte offset is past the end of the buffer is DROPPED by the hardware: masked lanes need no branch around the store
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t r, unsigned off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, off, 0, 0);
}


// Epilogue of the forward / data-gradient kernels.  C/D layout of the 32x32 MFMAs: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// dropout (counter hash on the flat output index) -> optional BN statistics partials -> optional fused inference BN -> store
// (plain rows, or scattered to a stride phase's pixels).  part = index of this wave's row block among all (pixel tile, wave row) pairs.
// The epilogue runs in PHASES, each a loop over the wave's 16*TM*TN values with its (uniform) feature test OUTSIDE the loop: first every
// phase that reads memory or only computes (dropout, residual add, statistics, fused inference BN + shortcut + activation), then ONE phase
// of pure stores.  Round 3's stage trace (tools/experiments/stage_trace.py) showed why: with loads and stores interleaved per value the
// compiler fenced every store with s_waitcnt vmcnt(0) (a later load may alias it), i.e. one memory round trip PER VALUE — the epilogue of a
// 128x64 tile took as long as its 18-stage main loop (44.6k vs 40.7k clocks), 24 % of a 72-stage tile's life, 13 % of a 128x128 tile's.
template <int TM, int TN>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, const Acc<TM, TN>& acc, float* __restrict__ yout, int m0, int n0, int wm0,
                                              int wn0, int lane, int part, bool first_split = true) {
    const float* __restrict__ res = first_split ? a.res_add : nullptr;
    const int l31 = lane & 31, h = lane >> 5;
    const bool scatter = a.o_s != 0 && a.nsplit == 1;                  // split partials stay row-major; the reduce kernel scatters them
    const bool stats = a.stat_ws != nullptr;
    Acc<TM, TN> o = acc;                      // the values on their way out (register copy)
    int ncol[TN];
    bool cok[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        ncol[tn] = n0 + wn0 + tn * 32 + l31;
        cok[tn] = ncol[tn] < a.K;
    }
    const int mbase = m0 + wm0 + 4 * h;
#define PNP_EP_FOR                                                      \
    _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                   \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)               \
            _Pragma("unroll") for (int r = 0; r < 16; ++r)
#define PNP_EP_M (mbase + tm * 32 + (r & 3) + 8 * (r >> 2))
    // ---- dropout (counter hash on the flat output index)
    if (a.do_drop) {
        const uint32_t dkey = pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid);
        PNP_EP_FOR {
            const uint32_t idx = (uint32_t)((size_t)PNP_EP_M * a.K + ncol[tn]);
            o.v[tm][tn][r] = pnp_drop_keep(idx, dkey, a.drop_thresh) ? o.v[tm][tn][r] / a.drop_keep : 0.f;
        }
    }
    // ---- residual add (data gradients: the gradient that reaches the same tensor through a shortcut): all loads, then all adds
    if (res) {
        const __amdgpu_buffer_rsrc_t rr = make_rsrc(res, (unsigned)((size_t)a.M * a.K * 4));     // [M][K] rows like the output
        Acc<TM, TN> rv;
        PNP_EP_FOR {
            const int m = PNP_EP_M;
            rv.v[tm][tn][r] = bload1(rr, ((m < a.M) & cok[tn]) ? (unsigned)(m * a.K + ncol[tn]) * 4u : OOB);      // out of range: 0
        }
        PNP_EP_FOR o.v[tm][tn][r] += rv.v[tm][tn][r];
    }
    // ---- batch-norm statistics partials of the (post-dropout) output
    if (stats) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const float shift = (a.stat_shift && cok[tn]) ? a.stat_shift[ncol[tn]] : 0.f;
            float ssum = 0.f, ssq = 0.f;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = (PNP_EP_M < a.M) ? o.v[tm][tn][r] - shift : 0.f;
                    ssum += d;
                    ssq = fmaf(d, d, ssq);
                }
            const float s = ssum + __shfl_xor(ssum, 32, 64);     // the two half-waves hold the same column, disjoint rows
            const float q = ssq + __shfl_xor(ssq, 32, 64);
            if (h == 0 && cok[tn]) {
                a.stat_ws[((size_t)part * 2 + 0) * a.K + ncol[tn]] = s;
                a.stat_ws[((size_t)part * 2 + 1) * a.K + ncol[tn]] = q;
            }
        }
    }
    // ---- fused inference-mode BN (+ shortcut, channel zero-padded) + leaky-ReLU
    if (a.ep_scale) {
        float sc[TN], sh[TN];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            sc[tn] = cok[tn] ? a.ep_scale[ncol[tn]] : 0.f;
            sh[tn] = cok[tn] ? a.ep_shift[ncol[tn]] : 0.f;
        }
        PNP_EP_FOR o.v[tm][tn][r] = fmaf(o.v[tm][tn][r], sc[tn], sh[tn]);
        if (a.ep_res) {
            const int cpad = (a.K - a.ep_cs) >> 1;
            Acc<TM, TN> rv;
            PNP_EP_FOR {
                const int m = PNP_EP_M;
                const int cs = ncol[tn] - cpad;
                const bool ok = (m < a.M) & ((unsigned)cs < (unsigned)a.ep_cs);
                const float x = a.ep_res[ok ? (size_t)m * a.ep_cs + cs : 0];
                rv.v[tm][tn][r] = ok ? x : 0.f;
            }
            PNP_EP_FOR o.v[tm][tn][r] += rv.v[tm][tn][r];
        }
        if (a.ep_alpha >= 0.f) {
            PNP_EP_FOR o.v[tm][tn][r] = o.v[tm][tn][r] < 0.f ? o.v[tm][tn][r] * a.ep_alpha : o.v[tm][tn][r];
        }
    }
    // ---- stores, nothing else (plain rows, or scattered to a stride phase's pixels).  Buffer stores: a masked value gets an offset past
    // the end of the tensor and the hardware drops it — no exec-mask branch per value, 32-bit offsets (host: tensors < 2^30 elements)
    const size_t yelems = scatter ? (size_t)a.N * a.o_H * a.o_W * a.K : (size_t)a.M * a.K;
    const __amdgpu_buffer_rsrc_t ry = make_rsrc(yout, (unsigned)(yelems * 4));
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = PNP_EP_M;
            const bool mok = m < a.M;
            const unsigned row = scatter ? (unsigned)out_row(a, mok ? m : 0, true) : (unsigned)(m * a.K);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bstore1(ry, (mok & cok[tn]) ? (row + ncol[tn]) * 4u : OOB, o.v[tm][tn][r]);
        }
    if (a.y_h && !scatter) {                  // the same values rounded to bf16 (round-to-nearest-even): 64 contiguous bytes per 32 lanes
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.y_h, 0, (unsigned)(yelems * 2), 0x00020000);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = PNP_EP_M;
                const bool mok = m < a.M;
                const unsigned row = (unsigned)(m * a.K);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const __bf16 hv = (__bf16)o.v[tm][tn][r];
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hv), rh, (mok & cok[tn]) ? (row + ncol[tn]) * 2u : OOB, 0, 0);
                }
            }
    }
#undef PNP_EP_FOR
#undef PNP_EP_M
}

// Epilogue of the filter-gradient kernels: the wave's 32x32 accumulator tiles -> dW rows [m' = (tap, channel)][k] (or this split's
// partial); a.accumulate adds to what is there (un-split launches writing straight into a gradient that already holds a contribution).
template <int TM, int TN>
__device__ __forceinline__ void wgrad_epilogue(const ConvArgs& a, const Acc<TM, TN>& acc, float* __restrict__ out, int mm0, int n0, int wm0, int wn0,
                                               int lane) {
    const int l31 = lane & 31, h = lane >> 5;
    Acc<TM, TN> o = acc;
    const int mbase = mm0 + wm0 + 4 * h;
    if (a.accumulate) {            // (loads first, then stores only: see conv_epilogue)
        Acc<TM, TN> old;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const int n = n0 + wn0 + tn * 32 + l31;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                    const bool ok = (m < a.Kred) & (n < a.K);
                    old.v[tm][tn][r] = out[ok ? (size_t)m * a.K + n : 0];
                }
            }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) o.v[tm][tn][r] += old.v[tm][tn][r];
    }
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn0 + tn * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + tm * 32 + (r & 3) + 8 * (r >> 2);
                if ((m < a.Kred) & (n < a.K)) out[(size_t)m * a.K + n] = o.v[tm][tn][r];
            }
        }
}

constexpr unsigned OOB2 = 0x80000000u;     // host guarantees both tensors are < 2 GiB on this path

__device__ __forceinline__ f32x4 bload4s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ float bload1s(__amdgpu_buffer_rsrc_t r, unsigned voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

// bf16-operand kernels (conv_bf16.hip), dispatched from conv_igemm.hip's planners when ConvArgs::dtype == PNP_DTYPE_BF16 and the
// layer is on the tap-unrolled / linear-wgrad fast paths; every other layer keeps the fp32 kernels.  grid / nsplit / chunks_per_split are
// planned by the caller exactly as for the fp32 kernels.  tile: 0 = 128x128, 1 = 128x64, 2 = 128x32.  Return false: no instance.
bool launch_taps_bf16(const ConvArgs& a, int tile, int kind, dim3 grid, hipStream_t st);
bool launch_wgrad_bf16(const ConvArgs& a, int tile, dim3 grid, hipStream_t st);
// bf16-RESIDENT filter gradient (conv_bf16r.hip): a.x / a.w point at bf16 tensors; reduction chunks of 64 pixels
constexpr int kWgradBf16rChunk = 64;
bool launch_wgrad_bf16r(const ConvArgs& a, int tile, dim3 grid, hipStream_t st);
int wgrad_bf16r_tile(const pnp_conv_geom* g);
// stride-phase decomposition of a strided data gradient (conv_igemm.hip): dx[h] only receives taps r = h + pad (mod stride), so each
// residue class is a stride-1 convolution of dy with a sub-filter, its rows scattered with pixel stride `stride`
struct DgradPhase {
    int pa, pb, T, U, h0, w0, I, J, pad_t, pad_l;
    size_t wt_off;      // float offset of this phase's flipped filter [T][U][K][C] in the workspace (fp32 path)
};
int plan_phases(const pnp_conv_geom* g, DgradPhase* ph);
pnp_conv_geom phase_geom(const pnp_conv_geom* g, const DgradPhase& p);
// 3x3 stride-1 convolutions with exactly 16 output channels and 16 / 32 input channels on the 16x16x4 MFMA (conv_small.hip): forward,
// data gradient (kind 1; honours res_add) and filter gradient (per-workgroup partials [n16_wgrad_blocks][9*C][16] -> splitk_reduce_many)
bool n16_geom_ok(const pnp_conv_geom* g);
bool n16_wgrad_ok(const pnp_conv_geom* g);       // superset: filter gradients of 16/32 -> 32/64 channel layers too
int launch_n16_fwd(const ConvArgs& a, int kind, hipStream_t st);
int n16_wgrad_blocks(const pnp_conv_geom* g);
int launch_n16_wgrad(const ConvArgs& a, float* part, hipStream_t st);
// Winograd F(2x2, 3x3) route of the wide stride-1 3x3 convolutions, forward and data gradient (conv_wino.hip).  eligible: the geometry can
// run it; chosen: eligible and the policy takes it (PNP_WINOGRAD = 0 never / 1 where the cost model says it pays / 2 wherever eligible).
// launch_wino honours the whole epilogue of ConvArgs (dropout, residual add, statistics partials, fused inference BN); its statistics
// partial rows are the tile slabs of the output transform (wino_stats_parts), not the direct kernel's wave rows.
// Round 5: the route has two output tiles, F(2x2, 3x3) and F(4x4, 3x3) (36 instead of 64 multiplications per 4x4 outputs; PNP_WINOGRAD_TILE /
// pnp_conv2d_wino_tile: the largest the planner may pick).  wino_tile: 0 (direct kernels), 2 or 4 for the forward of this geometry.
bool wino_eligible(const pnp_conv_geom* g);
// conv_x3_direct.hip: direct split-bf16 3x3 convolutions of the narrow layers; launch_wino() hands over what x3d_chosen() takes
bool x3d_chosen(const pnp_conv_geom* g);
bool x3d_chosen(const ConvArgs& a);
int x3d_stats_parts(const pnp_conv_geom* g);
size_t x3d_filter_bytes(int C, int K);
int launch_x3_direct(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st);
bool wino_chosen(const pnp_conv_geom* g);
int wino_tile(const pnp_conv_geom* g);
size_t wino_workspace_bytes(const pnp_conv_geom* g);
int wino_stats_parts(const pnp_conv_geom* g);
int launch_wino(const ConvArgs& a, int kind, bool flip_transpose, void* ws, size_t ws_bytes, hipStream_t st);
// the filter gradient on the same route (its own switch, PNP_WINOGRAD_WGRAD): a = make_args(x, dy, -, g) of the forward geometry
bool wino_wgrad_chosen(const pnp_conv_geom* g);
int wino_wgrad_tile(const pnp_conv_geom* g);
size_t wino_wgrad_workspace_bytes(const pnp_conv_geom* g);
int launch_wino_wgrad(const ConvArgs& a, float* dw, int accumulate, void* ws, size_t ws_bytes, hipStream_t st);
// split-bf16 GEMMs of the route (conv_wino_x3.hip): M[pos] = V3[pos] x U3[pos]^T, V3 [npos][3][T][C] / U3 [npos][3][K][C] bf16 planes
// (hi, mid, lo: their sum is the fp32 value), M [npos][T][K] fp32.  dims_ok: one buffer descriptor per operand and transform point.
// X3_SW = reduction elements (channels; tiles for the filter gradient) per STAGE of the GEMM = the innermost extent of the operand layout
// [pos][red / X3_SW][plane][rows][X3_SW].  32 (shipped): LDS rows of 64 bytes, two 16-deep MFMA slices per stage, three LDS stages of 48 KB.
// 16 (-DPNP_X3_SW=16; measured, tools/experiments/README.md round 6): rows of 32 bytes, six stages of 24 KB — the GEMM is no faster (108.0
// vs 107.7 us on 512->512: it is not the bytes in flight that bound it) and the input transform, writing 32-byte runs, is slower (30 -> 42 us).
#ifndef PNP_X3_SW
#define PNP_X3_SW 32
#endif
constexpr int X3_SW = PNP_X3_SW;
static_assert(X3_SW == 16 || X3_SW == 32, "stage width");
bool wino_x3_dims_ok(int T, int C, int K);
// sym 0..3: forward / data gradient of F(2x2) / F(4x4) (nsplit = 1, stages_per_split = C / 32); 4 / 5: the filter gradient's GEMMs
// (T := channels, C := tiles padded to a multiple of 64, reduction split nsplit ways: M = [nsplit][npos][T][K])
int launch_wino_gemm_x3(const unsigned short* V3, const unsigned short* U3, float* Mm, int T, int C, int K, int npos, int sym, int gn, int xcd,
                        int nsplit, int stages_per_split, hipStream_t st);
inline double conv_flops(const ConvArgs& a) { return 2.0 * (double)a.M * a.K * a.Kred; }
inline double conv_bytes(const ConvArgs& a) {
    return 4.0 * ((double)a.N * a.H * a.W * a.C + (double)a.M * a.K + (double)a.Kred * a.K);
}
inline int prof_class(int kind) { return kind == 0 ? PNP_PROF_CONV_FWD : PNP_PROF_CONV_DGRAD; }

}  // namespace pnpconv
