// elementwise.hip — the HBM-bound kernels of the hot path: batch-norm (+dropout-mask, +residual, +leaky-ReLU),
// 2x2 max-pool, phase-shift (PS) upsampling, critic-input assembly, dropout, axpby.
// Reference call sites: layers.py:95-100 (batch_norm), 145-189 (residual_block / DR_block tails),
// 102-103 (max_pool2d), ops.py:3-27 (PS), adversarial.py:325-335 (critic input), layers.py:25,74,93 (dropout).
// All tensors are [P][C] fp32 with C contiguous; every kernel moves 16 B per lane where C % 4 == 0.
#include <atomic>
#include <mutex>
#include "pnp_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
// four values rounded to bf16 (nearest-even), one 8-byte store: the bf16 copies the resident convolutions read (conv_bf16r.hip)
__device__ __forceinline__ void st4h(__bf16* p, f32x4 v) {
    bf16x4_t h;
    h[0] = (__bf16)v[0]; h[1] = (__bf16)v[1]; h[2] = (__bf16)v[2]; h[3] = (__bf16)v[3];
    *reinterpret_cast<bf16x4_t*>(p) = h;
}

// grid of the thread-owns-a-channel-quad kernels: x = row slabs (rpi rows per pass, <= cap workgroups over all channel slices), y = slices of 256 quads
inline dim3 grid_rows(long long P, int C4, int cap = 256 * 8) {
    const int ny = (C4 + NT - 1) / NT;
    const int c4s = C4 < NT ? C4 : NT;
    const int rpi = NT / c4s;
    long long bx = (P + rpi - 1) / rpi;
    const long long cx = cap / ny > 0 ? cap / ny : 1;
    if (bx > cx) bx = cx;
    if (bx < 1) bx = 1;
    return dim3((unsigned)bx, (unsigned)ny);
}

inline int grid_for(size_t nvec, int cap = 256 * 8) {
    long long b = (long long)((nvec + NT - 1) / NT);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ------------------------------------------------------------------------------------------------
// Column reductions over [P][C]: each block reduces a slab of rows for every channel.
// partial layout: ws[(blk * 2 + q) * C + c], q in {0,1}
// KIND 0: BN statistics  q0 = sum(x - s), q1 = sum((x-s)^2) with shift s = x[0][c]
// KIND 1: BN backward    q0 = sum(dz),    q1 = sum(dz * xhat)
struct ColArgs {
    const float* x;      // KIND0: x ; KIND1: x (pre-BN input)
    const float* dout;   // KIND1
    const float* out;    // KIND1 (post activation), may be null when alpha<0
    const float* mean;   // KIND1
    const float* var;    // KIND1
    const float* gamma;  // KIND1, out == null: the activation's sign is recomputed from x (needs gamma, beta)
    const float* beta;
    float* ws;
    long long P;
    int C;
    int rows_per_block;
    float eps, alpha;
    // one-launch combine (round 5): tick != null -> the LAST workgroup of every slab of COLRED_SLAB partial rows sums its slab (double,
    // fixed order), and the last of those sums the slabs and writes the results — no colreduce_compact / colreduce_final launch.
    // tick: nslab slab counters + 1 top counter, zero on entry and left zero (self-resetting)
    unsigned* tick;
    int nslab;
    float* o0;           // KIND0: mean ; KIND1: dbeta
    float* o1;           // KIND0: var  ; KIND1: dgamma
    float* acc0;         // KIND1: += (gradient arena slots), nullable
    float* acc1;
    float* mm;           // KIND0: moving mean / variance update, nullable
    float* mv;
    float decay;
};

constexpr int COLRED_SLAB = 128;

template <int KIND>
__global__ void __launch_bounds__(NT) colreduce_kernel(ColArgs a) {
    __shared__ float red[NT * 8];
    __shared__ int s_last;
    const int C4 = a.C >> 2;
    const int rpi = NT / C4;                 // rows per iteration (C4 <= 256 guaranteed by the host)
    const int t = threadIdx.x;
    const int cg = t % C4, rsub = t / C4;
    const bool active = rsub < rpi;
    const long long r0 = (long long)blockIdx.x * a.rows_per_block;
    long long r1 = r0 + a.rows_per_block;
    if (r1 > a.P) r1 = a.P;
    f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
    if (active) {
        const int c = cg * 4;
        f32x4 m, rs, gsc = {0, 0, 0, 0}, bet = {0, 0, 0, 0};
        if constexpr (KIND == 0) {
            m = ld4(a.x + c);   // shift
        } else {
            m = ld4(a.mean + c);
            f32x4 v = ld4(a.var + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) rs[e] = 1.0f / sqrtf(v[e] + a.eps);
            if (!a.out && a.alpha >= 0.f) {
                const f32x4 ga = ld4(a.gamma + c);
                bet = ld4(a.beta + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) gsc[e] = ga[e] * rs[e];
            }
        }
        for (long long r = r0 + rsub; r < r1; r += rpi) {
            const size_t off = (size_t)r * a.C + c;
            f32x4 xv = ld4(a.x + off);
            if constexpr (KIND == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float d = xv[e] - m[e];
                    s0[e] += d;
                    s1[e] = fmaf(d, d, s1[e]);
                }
            } else {
                f32x4 g = ld4(a.dout + off);
                if (a.alpha >= 0.f) {
                    if (a.out) {
                        f32x4 o = ld4(a.out + off);
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : g[e] * a.alpha;
                    } else {        // the pre-activation value exactly as bn_apply_kernel formed it
#pragma unroll
                        for (int e = 0; e < 4; ++e) g[e] = fmaf(xv[e] - m[e], gsc[e], bet[e]) > 0.f ? g[e] : g[e] * a.alpha;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float xh = (xv[e] - m[e]) * rs[e];
                    s0[e] += g[e];
                    s1[e] = fmaf(g[e], xh, s1[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        red[t * 8 + e] = s0[e];
        red[t * 8 + 4 + e] = s1[e];
    }
    __syncthreads();
    if (t < C4) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int j = 0; j < rpi; ++j) {
            const int tt = j * C4 + t;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a0[e] += red[tt * 8 + e];
                a1[e] += red[tt * 8 + 4 + e];
            }
        }
        float* w0 = a.ws + ((size_t)blockIdx.x * 2 + 0) * a.C + t * 4;
        float* w1 = a.ws + ((size_t)blockIdx.x * 2 + 1) * a.C + t * 4;
        st4(w0, a0);
        st4(w1, a1);
    }
    if (!a.tick) return;
    // ---- one-launch combine.  Release: the partial row is visible device-wide before the ticket is taken; acquire: the last taker sees
    // every row of its slab.  Summation orders are fixed (rows of a slab by (row lane, stride), slabs in order): deterministic whichever
    // workgroup happens to be last.
    const int nblk = (int)gridDim.x;
    const int slab = blockIdx.x / COLRED_SLAB;
    const int b0 = slab * COLRED_SLAB;
    const int b1 = (b0 + COLRED_SLAB < nblk) ? b0 + COLRED_SLAB : nblk;
    __threadfence();
    __syncthreads();
    if (t == 0) s_last = (atomicAdd(&a.tick[slab], 1u) == (unsigned)(b1 - b0 - 1));
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    double* dred = reinterpret_cast<double*>(red);          // NT x 4 doubles (the float scratch, reused): two passes, q0 then q1
    double d0[4] = {0, 0, 0, 0}, d1[4] = {0, 0, 0, 0};
    if (active) {
        for (int b = b0 + rsub; b < b1; b += rpi) {
            const f32x4 v0 = ld4(a.ws + ((size_t)b * 2 + 0) * a.C + cg * 4), v1 = ld4(a.ws + ((size_t)b * 2 + 1) * a.C + cg * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d0[e] += (double)v0[e];
                d1[e] += (double)v1[e];
            }
        }
    }
    double t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) dred[t * 4 + e] = q == 0 ? d0[e] : d1[e];
        __syncthreads();
        if (t < C4) {
            for (int j = 0; j < rpi; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) (q == 0 ? t0 : t1)[e] += dred[(j * C4 + t) * 4 + e];
        }
    }
    // the slab's sum as a (high, low) pair of floats over its own first two rows (colreduce_compact_kernel's layout): only this workgroup
    // reads or writes them from here on (a slab of one row keeps the high part only: its low part is exactly zero)
    if (t < C4) {
        f32x4 h0, h1, l0, l1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h0[e] = (float)t0[e];
            h1[e] = (float)t1[e];
            l0[e] = (fabsf(h0[e]) <= 3.4e38f) ? (float)(t0[e] - (double)h0[e]) : 0.f;
            l1[e] = (fabsf(h1[e]) <= 3.4e38f) ? (float)(t1[e] - (double)h1[e]) : 0.f;
        }
        st4(a.ws + ((size_t)b0 * 2 + 0) * a.C + t * 4, h0);
        st4(a.ws + ((size_t)b0 * 2 + 1) * a.C + t * 4, h1);
        if (b1 - b0 > 1) {
            st4(a.ws + ((size_t)(b0 + 1) * 2 + 0) * a.C + t * 4, l0);
            st4(a.ws + ((size_t)(b0 + 1) * 2 + 1) * a.C + t * 4, l1);
        }
    }
    __threadfence();
    __syncthreads();
    if (t == 0) {
        a.tick[slab] = 0u;                                   // every workgroup of the slab has taken its ticket: re-arm
        s_last = (atomicAdd(&a.tick[a.nslab], 1u) == (unsigned)(a.nslab - 1));
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    if (t == 0) a.tick[a.nslab] = 0u;
    if (t >= C4) return;
    double f0[4] = {0, 0, 0, 0}, f1[4] = {0, 0, 0, 0};
    for (int sl = 0; sl < a.nslab; ++sl) {
        const int r0s = sl * COLRED_SLAB;
        const int nrow = (r0s + 1 < nblk) ? 2 : 1;
        for (int j = 0; j < nrow; ++j) {
            const f32x4 v0 = ld4(a.ws + ((size_t)(r0s + j) * 2 + 0) * a.C + t * 4), v1 = ld4(a.ws + ((size_t)(r0s + j) * 2 + 1) * a.C + t * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                f0[e] += (double)v0[e];
                f1[e] += (double)v1[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = t * 4 + e;
        if constexpr (KIND == 0) {
            const double inv = 1.0 / (double)a.P;
            const double md = f0[e] * inv;
            double var = f1[e] * inv - md * md;
            if (var < 0.0) var = 0.0;
            const float meanf = (float)((double)a.x[c] + md), varf = (float)var;
            a.o0[c] = meanf;
            a.o1[c] = varf;
            if (a.mm) {
                const float one_minus = 1.0f - a.decay;
                const float bessel = a.P > 1 ? (float)((double)a.P / (double)(a.P - 1)) : 1.0f;
                a.mm[c] -= (a.mm[c] - meanf) * one_minus;
                a.mv[c] -= (a.mv[c] - varf * bessel) * one_minus;
            }
        } else {
            a.o0[c] = (float)f0[e];
            a.o1[c] = (float)f1[e];
            if (a.acc0) {
                a.acc0[c] += (float)f0[e];
                a.acc1[c] += (float)f1[e];
            }
        }
    }
}

// scalar fallback for C % 4 != 0 or C > 1024: one thread per channel per block-slab
template <int KIND>
__global__ void colreduce_scalar_kernel(ColArgs a) {
    const int c = blockIdx.y * blockDim.x + threadIdx.x;
    if (c >= a.C) return;
    const long long r0 = (long long)blockIdx.x * a.rows_per_block;
    long long r1 = r0 + a.rows_per_block;
    if (r1 > a.P) r1 = a.P;
    float s0 = 0.f, s1 = 0.f;
    float m, rs = 0.f;
    if constexpr (KIND == 0) m = a.x[c];
    else { m = a.mean[c]; rs = 1.0f / sqrtf(a.var[c] + a.eps); }
    for (long long r = r0; r < r1; ++r) {
        const size_t off = (size_t)r * a.C + c;
        float xv = a.x[off];
        if constexpr (KIND == 0) {
            float d = xv - m;
            s0 += d;
            s1 = fmaf(d, d, s1);
        } else {
            float g = a.dout[off];
            if (a.alpha >= 0.f) {
                const float o = a.out ? a.out[off] : fmaf(xv - m, a.gamma[c] * rs, a.beta[c]);
                g = o > 0.f ? g : g * a.alpha;
            }
            s0 += g;
            s1 = fmaf(g, (xv - m) * rs, s1);
        }
    }
    a.ws[((size_t)blockIdx.x * 2 + 0) * a.C + c] = s0;
    a.ws[((size_t)blockIdx.x * 2 + 1) * a.C + c] = s1;
}

// Long partial lists (the convolution epilogue leaves one partial per 64 output rows: 16 384 of them for a 64-channel layer at 256^2,
// which the C/32 = 2 workgroups of the final kernel walked alone: 240 us) are first compacted IN PLACE by (C/32) x nslab workgroups:
// slab s = partials [s*SLAB, (s+1)*SLAB) (the last slab takes the remainder) is summed in double and written back over its own first two
// partials as a (high, low) pair of floats, which together carry the double sum to 48 bits.  Only this workgroup ever reads those two
// rows of its 32 channels, so there is no race; the summation order is fixed => deterministic.  The partial list is consumed.
__global__ void __launch_bounds__(1024) colreduce_compact_kernel(float* __restrict__ ws, int nblk, int nslab, int C) {
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const int s = blockIdx.y;
    const int b0 = s * COLRED_SLAB;
    const int b1 = (s == nslab - 1) ? nblk : b0 + COLRED_SLAB;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        for (int b = b0 + sl; b < b1; b += 32) {
            s0 += (double)ws[((size_t)b * 2 + 0) * C + c];
            s1 += (double)ws[((size_t)b * 2 + 1) * C + c];
        }
    }
    red[0][sl][cl] = s0;
    red[1][sl][cl] = s1;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    s0 = 0.0;
    s1 = 0.0;
    for (int j = 0; j < 32; ++j) {
        s0 += red[0][j][cl];
        s1 += red[1][j][cl];
    }
    const float h0 = (float)s0, h1 = (float)s1;
    ws[((size_t)b0 * 2 + 0) * C + c] = h0;
    ws[((size_t)b0 * 2 + 1) * C + c] = h1;
    // (an overflowed sum stays +-inf like in the single-launch combine: inf - inf would turn it into NaN)
    ws[((size_t)(b0 + 1) * 2 + 0) * C + c] = (fabsf(h0) <= 3.4e38f) ? (float)(s0 - (double)h0) : 0.f;
    ws[((size_t)(b0 + 1) * 2 + 1) * C + c] = (fabsf(h1) <= 3.4e38f) ? (float)(s1 - (double)h1) : 0.f;
}

// final combine over blocks in double. KIND 0 -> mean,var ; KIND 1 -> dbeta (o0), dgamma (o1)
// slab > 0: the list was compacted; entry b is partial (b >> 1) * slab + (b & 1)
template <int KIND>
__global__ void __launch_bounds__(1024) colreduce_final_kernel(const float* __restrict__ ws, const float* __restrict__ x0,
                                                               float* o0, float* o1, int nblk, int C, long long P, float* mm, float* mv,
                                                               float decay, int slab, float* acc0, float* acc1) {
    // 1024 threads = 32 channels x 32 slices of the block list; fixed summation order => deterministic
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        for (int b = sl; b < nblk; b += 32) {
            const int row = slab > 0 ? (b >> 1) * slab + (b & 1) : b;
            s0 += (double)ws[((size_t)row * 2 + 0) * C + c];
            s1 += (double)ws[((size_t)row * 2 + 1) * C + c];
        }
    }
    red[0][sl][cl] = s0;
    red[1][sl][cl] = s1;
    __syncthreads();
    if (sl != 0 || c >= C) return;
    s0 = 0.0;
    s1 = 0.0;
    for (int j = 0; j < 32; ++j) {
        s0 += red[0][j][cl];
        s1 += red[1][j][cl];
    }
    if constexpr (KIND == 0) {
        const double inv = 1.0 / (double)P;
        const double md = s0 * inv;                 // mean of (x - shift)
        double var = s1 * inv - md * md;
        if (var < 0.0) var = 0.0;
        const float meanf = (float)((double)x0[c] + md), varf = (float)var;
        o0[c] = meanf;
        o1[c] = varf;
        if (mm) {       // tf.contrib.layers.batch_norm(updates_collections=None): the moving averages move with every training-mode forward
            const float one_minus = 1.0f - decay;
            const float bessel = P > 1 ? (float)((double)P / (double)(P - 1)) : 1.0f;
            mm[c] -= (mm[c] - meanf) * one_minus;
            mv[c] -= (mv[c] - varf * bessel) * one_minus;
        }
    } else {
        o0[c] = (float)s0;
        o1[c] = (float)s1;
        if (acc0) {         // the parameter gradients straight into the gradient arena (which may already hold another use's share)
            acc0[c] += (float)s0;
            acc1[c] += (float)s1;
        }
    }
}

// the final combine, in two launches when the list is long (>= 4 slabs)
template <int KIND>
int launch_colreduce_final(float* ws, const float* x0, float* o0, float* o1, int nblk, int C, long long P, float* mm, float* mv, float decay,
                           hipStream_t st, const char* who, float* acc0 = nullptr, float* acc1 = nullptr) {
    static const int two_stage = getenv("PNP_BN_FINAL_1STAGE") ? 0 : 1;
    int slab = 0, n = nblk;
    if (two_stage && nblk >= 4 * COLRED_SLAB) {
        const int nslab = nblk / COLRED_SLAB;
        hipLaunchKernelGGL(colreduce_compact_kernel, dim3(pnp_cdiv(C, 32), nslab), dim3(1024), 0, st, ws, nblk, nslab, C);
        PNP_CHECK_LAUNCH(who);
        slab = COLRED_SLAB;
        n = 2 * nslab;
    }
    hipLaunchKernelGGL(colreduce_final_kernel<KIND>, dim3(pnp_cdiv(C, 32)), dim3(1024), 0, st, (const float*)ws, x0, o0, o1, n, C, P, mm, mv,
                       decay, slab, acc0, acc1);
    PNP_CHECK_LAUNCH(who);
    return PNP_OK;
}

int colreduce_plan(long long P, int C, int* nblk, int* rpb) {
    // ~2048 blocks max, at least 64 rows per block
    static const int cap = getenv("PNP_BN_NBLK") ? atoi(getenv("PNP_BN_NBLK")) : 2048;
    long long b = (P + 63) / 64;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    long long r = (P + b - 1) / b;
    b = (P + r - 1) / r;
    *nblk = (int)b;
    *rpb = (int)r;
    (void)C;
    return 0;
}

// Ticket counters of the one-launch combine: a library-owned, zero-initialised device buffer of TICKET_SLOTS x TICKET_STRIDE counters,
// handed out round-robin (a slot is busy for the lifetime of ONE launch and re-arms itself: two launches share a slot only if
// TICKET_SLOTS launches are in flight at once).  Allocated at the first un-captured use (hipMalloc is not legal inside a stream
// capture: a recording that comes first falls back to the separate combine launches).
// OFF by default (PNP_BN_ONE_LAUNCH=1 switches it on): measured within one run on the joint step (round 5: tools/experiments/README.md), 242.4 ->
// 208.3 slices/s — colreduce_kernel<1> 32 -> 141 us per launch: the device-scope release / acquire every workgroup needs around its
// ticket (buffer_wbl2 + buffer_inv sc1 on a part whose 8 L2s are only coherent through them) costs far more than the two ~8 us combine
// launches it removes.  Kept as the measured answer to VERDICT r4 #5c; parity-tested (tests/test_gpu_elementwise.py).
constexpr int TICKET_SLOTS = 512, TICKET_STRIDE = 32;
unsigned* ticket_slot(hipStream_t st, int nslab) {
    static const int on = getenv("PNP_BN_ONE_LAUNCH") ? atoi(getenv("PNP_BN_ONE_LAUNCH")) : 0;
    if (!on || nslab + 1 > TICKET_STRIDE) return nullptr;
    // one buffer per DEVICE (a process that drives several GPUs must not hand device 0's tickets to a launch on device 1)
    constexpr int MAXDEV = 16;
    static std::atomic<unsigned*> bufs[MAXDEV];
    static std::atomic<unsigned> seq{0};
    static std::mutex mx;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAXDEV) {
        (void)hipGetLastError();
        return nullptr;
    }
    std::atomic<unsigned*>& buf = bufs[dev];
    unsigned* b = buf.load(std::memory_order_acquire);
    if (!b) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
            (void)hipGetLastError();
            return nullptr;
        }
        std::lock_guard<std::mutex> lk(mx);
        b = buf.load(std::memory_order_acquire);
        if (!b) {
            if (hipMalloc((void**)&b, sizeof(unsigned) * TICKET_SLOTS * TICKET_STRIDE) != hipSuccess ||
                hipMemset(b, 0, sizeof(unsigned) * TICKET_SLOTS * TICKET_STRIDE) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
                (void)hipGetLastError();
                return nullptr;
            }
            buf.store(b, std::memory_order_release);
        }
    }
    return b + (size_t)(seq.fetch_add(1, std::memory_order_relaxed) % TICKET_SLOTS) * TICKET_STRIDE;
}

template <int KIND>
int run_colreduce(ColArgs a, float* o0, float* o1, void* ws, size_t ws_bytes, hipStream_t st, const char* who, float* mm = nullptr,
                  float* mv = nullptr, float decay = 0.f, float* acc0 = nullptr, float* acc1 = nullptr) {
    int nblk, rpb;
    colreduce_plan(a.P, a.C, &nblk, &rpb);
    const size_t need = (size_t)nblk * 2 * a.C * sizeof(float);
    if (ws_bytes < need || !ws) {
        pnp_set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
        return PNP_EWORKSPACE;
    }
    a.ws = (float*)ws;
    a.rows_per_block = rpb;
    if ((a.C & 3) == 0 && a.C <= 1024) {
        a.nslab = pnp_cdiv(nblk, COLRED_SLAB);
        a.tick = ticket_slot(st, a.nslab);
        a.o0 = o0; a.o1 = o1; a.acc0 = acc0; a.acc1 = acc1; a.mm = mm; a.mv = mv; a.decay = decay;
        hipLaunchKernelGGL(colreduce_kernel<KIND>, dim3(nblk), dim3(NT), 0, st, a);
        if (a.tick) {
            PNP_CHECK_LAUNCH(who);
            return PNP_OK;
        }
    } else {
        hipLaunchKernelGGL(colreduce_scalar_kernel<KIND>, dim3(nblk, pnp_cdiv(a.C, 64)), dim3(64), 0, st, a);
    }
    PNP_CHECK_LAUNCH(who);
    return launch_colreduce_final<KIND>((float*)ws, a.x, o0, o1, nblk, a.C, a.P, mm, mv, decay, st, who, acc0, acc1);
}

__global__ void bn_update_moving_kernel(float* mm, float* mv, const float* mean, const float* var, long long P, int C,
                                        float decay) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float one_minus = 1.0f - decay;
    const float bessel = P > 1 ? (float)((double)P / (double)(P - 1)) : 1.0f;
    mm[c] -= (mm[c] - mean[c]) * one_minus;
    mv[c] -= (mv[c] - var[c] * bessel) * one_minus;
}

struct BnApplyArgs {
    const float *x, *mean, *var, *gamma, *beta, *shortcut;
    float* y;
    long long P;
    int C, Cs;
    float eps, alpha;
    __bf16* yh;      // bf16 copy of y (the next convolution's operand on the bf16-resident path); null: none
};

// VEC: a thread owns ONE channel quad and walks rows (block = rpi rows x C4s quads, blockIdx.y = slice of 256 quads): the per-channel
// coefficients — 1 / sqrt(var + eps) is a square root and a correctly rounded division — are formed once per thread, and no index needs a
// 64-bit division (round 6: the element-at-a-time form re-did both for every vector and ran VALU-bound at 1.8-2.4 TB/s on the critics'
// 537 MB maps).  Same expressions in the same order: bit-identical results.
template <bool VEC>
__global__ void __launch_bounds__(NT) bn_apply_kernel(BnApplyArgs a) {
    const int cpad = (a.C - a.Cs) / 2;
    if constexpr (VEC) {
        const int C4 = a.C >> 2;
        const int kq0 = blockIdx.y * NT;
        const int C4s = (C4 - kq0) < NT ? (C4 - kq0) : NT;
        const int rpi = NT / C4s;
        const int cg = threadIdx.x % C4s, rsub = threadIdx.x / C4s;
        if (rsub >= rpi) return;
        const int c = (kq0 + cg) * 4;
        const f32x4 m = ld4(a.mean + c), v = ld4(a.var + c), g = ld4(a.gamma + c), b = ld4(a.beta + c);
        f32x4 gsc;
#pragma unroll
        for (int e = 0; e < 4; ++e) gsc[e] = g[e] * (1.0f / sqrtf(v[e] + a.eps));
        const int cs = c - cpad;
        const bool has_s = a.shortcut && cs >= 0 && cs < a.Cs;
        // a workgroup streams ONE contiguous slab of rows (like colreduce_kernel, which reads the same tensors at 6 TB/s; strided passes of
        // 16 rows per workgroup ran at 2.7-3.6 TB/s on the 537 MB maps)
        const long long per = ((a.P + gridDim.x - 1) / gridDim.x + rpi - 1) / rpi * rpi;
        const long long rend = ((long long)blockIdx.x + 1) * per < a.P ? ((long long)blockIdx.x + 1) * per : a.P;
        for (long long row = (long long)blockIdx.x * per + rsub; row < rend; row += rpi) {
            const size_t i = (size_t)row * C4 + kq0 + cg;
            const f32x4 xv = ld4(a.x + i * 4);
            f32x4 r;
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = fmaf(xv[e] - m[e], gsc[e], b[e]);    // (explicit: the backward kernels recompute this value
            // bit for bit when they are not handed `out`)
            if (has_s) {
                const f32x4 sv = ld4(a.shortcut + (size_t)row * a.Cs + cs);
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] += sv[e];
            }
            if (a.alpha >= 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = r[e] > 0.f ? r[e] : r[e] * a.alpha;
            }
            st4(a.y + i * 4, r);
            if (a.yh) st4h(a.yh + i * 4, r);
        }
    } else {
        const int CV = a.C;
        const size_t nvec = (size_t)a.P * CV;
        const size_t gs = (size_t)gridDim.x * NT;
        for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += gs) {
            const size_t row = i / CV;
            const int c = (int)(i - row * CV);
            float r = fmaf(a.x[i] - a.mean[c], a.gamma[c] * (1.0f / sqrtf(a.var[c] + a.eps)), a.beta[c]);
            if (a.shortcut) {
                const int cs = c - cpad;
                if (cs >= 0 && cs < a.Cs) r += a.shortcut[row * a.Cs + cs];
            }
            if (a.alpha >= 0.f) r = r > 0.f ? r : r * a.alpha;
            a.y[i] = r;
            if (a.yh) a.yh[i] = (__bf16)r;
        }
    }
}

struct BnBwdArgs {
    const float *dout, *out, *x, *mean, *var, *gamma, *beta, *dgamma, *dbeta;      // out == null: sign recomputed from x (needs beta)
    float *dx, *dshortcut;
    long long P;
    long long P_norm;    // rows behind dgamma / dbeta: P, or the global row count when the sums were all-reduced (SyncBN)
    int C, Cs;
    float eps, alpha;
    int training;
    int do_drop;
    uint32_t drop_key, drop_thresh;
    float drop_keep;
    const pnp_step_params* sp;       // step capture: dropout seed from device memory (pnp_common.h)
    uint32_t drop_sid;
    __bf16* dxh;     // bf16 copy of dx (operand of the bf16-resident data / filter gradient kernels); null: none
};

// (VEC: the thread-owns-a-channel-quad mapping of bn_apply_kernel; same expressions in the same order as the element-at-a-time form)
template <bool VEC>
__global__ void __launch_bounds__(NT) bn_bwd_apply_kernel(BnBwdArgs a) {
    const int cpad = (a.C - a.Cs) / 2;
    const float invP = (float)(1.0 / (double)a.P_norm);
    const uint32_t dkey = a.do_drop ? pnp_eff_drop_key(a.drop_key, a.sp, a.drop_sid) : 0u;
    if constexpr (VEC) {
        const int C4 = a.C >> 2;
        const int kq0 = blockIdx.y * NT;
        const int C4s = (C4 - kq0) < NT ? (C4 - kq0) : NT;
        const int rpi = NT / C4s;
        const int cg = threadIdx.x % C4s, rsub = threadIdx.x / C4s;
        if (rsub >= rpi) return;
        const int c = (kq0 + cg) * 4;
        const f32x4 m = ld4(a.mean + c), v = ld4(a.var + c), ga = ld4(a.gamma + c);
        f32x4 rs, gsc, b = {0, 0, 0, 0}, dgp = {0, 0, 0, 0}, dbp = {0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rs[e] = 1.0f / sqrtf(v[e] + a.eps);
            gsc[e] = ga[e] * rs[e];
        }
        if (a.alpha >= 0.f && !a.out) b = ld4(a.beta + c);
        if (a.training) {
            const f32x4 dg = ld4(a.dgamma + c), db = ld4(a.dbeta + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                dgp[e] = dg[e] * invP;
                dbp[e] = db[e] * invP;
            }
        }
        const int cs = c - cpad;
        const bool has_s = a.dshortcut && cs >= 0 && cs < a.Cs;
        // a workgroup streams ONE contiguous slab of rows (like colreduce_kernel, which reads the same tensors at 6 TB/s; strided passes of
        // 16 rows per workgroup ran at 2.7-3.6 TB/s on the 537 MB maps)
        const long long per = ((a.P + gridDim.x - 1) / gridDim.x + rpi - 1) / rpi * rpi;
        const long long rend = ((long long)blockIdx.x + 1) * per < a.P ? ((long long)blockIdx.x + 1) * per : a.P;
        for (long long row = (long long)blockIdx.x * per + rsub; row < rend; row += rpi) {
            const size_t i = (size_t)row * C4 + kq0 + cg;
            f32x4 g = ld4(a.dout + i * 4);
            const f32x4 xv = ld4(a.x + i * 4);
            if (a.alpha >= 0.f) {
                if (a.out) {
                    const f32x4 o = ld4(a.out + i * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : g[e] * a.alpha;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) g[e] = fmaf(xv[e] - m[e], gsc[e], b[e]) > 0.f ? g[e] : g[e] * a.alpha;
                }
            }
            if (has_s) st4(a.dshortcut + (size_t)row * a.Cs + cs, g);
            f32x4 r;
            if (a.training) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (xv[e] - m[e]) * rs[e];
                    r[e] = gsc[e] * (g[e] - dbp[e] - xh * dgp[e]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = gsc[e] * g[e];
            }
            if (a.do_drop) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    r[e] = pnp_drop_keep((uint32_t)(i * 4 + e), dkey, a.drop_thresh) ? r[e] / a.drop_keep : 0.f;
            }
            if (a.dx) st4(a.dx + i * 4, r);       // (null: only the bf16 copy is wanted — dx feeds nothing but resident convolutions)
            if (a.dxh) st4h(a.dxh + i * 4, r);
        }
    } else {
        const int CV = a.C;
        const size_t nvec = (size_t)a.P * CV;
        const size_t gs = (size_t)gridDim.x * NT;
        for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += gs) {
            const size_t row = i / CV;
            const int c = (int)(i - row * CV);
            float g = a.dout[i];
            if (a.alpha >= 0.f) {
                const float o = a.out ? a.out[i] : fmaf(a.x[i] - a.mean[c], a.gamma[c] * (1.0f / sqrtf(a.var[c] + a.eps)), a.beta[c]);
                g = o > 0.f ? g : g * a.alpha;
            }
            if (a.dshortcut) {
                const int cs = c - cpad;
                if (cs >= 0 && cs < a.Cs) a.dshortcut[row * a.Cs + cs] = g;
            }
            float rs = 1.0f / sqrtf(a.var[c] + a.eps);
            float r;
            if (a.training) {
                float xh = (a.x[i] - a.mean[c]) * rs;
                r = a.gamma[c] * rs * (g - a.dbeta[c] * invP - xh * (a.dgamma[c] * invP));
            } else {
                r = a.gamma[c] * rs * g;
            }
            if (a.do_drop) r = pnp_drop_keep((uint32_t)i, dkey, a.drop_thresh) ? r / a.drop_keep : 0.f;
            if (a.dx) a.dx[i] = r;
            if (a.dxh) a.dxh[i] = (__bf16)r;
        }
    }
}

__global__ void __launch_bounds__(NT) dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                     uint32_t key, uint32_t thresh, float keep, __bf16* __restrict__ yh,
                                                     const pnp_step_params* sp, uint32_t sid) {
    key = pnp_eff_drop_key(key, sp, sid);
    const size_t gs = (size_t)gridDim.x * NT;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += gs) {
        f32x4 v = ld4(x + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pnp_drop_keep((uint32_t)(i * 4 + e), key, thresh) ? v[e] / keep : 0.f;
        if (y) st4(y + i * 4, v);
        if (yh) st4h(yh + i * 4, v);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += gs) {
        const float v = pnp_drop_keep((uint32_t)i, key, thresh) ? x[i] / keep : 0.f;
        if (y) y[i] = v;
        if (yh) yh[i] = (__bf16)v;
    }
}

__global__ void __launch_bounds__(NT) axpby_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n, float a,
                                                   float b) {
    const size_t gs = (size_t)gridDim.x * NT;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += gs) {
        f32x4 xv = ld4(x + i * 4), yv = ld4(y + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) yv[e] = a * xv[e] + b * yv[e];
        st4(y + i * 4, yv);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += gs) y[i] = a * x[i] + b * y[i];
}

// out = x + y: the sum TF's autodiff emits as AddN where a tensor feeds two branches (functional.FanOutFn)
__global__ void __launch_bounds__(NT) add_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, size_t n) {
    const size_t gs = (size_t)gridDim.x * NT;
    const size_t n4 = n >> 2;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += gs) {
        const f32x4 xv = ld4(x + i * 4), yv = ld4(y + i * 4);
        st4(out + i * 4, xv + yv);
    }
    for (size_t i = (n4 << 2) + (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += gs) out[i] = x[i] + y[i];
}

// ---- 2x2/2 max-pool -----------------------------------------------------------------------------
template <bool BWD>
__global__ void __launch_bounds__(NT) maxpool2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                      float* __restrict__ out, int N, int H, int W, int C) {
    const int OH = H >> 1, OW = W >> 1;
    const int CV = ((C & 3) == 0) ? (C >> 2) : C;
    const bool vec = (C & 3) == 0;
    const size_t total = (size_t)N * OH * OW * CV;
    const size_t gs = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += gs) {
        const int cv = (int)(i % CV);
        size_t q = i / CV;
        const int ow = (int)(q % OW);
        q /= OW;
        const int oh = (int)(q % OH);
        const int n = (int)(q / OH);
        const size_t base = (((size_t)n * H + 2 * oh) * W + 2 * ow) * C;
        const size_t o01 = (size_t)C, o10 = (size_t)W * C, o11 = (size_t)W * C + C;
        if (vec) {
            const int c = cv * 4;
            f32x4 v00 = ld4(x + base + c), v01 = ld4(x + base + o01 + c), v10 = ld4(x + base + o10 + c),
                  v11 = ld4(x + base + o11 + c);
            if constexpr (!BWD) {
                f32x4 r;
#pragma unroll
                for (int e = 0; e < 4; ++e) r[e] = fmaxf(fmaxf(v00[e], v01[e]), fmaxf(v10[e], v11[e]));
                st4(out + i * 4, r);
            } else {
                f32x4 g = ld4(dy + i * 4);
                f32x4 d00, d01, d10, d11;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // first maximal element in window scan order (row-major) gets the gradient
                    float m = v00[e];
                    int am = 0;
                    if (v01[e] > m) { m = v01[e]; am = 1; }
                    if (v10[e] > m) { m = v10[e]; am = 2; }
                    if (v11[e] > m) { m = v11[e]; am = 3; }
                    d00[e] = am == 0 ? g[e] : 0.f;
                    d01[e] = am == 1 ? g[e] : 0.f;
                    d10[e] = am == 2 ? g[e] : 0.f;
                    d11[e] = am == 3 ? g[e] : 0.f;
                }
                st4(out + base + c, d00);
                st4(out + base + o01 + c, d01);
                st4(out + base + o10 + c, d10);
                st4(out + base + o11 + c, d11);
            }
        } else {
            const int c = cv;
            float v00 = x[base + c], v01 = x[base + o01 + c], v10 = x[base + o10 + c], v11 = x[base + o11 + c];
            if constexpr (!BWD) {
                out[i] = fmaxf(fmaxf(v00, v01), fmaxf(v10, v11));
            } else {
                float g = dy[i];
                float m = v00;
                int am = 0;
                if (v01 > m) { m = v01; am = 1; }
                if (v10 > m) { m = v10; am = 2; }
                if (v11 > m) { m = v11; am = 3; }
                out[base + c] = am == 0 ? g : 0.f;
                out[base + o01 + c] = am == 1 ? g : 0.f;
                out[base + o10 + c] = am == 2 ? g : 0.f;
                out[base + o11 + c] = am == 3 ? g : 0.f;
            }
        }
    }
}

// ---- PS (phase shift): out[n, i*r+u, j*r+v, c] = x[n,i,j, c*r*r + v*r + u] ----------------------
// One thread per OUTPUT element (fwd) / per INPUT-GRAD element (bwd); the thread index runs over the
// output layout so stores are coalesced; loads hit the same 10 KB input pixel row (L1/L2 resident).
template <bool BWD>
__global__ void __launch_bounds__(NT) ps_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int A, int B,
                                                int r, int nc) {
    const int Cin = nc * r * r;
    const size_t total = (size_t)N * A * B * Cin;
    const size_t gs = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += gs) {
        if constexpr (!BWD) {
            // decode i in the fine (output) layout [N][A*r][B*r][nc]
            const int c = (int)(i % nc);
            size_t q = i / nc;
            const int wo = (int)(q % (B * r));
            q /= (B * r);
            const int ho = (int)(q % (A * r));
            const int n = (int)(q / (A * r));
            const int ii = ho / r, u = ho - ii * r;
            const int jj = wo / r, v = wo - jj * r;
            const size_t coarse = (((size_t)n * A + ii) * B + jj) * Cin + (size_t)c * r * r + v * r + u;
            dst[i] = src[coarse];
        } else {
            // decode i in the coarse (input-gradient) layout [N][A][B][nc*r*r]: coalesced stores, gathered loads
            const int cc = (int)(i % Cin);
            size_t q = i / Cin;
            const int jj = (int)(q % B);
            q /= B;
            const int ii = (int)(q % A);
            const int n = (int)(q / A);
            const int c = cc / (r * r);
            const int vu = cc - c * r * r;
            const int v = vu / r, u = vu - v * r;
            const size_t fine = (((size_t)n * A * r + ii * r + u) * (B * r) + jj * r + v) * nc + c;
            dst[i] = src[fine];
        }
    }
}

// LDS-transposing version: the permutation is contiguous in the COARSE layout per pixel (all nc*r*r channels) and in the FINE layout per
// (pixel row u: r*nc floats, and consecutive coarse pixels jj continue the same fine row), but a thread-per-element copy is 4-byte
// scattered on one side (173 / 262 us for the 168 MB of g10's output at B=16, 2-4x the HBM time).  A workgroup takes T coarse pixels of
// one row (T*Cin <= 4096 floats), moves them through LDS and is coalesced on both sides.  LDS slot of coarse element x: x + x/64 — the
// fine order walks channels first (stride r*r = 64 floats in x), 65 makes that walk conflict-free.
template <bool BWD>
__global__ void __launch_bounds__(NT) ps_tile_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int A, int B, int r,
                                                     int nc, int T) {
    __shared__ float tile[4096 + 64];
    const int t = threadIdx.x;
    const int rr = r * r, Cin = nc * rr, X = T * Cin, rn = r * nc, run = T * rn;
    const int bpr = B / T;
    int b = blockIdx.x;
    const int jb = b % bpr;
    b /= bpr;
    const int ii = b % A, n = b / A;
    const int jj0 = jb * T;
    const size_t base_c = (((size_t)n * A + ii) * B + jj0) * Cin;
    const size_t fine0 = (((size_t)n * A * r + (size_t)ii * r) * (size_t)(B * r) + (size_t)jj0 * r) * nc;
    const size_t row_stride = (size_t)(B * r) * nc;
    const float inv_run = 1.0f / (float)run, inv_rn = 1.0f / (float)rn, inv_nc = 1.0f / (float)nc;
    // exact for these small integers: (f + 0.5) / d is at least 0.5/d away from an integer
    auto fdiv = [](int f, float inv) { return (int)(((float)f + 0.5f) * inv); };
    auto decode = [&](int f, int& u, int& rem) {      // fine-order index -> coarse-order index x
        u = fdiv(f, inv_run);
        rem = f - u * run;
        const int tj = fdiv(rem, inv_rn);
        const int q = rem - tj * rn;
        const int v = fdiv(q, inv_nc);
        const int c = q - v * nc;
        return tj * Cin + c * rr + v * r + u;
    };
    if constexpr (!BWD) {
        for (int x = t; x < X; x += NT) tile[x + (x >> 6)] = src[base_c + x];
        __syncthreads();
        for (int f = t; f < X; f += NT) {
            int u, rem;
            const int x = decode(f, u, rem);
            dst[fine0 + (size_t)u * row_stride + rem] = tile[x + (x >> 6)];
        }
    } else {
        for (int f = t; f < X; f += NT) {
            int u, rem;
            const int x = decode(f, u, rem);
            tile[x + (x >> 6)] = src[fine0 + (size_t)u * row_stride + rem];
        }
        __syncthreads();
        for (int x = t; x < X; x += NT) dst[base_c + x] = tile[x + (x >> 6)];
    }
}

// coarse pixels per workgroup of ps_tile_kernel: the largest divisor of B with T*Cin <= 4096 (0: use the per-element kernel)
inline int ps_tile_T(int B, int Cin) {
    static const int off = getenv("PNP_PS_NOTILE") ? 1 : 0;
    if (off || Cin > 4096) return 0;
    int T = 4096 / Cin;
    if (T > B) T = B;
    while (T > 1 && (B % T) != 0) --T;
    return T;
}

// ---- critic input assembly (adversarial.py:325-335) ---------------------------------------------
struct CriticArgs {
    const float *a, *b, *c, *d, *logits;
    float* out;
    long long P;
    int Ca, tile_a, Cb, Cc, Cd, ncls, Ctot;
};
__device__ __forceinline__ float critic_channel(const CriticArgs& k, size_t p, int ch, int o1, int o2, int o3, int o4, int o5) {
    if (ch < o1) return k.a[p * k.Ca + (ch % k.Ca)];
    if (ch < o2) return k.b[p * k.Cb + (ch - o1)];
    if (ch < o3) return k.c[p * k.Cc + (ch - o2)];
    if (ch < o4) return k.d[p * k.Cd + (ch - o3)];
    if (ch < o5) return k.logits[p * k.ncls + (ch - o4)];
    const float* z = k.logits + p * k.ncls;
    int am = 0;
    float m = z[0];
    for (int j = 1; j < k.ncls; ++j)
        if (z[j] > m) { m = z[j]; am = j; }
    return (float)am;
}
// VEC: one 16-byte store per thread (Ctot % 4 == 0: the reference's 32 channels), i.e. 8 lanes cover one pixel's 128 contiguous bytes
template <bool VEC>
__global__ void __launch_bounds__(NT) critic_input_fwd_kernel(CriticArgs k) {
    const int CV = VEC ? (k.Ctot >> 2) : k.Ctot;
    const size_t total = (size_t)k.P * CV;
    const size_t gs = (size_t)gridDim.x * NT;
    const int o1 = k.Ca * k.tile_a, o2 = o1 + k.Cb, o3 = o2 + k.Cc, o4 = o3 + k.Cd, o5 = o4 + k.ncls;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += gs) {
        const size_t p = i / CV;
        const int cv = (int)(i - p * CV);
        if constexpr (VEC) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = critic_channel(k, p, 4 * cv + e, o1, o2, o3, o4, o5);
            st4(k.out + i * 4, v);
        } else {
            k.out[i] = critic_channel(k, p, cv, o1, o2, o3, o4, o5);
        }
    }
}
struct CriticBwdArgs {
    const float* dout;
    float *da, *db, *dc, *dd, *dlogits;
    long long P;
    int Ca, tile_a, Cb, Cc, Cd, ncls, Ctot;
};
// one thread per (pixel, source channel) over the concatenated *source* channel space Ca+Cb+Cc+Cd+ncls
__global__ void __launch_bounds__(NT) critic_input_bwd_kernel(CriticBwdArgs k) {
    const int Csrc = k.Ca + k.Cb + k.Cc + k.Cd + k.ncls;
    const size_t total = (size_t)k.P * Csrc;
    const size_t gs = (size_t)gridDim.x * NT;
    const int o1 = k.Ca * k.tile_a, o2 = o1 + k.Cb, o3 = o2 + k.Cc, o4 = o3 + k.Cd;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < total; i += gs) {
        const size_t p = i / Csrc;
        int ch = (int)(i - p * Csrc);
        const float* g = k.dout + p * k.Ctot;
        if (ch < k.Ca) {
            float s = 0.f;
            for (int t = 0; t < k.tile_a; ++t) s += g[t * k.Ca + ch];
            if (k.da) k.da[p * k.Ca + ch] = s;
            continue;
        }
        ch -= k.Ca;
        if (ch < k.Cb) { if (k.db) k.db[p * k.Cb + ch] = g[o1 + ch]; continue; }
        ch -= k.Cb;
        if (ch < k.Cc) { if (k.dc) k.dc[p * k.Cc + ch] = g[o2 + ch]; continue; }
        ch -= k.Cc;
        if (ch < k.Cd) { if (k.dd) k.dd[p * k.Cd + ch] = g[o3 + ch]; continue; }
        ch -= k.Cd;
        if (k.dlogits) k.dlogits[p * k.ncls + ch] = g[o4 + ch];   // argmax channel has zero gradient
    }
}

}  // namespace

extern "C" {

size_t pnp_bn_workspace_bytes(int64_t P, int32_t C) {
    int nblk, rpb;
    colreduce_plan(P, C, &nblk, &rpb);
    return (size_t)nblk * 2 * C * sizeof(float);
}

int pnp_bn_stats(const float* x, float* mean, float* var, int64_t P, int32_t C, void* workspace, size_t workspace_bytes,
                 void* stream) {
    PNP_REQUIRE(x && mean && var && P > 0 && C > 0, "pnp_bn_stats: bad argument");
    ColArgs a{};
    a.x = x; a.P = P; a.C = C;
    return run_colreduce<0>(a, mean, var, workspace, workspace_bytes, (hipStream_t)stream, "pnp_bn_stats");
}

int pnp_bn_stats_update(const float* x, float* mean, float* var, float* moving_mean, float* moving_var, int64_t P, int32_t C, float decay,
                        void* workspace, size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(x && mean && var && moving_mean && moving_var && P > 0 && C > 0, "pnp_bn_stats_update: bad argument");
    ColArgs a{};
    a.x = x; a.P = P; a.C = C;
    return run_colreduce<0>(a, mean, var, workspace, workspace_bytes, (hipStream_t)stream, "pnp_bn_stats_update", moving_mean, moving_var, decay);
}

int pnp_bn_stats_finish(float* parts, int32_t nparts, const float* shift, float* mean, float* var, float* moving_mean,
                        float* moving_var, int64_t P, int32_t C, float decay, void* stream) {
    PNP_REQUIRE(parts && shift && mean && var && nparts > 0 && P > 0 && C > 0, "pnp_bn_stats_finish: bad argument");
    PNP_REQUIRE((moving_mean != nullptr) == (moving_var != nullptr), "pnp_bn_stats_finish: moving_mean and moving_var go together");
    return launch_colreduce_final<0>(parts, shift, mean, var, nparts, C, (long long)P, moving_mean, moving_var, decay, (hipStream_t)stream,
                                     "pnp_bn_stats_finish");
}

int pnp_bn_update_moving(float* moving_mean, float* moving_var, const float* mean, const float* var, int64_t P, int32_t C,
                         float decay, void* stream) {
    PNP_REQUIRE(moving_mean && moving_var && mean && var && P > 0 && C > 0, "pnp_bn_update_moving: bad argument");
    hipLaunchKernelGGL(bn_update_moving_kernel, dim3(pnp_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, moving_mean,
                       moving_var, mean, var, (long long)P, C, decay);
    PNP_CHECK_LAUNCH("pnp_bn_update_moving");
    return PNP_OK;
}

int pnp_bn_apply(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                 const float* shortcut, int32_t Cs, float* y, int64_t P, int32_t C, float eps, float alpha, void* stream) {
    return pnp_bn_apply_h(x, mean, var, gamma, beta, shortcut, Cs, y, nullptr, P, C, eps, alpha, stream);
}

int pnp_bn_apply_h(const float* x, const float* mean, const float* var, const float* gamma, const float* beta,
                   const float* shortcut, int32_t Cs, float* y, void* yh, int64_t P, int32_t C, float eps, float alpha, void* stream) {
    PNP_REQUIRE(x && mean && var && gamma && beta && y && P > 0 && C > 0, "pnp_bn_apply: bad argument");
    if (shortcut) PNP_REQUIRE(Cs > 0 && Cs <= C && ((C - Cs) % 2) == 0, "pnp_bn_apply: bad shortcut channels %d vs %d", Cs, C);
    BnApplyArgs a{x, mean, var, gamma, beta, shortcut, y, (long long)P, C, shortcut ? Cs : C, eps, alpha, (__bf16*)yh};
    const bool vec = (C % 4 == 0) && (!shortcut || (Cs % 4 == 0 && ((C - Cs) / 2) % 4 == 0));
    const size_t nvec = (size_t)P * (vec ? C / 4 : C);
    if (vec) hipLaunchKernelGGL(bn_apply_kernel<true>, grid_rows(P, C / 4), dim3(NT), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(bn_apply_kernel<false>, dim3(grid_for(nvec)), dim3(NT), 0, (hipStream_t)stream, a);
    PNP_CHECK_LAUNCH("pnp_bn_apply");
    return PNP_OK;
}

int pnp_bn_bwd_reduce(const float* dout, const float* out, const float* x, const float* mean, const float* var, const float* gamma,
                      const float* beta, float* dgamma, float* dbeta, int64_t P, int32_t C, float eps, float alpha, void* workspace,
                      size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(dout && x && mean && var && dgamma && dbeta && P > 0 && C > 0, "pnp_bn_bwd_reduce: bad argument");
    PNP_REQUIRE(alpha < 0.f || out || (gamma && beta), "pnp_bn_bwd_reduce: a fused activation needs `out`, or gamma and beta to recompute its sign");
    PNP_REQUIRE((size_t)P * C < (1ull << 32), "pnp_bn_bwd_reduce: tensor exceeds 2^32 elements");
    ColArgs ca{};
    ca.x = x; ca.dout = dout; ca.out = out; ca.mean = mean; ca.var = var; ca.P = P; ca.C = C; ca.eps = eps; ca.alpha = alpha;
    ca.gamma = gamma; ca.beta = beta;
    return run_colreduce<1>(ca, dbeta, dgamma, workspace, workspace_bytes, (hipStream_t)stream, "pnp_bn_bwd_reduce");
}

int pnp_bn_bwd_apply(const float* dout, const float* out, const float* x, const float* mean, const float* var,
                     const float* gamma, const float* beta, const float* dgamma, const float* dbeta, float* dx, float* dshortcut, int32_t Cs,
                     int64_t P, int64_t P_norm, int32_t C, float eps, float alpha, int32_t training, float keep_prob,
                     uint64_t seed, uint32_t stream_id, void* stream) {
    return pnp_bn_bwd_apply_h(dout, out, x, mean, var, gamma, beta, dgamma, dbeta, dx, nullptr, dshortcut, Cs, P, P_norm, C, eps, alpha,
                              training, keep_prob, seed, stream_id, stream);
}

int pnp_bn_bwd_apply_h(const float* dout, const float* out, const float* x, const float* mean, const float* var,
                       const float* gamma, const float* beta, const float* dgamma, const float* dbeta, float* dx, void* dxh,
                       float* dshortcut, int32_t Cs, int64_t P, int64_t P_norm, int32_t C, float eps, float alpha, int32_t training,
                       float keep_prob, uint64_t seed, uint32_t stream_id, void* stream) {
    PNP_REQUIRE(dout && x && mean && var && gamma && (dx || dxh) && P > 0 && P_norm >= P && C > 0, "pnp_bn_bwd_apply: bad argument");
    PNP_REQUIRE(!training || (dgamma && dbeta), "pnp_bn_bwd_apply: training mode needs the dgamma / dbeta sums");
    PNP_REQUIRE(alpha < 0.f || out || (beta && !dshortcut), "pnp_bn_bwd_apply: a fused activation needs `out`, or beta (and no shortcut) to recompute its sign");
    PNP_REQUIRE((size_t)P * C < (1ull << 32), "pnp_bn_bwd_apply: tensor exceeds 2^32 elements");
    if (dshortcut) PNP_REQUIRE(Cs > 0 && Cs <= C && ((C - Cs) % 2) == 0, "pnp_bn_bwd_apply: bad shortcut channels");
    hipStream_t st = (hipStream_t)stream;
    BnBwdArgs a{};
    a.dout = dout; a.out = out; a.x = x; a.mean = mean; a.var = var; a.gamma = gamma; a.beta = beta; a.dgamma = dgamma; a.dbeta = dbeta;
    a.dx = dx; a.dshortcut = dshortcut; a.P = P; a.P_norm = P_norm; a.C = C; a.Cs = dshortcut ? Cs : C; a.eps = eps; a.alpha = alpha;
    a.training = training;
    a.do_drop = keep_prob < 1.f;
    a.drop_key = pnp_drop_key(seed, stream_id);
    a.drop_thresh = pnp_drop_thresh(keep_prob);
    a.sp = pnp_step_params_ptr(); a.drop_sid = stream_id;
    a.drop_keep = keep_prob < 1.f ? keep_prob : 1.f;
    a.dxh = (__bf16*)dxh;
    const bool vec = (C % 4 == 0) && (!dshortcut || (Cs % 4 == 0 && ((C - Cs) / 2) % 4 == 0));
    const size_t nvec = (size_t)P * (vec ? C / 4 : C);
    if (vec) hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, grid_rows(P, C / 4), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid_for(nvec)), dim3(NT), 0, st, a);
    PNP_CHECK_LAUNCH("pnp_bn_bwd_apply");
    return PNP_OK;
}

int pnp_bn_bwd(const float* dout, const float* out, const float* x, const float* mean, const float* var,
               const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, float* dshortcut, int32_t Cs, int64_t P,
               int32_t C, float eps, float alpha, int32_t training, float keep_prob, uint64_t seed, uint32_t stream_id,
               void* workspace, size_t workspace_bytes, void* stream) {
    return pnp_bn_bwd_acc(dout, out, x, mean, var, gamma, beta, dx, dgamma, dbeta, nullptr, nullptr, dshortcut, Cs, P, C, eps, alpha, training,
                          keep_prob, seed, stream_id, workspace, workspace_bytes, stream);
}

int pnp_bn_bwd_acc(const float* dout, const float* out, const float* x, const float* mean, const float* var,
                   const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, float* dgamma_acc, float* dbeta_acc, float* dshortcut,
                   int32_t Cs, int64_t P, int32_t C, float eps, float alpha, int32_t training, float keep_prob, uint64_t seed,
                   uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream) {
    return pnp_bn_bwd_acc_h(dout, out, x, mean, var, gamma, beta, dx, nullptr, dgamma, dbeta, dgamma_acc, dbeta_acc, dshortcut, Cs, P, C, eps,
                            alpha, training, keep_prob, seed, stream_id, workspace, workspace_bytes, stream);
}

int pnp_bn_bwd_acc_h(const float* dout, const float* out, const float* x, const float* mean, const float* var,
                     const float* gamma, const float* beta, float* dx, void* dxh, float* dgamma, float* dbeta, float* dgamma_acc,
                     float* dbeta_acc, float* dshortcut, int32_t Cs, int64_t P, int32_t C, float eps, float alpha, int32_t training,
                     float keep_prob, uint64_t seed, uint32_t stream_id, void* workspace, size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(dout && x && mean && var && dgamma && dbeta && P > 0 && C > 0, "pnp_bn_bwd: bad argument");
    PNP_REQUIRE(alpha < 0.f || out || (gamma && beta && !dshortcut), "pnp_bn_bwd: a fused activation needs `out`, or beta (and no shortcut) to recompute its sign");
    PNP_REQUIRE((size_t)P * C < (1ull << 32), "pnp_bn_bwd: tensor exceeds 2^32 elements");
    PNP_REQUIRE((dgamma_acc != nullptr) == (dbeta_acc != nullptr), "pnp_bn_bwd_acc: dgamma_acc and dbeta_acc go together");
    ColArgs ca{};
    ca.x = x; ca.dout = dout; ca.out = out; ca.mean = mean; ca.var = var; ca.P = P; ca.C = C; ca.eps = eps; ca.alpha = alpha;
    ca.gamma = gamma; ca.beta = beta;
    if (int e = run_colreduce<1>(ca, dbeta, dgamma, workspace, workspace_bytes, (hipStream_t)stream, "pnp_bn_bwd", nullptr, nullptr, 0.f,
                                 dbeta_acc, dgamma_acc))
        return e;
    return pnp_bn_bwd_apply_h(dout, out, x, mean, var, gamma, beta, dgamma, dbeta, dx, dxh, dshortcut, Cs, P, P, C, eps, alpha, training,
                              keep_prob, seed, stream_id, stream);
}

int pnp_dropout(const float* x, float* y, size_t n, float keep_prob, uint64_t seed, uint32_t stream_id, void* stream) {
    PNP_REQUIRE(y, "pnp_dropout: bad argument");
    return pnp_dropout_h(x, y, nullptr, n, keep_prob, seed, stream_id, stream);
}

int pnp_dropout_h(const float* x, float* y, void* yh, size_t n, float keep_prob, uint64_t seed, uint32_t stream_id, void* stream) {
    PNP_REQUIRE(x && (y || yh) && keep_prob > 0.f, "pnp_dropout: bad argument");
    PNP_REQUIRE(n < (1ull << 32), "pnp_dropout: tensor exceeds 2^32 elements");
    if (n == 0) return PNP_OK;
    const float keepv = keep_prob < 1.f ? keep_prob : 1.f;      // tf.nn.dropout divides: div(x, keep_prob) * mask
    const uint32_t thresh = keep_prob < 1.f ? pnp_drop_thresh(keep_prob) : 0u;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, x, y, n,
                       pnp_drop_key(seed, stream_id), thresh, keepv, (__bf16*)yh, pnp_step_params_ptr(), stream_id);
    PNP_CHECK_LAUNCH("pnp_dropout");
    return PNP_OK;
}

int pnp_add(const float* x, const float* y, float* out, size_t n, void* stream) {
    PNP_REQUIRE(x && y && out, "pnp_add: null pointer");
    if (n == 0) return PNP_OK;
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, x, y, out, n);
    PNP_CHECK_LAUNCH("pnp_add");
    return PNP_OK;
}

int pnp_axpby(const float* x, float* y, size_t n, float a, float b, void* stream) {
    PNP_REQUIRE(x && y, "pnp_axpby: null pointer");
    if (n == 0) return PNP_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid_for(n / 4 + 1)), dim3(NT), 0, (hipStream_t)stream, x, y, n, a, b);
    PNP_CHECK_LAUNCH("pnp_axpby");
    return PNP_OK;
}

int pnp_maxpool2_fwd(const float* x, float* y, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    PNP_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "pnp_maxpool2_fwd: bad argument (H,W must be even)");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * ((C % 4 == 0) ? C / 4 : C);
    hipLaunchKernelGGL(maxpool2_kernel<false>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, y, N, H, W, C);
    PNP_CHECK_LAUNCH("pnp_maxpool2_fwd");
    return PNP_OK;
}

int pnp_maxpool2_bwd(const float* x, const float* dy, float* dx, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    PNP_REQUIRE(x && dy && dx && N > 0 && C > 0 && H > 0 && W > 0 && (H % 2) == 0 && (W % 2) == 0, "pnp_maxpool2_bwd: bad argument");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * ((C % 4 == 0) ? C / 4 : C);
    hipLaunchKernelGGL(maxpool2_kernel<true>, dim3(grid_for(total)), dim3(NT), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C);
    PNP_CHECK_LAUNCH("pnp_maxpool2_bwd");
    return PNP_OK;
}

int pnp_ps_fwd(const float* x, float* y, int32_t N, int32_t A, int32_t B, int32_t r, int32_t nc, void* stream) {
    PNP_REQUIRE(x && y && N > 0 && A > 0 && B > 0 && r > 0 && nc > 0, "pnp_ps_fwd: bad argument");
    const size_t total = (size_t)N * A * B * nc * r * r;
    if (const int T = ps_tile_T(B, nc * r * r)) {
        hipLaunchKernelGGL(ps_tile_kernel<false>, dim3((unsigned)(N * A * (B / T))), dim3(NT), 0, (hipStream_t)stream, x, y, N, A, B, r, nc, T);
        PNP_CHECK_LAUNCH("pnp_ps_fwd");
        return PNP_OK;
    }
    hipLaunchKernelGGL(ps_kernel<false>, dim3(grid_for(total, 256 * 16)), dim3(NT), 0, (hipStream_t)stream, x, y, N, A, B, r, nc);
    PNP_CHECK_LAUNCH("pnp_ps_fwd");
    return PNP_OK;
}

int pnp_ps_bwd(const float* dy, float* dx, int32_t N, int32_t A, int32_t B, int32_t r, int32_t nc, void* stream) {
    PNP_REQUIRE(dy && dx && N > 0 && A > 0 && B > 0 && r > 0 && nc > 0, "pnp_ps_bwd: bad argument");
    const size_t total = (size_t)N * A * B * nc * r * r;
    if (const int T = ps_tile_T(B, nc * r * r)) {
        hipLaunchKernelGGL(ps_tile_kernel<true>, dim3((unsigned)(N * A * (B / T))), dim3(NT), 0, (hipStream_t)stream, dy, dx, N, A, B, r, nc, T);
        PNP_CHECK_LAUNCH("pnp_ps_bwd");
        return PNP_OK;
    }
    hipLaunchKernelGGL(ps_kernel<true>, dim3(grid_for(total, 256 * 16)), dim3(NT), 0, (hipStream_t)stream, dy, dx, N, A, B, r, nc);
    PNP_CHECK_LAUNCH("pnp_ps_bwd");
    return PNP_OK;
}

int pnp_critic_input_fwd(const float* a, int32_t Ca, int32_t tile_a, const float* b, int32_t Cb, const float* c, int32_t Cc,
                         const float* d, int32_t Cd, const float* logits, int32_t ncls, float* out, int64_t P, void* stream) {
    PNP_REQUIRE(a && b && c && d && logits && out && P > 0 && Ca > 0 && tile_a > 0 && Cb > 0 && Cc > 0 && Cd > 0 && ncls > 0,
                "pnp_critic_input_fwd: bad argument");
    CriticArgs k{a, b, c, d, logits, out, (long long)P, Ca, tile_a, Cb, Cc, Cd, ncls, Ca * tile_a + Cb + Cc + Cd + ncls + 1};
    if ((k.Ctot & 3) == 0)
        hipLaunchKernelGGL(critic_input_fwd_kernel<true>, dim3(grid_for((size_t)P * (k.Ctot / 4), 256 * 16)), dim3(NT), 0, (hipStream_t)stream, k);
    else
        hipLaunchKernelGGL(critic_input_fwd_kernel<false>, dim3(grid_for((size_t)P * k.Ctot, 256 * 16)), dim3(NT), 0, (hipStream_t)stream, k);
    PNP_CHECK_LAUNCH("pnp_critic_input_fwd");
    return PNP_OK;
}

int pnp_critic_input_bwd(const float* dout, float* da, int32_t Ca, int32_t tile_a, float* db, int32_t Cb, float* dc, int32_t Cc,
                         float* dd, int32_t Cd, float* dlogits, int32_t ncls, int64_t P, void* stream) {
    PNP_REQUIRE(dout && P > 0 && Ca > 0 && tile_a > 0 && Cb > 0 && Cc > 0 && Cd > 0 && ncls > 0, "pnp_critic_input_bwd: bad argument");
    CriticBwdArgs k{dout, da, db, dc, dd, dlogits, (long long)P, Ca, tile_a, Cb, Cc, Cd, ncls, Ca * tile_a + Cb + Cc + Cd + ncls + 1};
    hipLaunchKernelGGL(critic_input_bwd_kernel, dim3(grid_for((size_t)P * (Ca + Cb + Cc + Cd + ncls), 256 * 16)), dim3(NT), 0,
                       (hipStream_t)stream, k);
    PNP_CHECK_LAUNCH("pnp_critic_input_bwd");
    return PNP_OK;
}

}  // extern "C"
