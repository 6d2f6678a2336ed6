// loss_optim.hip — segmentation loss (weighted cross-entropy + soft Dice), softmax/argmax, hard Dice,
// and the flat-arena optimisers (Adam / RMSProp / Momentum / weight clip / L2).
// Reference: source_segmenter.py:211-273 (_get_cost, _softmax_weighted_loss, _dice_loss_fun),
//            layers.py:134-138 (pixel_wise_softmax_2), lib.py:96-110 (_dice_eval),
//            source_segmenter.py:357-381 (Adam / Momentum), adversarial.py:633-656 (RMSProp, clip).
#include "pnp_common.h"

namespace {

constexpr int NT = 256;
constexpr int MAXC = 8;

// reduce NV per-thread floats over the block; result valid in thread 0's `v`
template <int NV>
__device__ __forceinline__ void block_reduce(float (&v)[NV], float* lds /* >= 4*NV floats */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        float x = v[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        v[i] = x;
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = lds[i] + lds[NV + i] + lds[2 * NV + i] + lds[3 * NV + i];
    }
}

__device__ __forceinline__ void softmax_stable(const float* z, int ncls, float* p) {
    float m = z[0];
    for (int j = 1; j < ncls; ++j) m = fmaxf(m, z[j]);
    float s = 0.f;
    for (int j = 0; j < ncls; ++j) {
        p[j] = expf(z[j] - m);
        s += p[j];
    }
    const float inv = 1.0f / s;
    for (int j = 0; j < ncls; ++j) p[j] *= inv;
}

// partial sums per class: [0]=n_i (sum y), [1]=I_i (sum p*y), [2]=S_i (sum p*p), [3]=X_i (sum -y*log(clip(p)))
__global__ void __launch_bounds__(NT) seg_loss_partial_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                              float* __restrict__ part, long long P, int ncls) {
    __shared__ float lds[4 * 4 * MAXC];
    float acc[4 * MAXC];
#pragma unroll
    for (int i = 0; i < 4 * MAXC; ++i) acc[i] = 0.f;
    const long long gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += gs) {
        float z[MAXC], p[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) z[j] = j < ncls ? logits[i * ncls + j] : 0.f;
        softmax_stable(z, ncls, p);
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            if (j < ncls) {
                const float yy = y[i * ncls + j];
                const float pc = fminf(fmaxf(p[j], 0.005f), 1.0f);
                acc[j] += yy;
                acc[MAXC + j] = fmaf(p[j], yy, acc[MAXC + j]);
                acc[2 * MAXC + j] = fmaf(p[j], p[j], acc[2 * MAXC + j]);
                acc[3 * MAXC + j] -= yy * logf(pc);
            }
        }
    }
    block_reduce<4 * MAXC>(acc, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 4 * MAXC; ++i) part[(size_t)blockIdx.x * 4 * MAXC + i] = acc[i];
    }
}

// ws layout: double sums[32] | float partials[nblk][32]
__global__ void seg_loss_final_kernel(const float* __restrict__ part, double* __restrict__ sums, float* __restrict__ out,
                                      int nblk, long long P, int ncls, float miu_cross, float miu_dice) {
    // launched with 1024 threads = 32 sums x 32 slices of the block list (fixed order => deterministic)
    __shared__ double s[4 * MAXC];
    __shared__ double red[32][33];
    const int t = threadIdx.x;
    const int q = t & 31, sl = t >> 5;
    {
        double a = 0.0;
        for (int b = sl; b < nblk; b += 32) a += (double)part[(size_t)b * 4 * MAXC + q];
        red[sl][q] = a;
    }
    __syncthreads();
    if (t < 4 * MAXC) {
        double a = 0.0;
        for (int j = 0; j < 32; ++j) a += red[j][t];
        s[t] = a;
        sums[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        double ntot = 0.0;
        for (int j = 0; j < ncls; ++j) ntot += s[j];
        double xent = 0.0, dice = 0.0;
        for (int j = 0; j < ncls; ++j) {
            const double w = 1.0 - s[j] / ntot;
            xent += w * s[3 * MAXC + j];
            dice += 2.0 * s[MAXC + j] / (s[2 * MAXC + j] + s[j] + 1e-7);
        }
        xent /= (double)P;
        dice = -dice / (double)ncls;
        out[0] = (float)((double)miu_cross * xent + (double)miu_dice * dice);
        out[1] = (float)xent;
        out[2] = (float)dice;
    }
}

__global__ void __launch_bounds__(NT) seg_loss_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ y,
                                                          float* __restrict__ dlogits, const double* __restrict__ sums,
                                                          long long P, long long P_norm, int ncls, float miu_cross, float miu_dice,
                                                          float gscale) {
    float w[MAXC], invD[MAXC], I2[MAXC];
    {
        double ntot = 0.0;
        for (int j = 0; j < ncls; ++j) ntot += sums[j];
        for (int j = 0; j < MAXC; ++j) {
            if (j < ncls) {
                const double D = sums[2 * MAXC + j] + sums[j] + 1e-7;
                w[j] = (float)((1.0 - sums[j] / ntot) / (double)P_norm);
                invD[j] = (float)(1.0 / D);
                I2[j] = (float)(2.0 * sums[MAXC + j] / (D * D));
            } else {
                w[j] = invD[j] = I2[j] = 0.f;
            }
        }
    }
    const float cd = -2.0f * miu_dice / (float)ncls;
    const long long gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += gs) {
        float z[MAXC], p[MAXC], g[MAXC];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) z[j] = j < ncls ? logits[i * ncls + j] : 0.f;
        softmax_stable(z, ncls, p);
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            g[j] = 0.f;
            if (j < ncls) {
                const float yy = y[i * ncls + j];
                float gx = (p[j] >= 0.005f) ? (-miu_cross * w[j] * yy / p[j]) : 0.f;
                float gd = cd * (yy * invD[j] - I2[j] * p[j]);
                g[j] = gx + gd;
                dot = fmaf(g[j], p[j], dot);
            }
        }
#pragma unroll
        for (int j = 0; j < MAXC; ++j)
            if (j < ncls) dlogits[i * ncls + j] = gscale * p[j] * (g[j] - dot);
    }
}

__global__ void __launch_bounds__(NT) softmax_argmax_kernel(const float* __restrict__ logits, float* __restrict__ prob,
                                                            int64_t* __restrict__ label, long long P, int ncls) {
    const long long gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += gs) {
        float e[MAXC];
        float s = 0.f;
        for (int j = 0; j < ncls; ++j) {
            e[j] = expf(logits[i * ncls + j]);
            s += e[j];
        }
        int am = 0;
        float best = 0.f;
        for (int j = 0; j < ncls; ++j) {
            float p = e[j] / s;
            p = fminf(fmaxf(p, -1e15f), 1e15f);
            if (prob) prob[i * ncls + j] = p;
            if (j == 0 || p > best) { best = p; am = j; }
        }
        if (label) label[i] = am;
    }
}

// partial: [0..7] count(label==i), [8..15] sum y_i, [16..23] sum [label==i]*y_i
__global__ void __launch_bounds__(NT) dice_eval_partial_kernel(const int64_t* __restrict__ label, const float* __restrict__ y,
                                                               float* __restrict__ part, long long P, int ncls) {
    __shared__ float lds[4 * 3 * MAXC];
    float acc[3 * MAXC];
#pragma unroll
    for (int i = 0; i < 3 * MAXC; ++i) acc[i] = 0.f;
    const long long gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += gs) {
        const int l = (int)label[i];
#pragma unroll
        for (int j = 0; j < MAXC; ++j) {
            if (j < ncls) {
                const float yy = y[i * ncls + j];
                const float pr = (l == j) ? 1.f : 0.f;
                acc[j] += pr;
                acc[MAXC + j] += yy;
                acc[2 * MAXC + j] += pr * yy;
            }
        }
    }
    block_reduce<3 * MAXC>(acc, lds);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 3 * MAXC; ++i) part[(size_t)blockIdx.x * 3 * MAXC + i] = acc[i];
    }
}
__global__ void dice_eval_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nblk, int ncls) {
    __shared__ double s[3 * MAXC];
    const int t = threadIdx.x;
    if (t < 3 * MAXC) {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += (double)part[(size_t)b * 3 * MAXC + t];
        s[t] = a;
    }
    __syncthreads();
    if (t == 0) {
        double tot = 0.0;
        for (int j = 0; j < ncls; ++j) {
            const double d = 2.0 * s[2 * MAXC + j] / (s[j] + s[MAXC + j] + 1e-7);
            out[1 + j] = (float)d;
            tot += d;
        }
        out[0] = (float)(tot / ncls);
    }
}

inline int loss_blocks(long long P) {
    long long b = (P + NT * 4 - 1) / (NT * 4);
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- optimisers ---------------------------------------------------------------------------------
struct OptArgs {
    float* w;
    const float* g;
    float* s0;
    float* s1;
    size_t n;
    const float* chunk_l2;
    const uint8_t* chunk_mask;
    float lr, b1, b2, eps, aux;
    const pnp_step_params* sp;      // step capture (Adam): the bias-corrected learning rate from device memory; null: `lr`
};

// KIND 0 adam, 1 rmsprop, 2 momentum, 3 clip
template <int KIND>
__global__ void __launch_bounds__(NT) opt_kernel(OptArgs a) {
    // one block per PNP_OPT_CHUNK (1024) elements: 256 threads x float4
    const size_t chunk = blockIdx.x;
    if (a.chunk_mask && a.chunk_mask[chunk] == 0) return;
    const float l2 = (KIND != 3 && a.chunk_l2) ? a.chunk_l2[chunk] : 0.f;
    const size_t base = chunk * PNP_OPT_CHUNK + (size_t)threadIdx.x * 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const size_t i = base + e;
        if (i >= a.n) break;
        float w = a.w[i];
        if constexpr (KIND == 3) {
            a.w[i] = fminf(fmaxf(w, a.lr), a.b1);   // lr=lo, b1=hi
        } else {
            float g = a.g[i] + l2 * w;
            if constexpr (KIND == 0) {
                float m = a.s0[i], v = a.s1[i];
                m = m + (g - m) * (1.0f - a.b1);
                v = v + (g * g - v) * (1.0f - a.b2);
                a.s0[i] = m;
                a.s1[i] = v;
                a.w[i] = w - (a.sp ? a.sp->adam_lr_t : a.lr) * m / (sqrtf(v) + a.eps);   // lr already bias-corrected (lr_t)
            } else if constexpr (KIND == 1) {
                float ms = a.s0[i];
                ms = ms + (g * g - ms) * (1.0f - a.b1);   // b1 = decay
                a.s0[i] = ms;
                a.w[i] = w - a.lr * g / sqrtf(ms + a.eps);
            } else {
                float acc = a.s0[i] * a.b1 + g;   // b1 = momentum
                a.s0[i] = acc;
                a.w[i] = w - a.lr * acc;
            }
        }
    }
}

__global__ void __launch_bounds__(NT) l2_partial_kernel(const float* __restrict__ w, size_t n, const float* __restrict__ chunk_l2,
                                                        float* __restrict__ part) {
    __shared__ float lds[4];
    float acc[1] = {0.f};
    const size_t nchunks = (n + PNP_OPT_CHUNK - 1) / PNP_OPT_CHUNK;
    for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const float l2 = chunk_l2 ? chunk_l2[chunk] : 1.f;
        if (l2 == 0.f) continue;
        const size_t base = chunk * PNP_OPT_CHUNK + (size_t)threadIdx.x * 4;
        float s = 0.f;
        for (int e = 0; e < 4; ++e) {
            const size_t i = base + e;
            if (i < n) s = fmaf(w[i], w[i], s);
        }
        acc[0] += 0.5f * l2 * s;
    }
    block_reduce<1>(acc, lds);
    if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}
__global__ void sum_final_kernel(const float* __restrict__ part, int n, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < n; ++i) s += (double)part[i];
        out[0] = (float)s;
    }
}

__global__ void wgan_loss_kernel(const float* a, const float* b, const float* c, const float* d, int B, float ca, float cb, float cc,
                                 float cd, float* out) {
    // B is the batch size (<= a few hundred): one wave, fixed order
    const float* p[4] = {a, b, c, d};
    const float co[4] = {ca, cb, cc, cd};
    double tot = 0.0;
    for (int q = 0; q < 4; ++q) {
        if (!p[q]) continue;
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 64) s += p[q][i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
        tot += (double)co[q] * ((double)s / (double)B);
    }
    if (threadIdx.x == 0) out[0] = (float)tot;
}

__global__ void __launch_bounds__(NT) fill_kernel(float* p, size_t n, float v) {
    const size_t gs = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += gs) p[i] = v;
}

template <int KIND>
int run_opt(OptArgs a, hipStream_t st, const char* who) {
    if (a.n == 0) return PNP_OK;
    const size_t nchunks = (a.n + PNP_OPT_CHUNK - 1) / PNP_OPT_CHUNK;
    hipLaunchKernelGGL(opt_kernel<KIND>, dim3((unsigned)nchunks), dim3(NT), 0, st, a);
    PNP_CHECK_LAUNCH(who);
    return PNP_OK;
}


// ---- label helpers of the step / monitoring path (lib.py:75-92, source_segmenter.py:85,479-481) ---------------------------------------
// lib._label_decomp: integer-valued float label map -> one-hot float rows (labels >= ncls: all-zero row)
__global__ void __launch_bounds__(NT) label_decomp_kernel(const float* __restrict__ label, float* __restrict__ onehot, long long P, int ncls) {
    const long long n = P * ncls, gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < n; i += gs) {
        const long long p = i / ncls;
        onehot[i] = (label[p] == (float)(i - p * ncls)) ? 1.f : 0.f;
    }
}

// tf.argmax(y, 3) of the one-hot labels (compact_y) and tf.confusion_matrix(compact_y, compact_pred) in one pass: per-block counts in
// LDS, then integer atomics into cm[truth][pred] (integer sums: order-independent, deterministic).  compact_y may be null.
__global__ void __launch_bounds__(NT) confusion_kernel(const float* __restrict__ y, const int64_t* __restrict__ pred,
                                                       int64_t* __restrict__ compact_y, unsigned long long* __restrict__ cm, long long P, int ncls) {
    __shared__ unsigned int cnt[MAXC * MAXC];
    for (int i = threadIdx.x; i < MAXC * MAXC; i += NT) cnt[i] = 0;
    __syncthreads();
    const long long gs = (long long)gridDim.x * NT;
    for (long long i = (long long)blockIdx.x * NT + threadIdx.x; i < P; i += gs) {
        int am = 0;
        float best = y[i * ncls];
        for (int j = 1; j < ncls; ++j) {
            const float v = y[i * ncls + j];
            if (v > best) { best = v; am = j; }          // lowest index on ties, like tf.argmax's kernels
        }
        if (compact_y) compact_y[i] = am;
        if (pred) {
            const int pr = (int)pred[i];
            if ((unsigned)pr < (unsigned)ncls) atomicAdd(&cnt[am * ncls + pr], 1u);
        }
    }
    __syncthreads();
    if (pred)
        for (int i = threadIdx.x; i < ncls * ncls; i += NT)
            if (cnt[i]) atomicAdd(&cm[i], (unsigned long long)cnt[i]);
}

// synchronised batch statistics (opt-in): per-replica (mean, biased variance) -> raw moments in double, and back after the
// sum over `world` replicas of equally many rows:  E[x] = sum(mean_r)/W,  E[x^2] = sum(var_r + mean_r^2)/W
__global__ void bn_moments_kernel(const float* __restrict__ mean, const float* __restrict__ var, double* __restrict__ mom, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = (double)mean[c];
    mom[c] = m;
    mom[C + c] = (double)var[c] + m * m;
}
__global__ void bn_from_moments_kernel(const double* __restrict__ mom, int world, float* __restrict__ mean, float* __restrict__ var, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = mom[c] / (double)world;
    double v = mom[C + c] / (double)world - m * m;
    if (v < 0.0) v = 0.0;
    mean[c] = (float)m;
    var[c] = (float)v;
}

}  // namespace

extern "C" {

int pnp_label_decomp(const float* label, float* onehot, int64_t P, int32_t ncls, void* stream) {
    PNP_REQUIRE(label && onehot && P > 0 && ncls > 0, "pnp_label_decomp: bad argument");
    long long nb = (P * ncls + NT - 1) / NT;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(label_decomp_kernel, dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, label, onehot, (long long)P, ncls);
    PNP_CHECK_LAUNCH("pnp_label_decomp");
    return PNP_OK;
}

int pnp_confusion_matrix(const float* y, const int64_t* pred, int64_t* compact_y, int64_t* cm, int64_t P, int32_t ncls, void* stream) {
    PNP_REQUIRE(y && P > 0 && ncls > 0 && ncls <= MAXC, "pnp_confusion_matrix: bad argument (1..%d classes)", MAXC);
    PNP_REQUIRE((pred != nullptr) == (cm != nullptr), "pnp_confusion_matrix: pred and cm go together");
    PNP_REQUIRE(pred || compact_y, "pnp_confusion_matrix: nothing to compute");
    hipStream_t st = (hipStream_t)stream;
    if (cm && hipMemsetAsync(cm, 0, sizeof(int64_t) * ncls * ncls, st) != hipSuccess) {
        pnp_set_error("pnp_confusion_matrix: memset failed");
        return PNP_ELAUNCH;
    }
    long long nb = (P + NT - 1) / NT;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)nb), dim3(NT), 0, st, y, pred, compact_y, (unsigned long long*)cm, (long long)P, ncls);
    PNP_CHECK_LAUNCH("pnp_confusion_matrix");
    return PNP_OK;
}

int pnp_bn_moments(const float* mean, const float* var, double* moments, int32_t C, void* stream) {
    PNP_REQUIRE(mean && var && moments && C > 0, "pnp_bn_moments: bad argument");
    hipLaunchKernelGGL(bn_moments_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, mean, var, moments, C);
    PNP_CHECK_LAUNCH("pnp_bn_moments");
    return PNP_OK;
}

int pnp_bn_from_moments(const double* moments, int32_t world, float* mean, float* var, int32_t C, void* stream) {
    PNP_REQUIRE(mean && var && moments && C > 0 && world > 0, "pnp_bn_from_moments: bad argument");
    hipLaunchKernelGGL(bn_from_moments_kernel, dim3((unsigned)((C + 63) / 64)), dim3(64), 0, (hipStream_t)stream, moments, world, mean, var, C);
    PNP_CHECK_LAUNCH("pnp_bn_from_moments");
    return PNP_OK;
}

size_t pnp_seg_loss_workspace_bytes(int64_t P, int32_t ncls) {
    (void)ncls;
    return 256 + (size_t)loss_blocks(P) * 4 * MAXC * sizeof(float);
}

int pnp_seg_loss_fwd(const float* logits, const float* y, float* out, int64_t P, int32_t ncls, float miu_cross, float miu_dice,
                     void* workspace, size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(logits && y && out && workspace && P > 0 && ncls > 0 && ncls <= MAXC, "pnp_seg_loss_fwd: bad argument");
    PNP_REQUIRE(workspace_bytes >= pnp_seg_loss_workspace_bytes(P, ncls), "pnp_seg_loss_fwd: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = loss_blocks(P);
    double* sums = (double*)workspace;
    float* part = (float*)((char*)workspace + 256);
    hipLaunchKernelGGL(seg_loss_partial_kernel, dim3(nblk), dim3(NT), 0, st, logits, y, part, (long long)P, ncls);
    PNP_CHECK_LAUNCH("seg_loss_partial_kernel");
    hipLaunchKernelGGL(seg_loss_final_kernel, dim3(1), dim3(1024), 0, st, (const float*)part, sums, out, nblk, (long long)P, ncls,
                       miu_cross, miu_dice);
    PNP_CHECK_LAUNCH("seg_loss_final_kernel");
    return PNP_OK;
}

int pnp_seg_loss_bwd_norm(const float* logits, const float* y, float* dlogits, int64_t P, int64_t P_norm, int32_t ncls,
                          float miu_cross, float miu_dice, float gscale, const void* workspace, size_t workspace_bytes,
                          void* stream) {
    PNP_REQUIRE(logits && y && dlogits && workspace && P > 0 && P_norm >= P && ncls > 0 && ncls <= MAXC,
                "pnp_seg_loss_bwd: bad argument");
    PNP_REQUIRE(workspace_bytes >= 256, "pnp_seg_loss_bwd: workspace too small");
    hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(loss_blocks(P)), dim3(NT), 0, (hipStream_t)stream, logits, y, dlogits,
                       (const double*)workspace, (long long)P, (long long)P_norm, ncls, miu_cross, miu_dice, gscale);
    PNP_CHECK_LAUNCH("seg_loss_bwd_kernel");
    return PNP_OK;
}

int pnp_seg_loss_bwd(const float* logits, const float* y, float* dlogits, int64_t P, int32_t ncls, float miu_cross,
                     float miu_dice, float gscale, const void* workspace, size_t workspace_bytes, void* stream) {
    return pnp_seg_loss_bwd_norm(logits, y, dlogits, P, P, ncls, miu_cross, miu_dice, gscale, workspace, workspace_bytes, stream);
}

int pnp_softmax_argmax(const float* logits, float* prob, int64_t* label, int64_t P, int32_t ncls, void* stream) {
    PNP_REQUIRE(logits && (prob || label) && P > 0 && ncls > 0 && ncls <= MAXC, "pnp_softmax_argmax: bad argument");
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3(loss_blocks(P)), dim3(NT), 0, (hipStream_t)stream, logits, prob, label,
                       (long long)P, ncls);
    PNP_CHECK_LAUNCH("softmax_argmax_kernel");
    return PNP_OK;
}

int pnp_dice_eval(const int64_t* label, const float* y, float* out, int64_t P, int32_t ncls, void* workspace,
                  size_t workspace_bytes, void* stream) {
    PNP_REQUIRE(label && y && out && workspace && P > 0 && ncls > 0 && ncls <= MAXC, "pnp_dice_eval: bad argument");
    const int nblk = loss_blocks(P);
    PNP_REQUIRE(workspace_bytes >= (size_t)nblk * 3 * MAXC * sizeof(float), "pnp_dice_eval: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dice_eval_partial_kernel, dim3(nblk), dim3(NT), 0, st, label, y, (float*)workspace, (long long)P, ncls);
    PNP_CHECK_LAUNCH("dice_eval_partial_kernel");
    hipLaunchKernelGGL(dice_eval_final_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, out, nblk, ncls);
    PNP_CHECK_LAUNCH("dice_eval_final_kernel");
    return PNP_OK;
}

int pnp_adam_step(float* w, const float* g, float* m, float* v, size_t n, const float* chunk_l2, const uint8_t* chunk_mask,
                  float lr, float beta1, float beta2, float eps, int32_t t, void* stream) {
    PNP_REQUIRE(w && g && m && v && t >= 1, "pnp_adam_step: bad argument");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) / (1.0 - pow((double)beta1, (double)t));
    OptArgs a{w, g, m, v, n, chunk_l2, chunk_mask, (float)lr_t, beta1, beta2, eps, 0.f, pnp_step_params_ptr()};
    return run_opt<0>(a, (hipStream_t)stream, "pnp_adam_step");
}

int pnp_rmsprop_step(float* w, const float* g, float* ms, size_t n, const float* chunk_l2, const uint8_t* chunk_mask, float lr,
                     float decay, float eps, void* stream) {
    PNP_REQUIRE(w && g && ms, "pnp_rmsprop_step: bad argument");
    OptArgs a{w, g, ms, nullptr, n, chunk_l2, chunk_mask, lr, decay, 0.f, eps, 0.f, nullptr};
    return run_opt<1>(a, (hipStream_t)stream, "pnp_rmsprop_step");
}

int pnp_momentum_step(float* w, const float* g, float* acc, size_t n, const float* chunk_l2, const uint8_t* chunk_mask,
                      float lr, float momentum, void* stream) {
    PNP_REQUIRE(w && g && acc, "pnp_momentum_step: bad argument");
    OptArgs a{w, g, acc, nullptr, n, chunk_l2, chunk_mask, lr, momentum, 0.f, 0.f, 0.f, nullptr};
    return run_opt<2>(a, (hipStream_t)stream, "pnp_momentum_step");
}

int pnp_clip(float* w, size_t n, const uint8_t* chunk_mask, float lo, float hi, void* stream) {
    PNP_REQUIRE(w && lo <= hi, "pnp_clip: bad argument");
    OptArgs a{w, nullptr, nullptr, nullptr, n, nullptr, chunk_mask, lo, hi, 0.f, 0.f, 0.f, nullptr};
    return run_opt<3>(a, (hipStream_t)stream, "pnp_clip");
}

int pnp_wgan_loss(const float* ct_cls, const float* mr_cls, const float* ct_mask, const float* mr_mask, int32_t B, float c_ct_cls,
                  float c_mr_cls, float c_ct_mask, float c_mr_mask, float* out, void* stream) {
    PNP_REQUIRE(out && B > 0 && (ct_cls || mr_cls || ct_mask || mr_mask), "pnp_wgan_loss: bad argument");
    hipLaunchKernelGGL(wgan_loss_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ct_cls, mr_cls, ct_mask, mr_mask, B, c_ct_cls,
                       c_mr_cls, c_ct_mask, c_mr_mask, out);
    PNP_CHECK_LAUNCH("wgan_loss_kernel");
    return PNP_OK;
}

int pnp_fill(float* p, size_t n, float value, void* stream) {
    PNP_REQUIRE(p || n == 0, "pnp_fill: null pointer");
    if (n == 0) return PNP_OK;
    size_t nb = (n + NT - 1) / NT;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(fill_kernel, dim3((unsigned)nb), dim3(NT), 0, (hipStream_t)stream, p, n, value);
    PNP_CHECK_LAUNCH("fill_kernel");
    return PNP_OK;
}

size_t pnp_reduce_workspace_bytes(size_t n) {
    (void)n;
    return 1024 * sizeof(float);
}

int pnp_l2_loss(const float* w, size_t n, const float* chunk_l2, float* out, void* workspace, size_t workspace_bytes,
                void* stream) {
    PNP_REQUIRE(w && out && workspace && workspace_bytes >= pnp_reduce_workspace_bytes(n), "pnp_l2_loss: bad argument");
    hipStream_t st = (hipStream_t)stream;
    size_t nchunks = (n + PNP_OPT_CHUNK - 1) / PNP_OPT_CHUNK;
    int nblk = nchunks > 1024 ? 1024 : (int)(nchunks ? nchunks : 1);
    hipLaunchKernelGGL(l2_partial_kernel, dim3(nblk), dim3(NT), 0, st, w, n, chunk_l2, (float*)workspace);
    PNP_CHECK_LAUNCH("l2_partial_kernel");
    hipLaunchKernelGGL(sum_final_kernel, dim3(1), dim3(64), 0, st, (const float*)workspace, nblk, out);
    PNP_CHECK_LAUNCH("sum_final_kernel");
    return PNP_OK;
}

}  // extern "C"
