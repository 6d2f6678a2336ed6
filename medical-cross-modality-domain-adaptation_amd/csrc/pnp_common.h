// Shared host/device helpers for libpnp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "pnp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

void pnp_set_error(const char* fmt, ...);

#define PNP_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            pnp_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return PNP_ELAUNCH;                                                  \
        }                                                                        \
    } while (0)

#define PNP_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            pnp_set_error(__VA_ARGS__);   \
            return PNP_EINVAL;            \
        }                                 \
    } while (0)

// ---- dropout counter hash -------------------------------------------------------------------
// The reference's tf.nn.dropout RNG (TF-1.4 Philox stream, graph-seed dependent) is not
// reproducible outside TF, so the mask stream is OURS and is specified here; oracle/dropout.py
// restates it in numpy.  mask(idx) = (fmix32((idx * 0xCC9E2D51) ^ key) >> 8) >= thresh
//   key    = pnp_drop_key(seed, stream_id)        (host)
//   thresh = round((1-keep) * 2^24)               (host)
__host__ __device__ __forceinline__ uint32_t pnp_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}
__host__ __device__ __forceinline__ uint32_t pnp_drop_key(uint64_t seed, uint32_t stream_id) {
    uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    uint32_t k = pnp_fmix32(hi + 0x9E3779B9u * (stream_id + 1u));
    return pnp_fmix32(lo ^ k);
}
__host__ __device__ __forceinline__ uint32_t pnp_drop_thresh(float keep) {
    double t = (1.0 - (double)keep) * 16777216.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 16777216.0) t = 16777216.0;
    return (uint32_t)t;
}
__host__ __device__ __forceinline__ bool pnp_drop_keep(uint32_t idx, uint32_t key, uint32_t thresh) {
    uint32_t h = pnp_fmix32((idx * 0xCC9E2D51u) ^ key);
    return (h >> 8) >= thresh;
}

// ---- per-step scalars read from DEVICE memory (step capture: a hipGraph replays the launches of a recorded step, with every by-value
// argument frozen — the dropout seed and Adam's bias-corrected learning rate change every step, so a captured step reads them from a
// 16-byte device block instead; core.hip: pnp_step_params_bind / pnp_step_params_set).  Not bound (the default): by-value scalars.
struct pnp_step_params {
    unsigned long long drop_seed;      // replaces the `seed` argument of every dropout-carrying call
    float adam_lr_t;                   // replaces lr * sqrt(1 - beta2^t) / (1 - beta1^t) of pnp_adam_step
    float reserved;
};
const pnp_step_params* pnp_step_params_ptr();
__device__ __forceinline__ uint32_t pnp_eff_drop_key(uint32_t key, const pnp_step_params* sp, uint32_t stream_id) {
    return sp ? pnp_drop_key(sp->drop_seed, stream_id) : key;      // the SAME mask stream: key = f(seed, call site), wherever the seed lives
}

// Times one kernel launch with HIP events on its own stream when pnp_prof_enable(mask) selects its class (core.hip); a no-op otherwise.
// name = the kernel symbol as rocprofv3 prints it, so that bench.py's live numbers and the committed --kernel-trace summaries line up.
struct PnpProfScope {
    PnpProfScope(int cls, hipStream_t st, double flops, double bytes, const char* fmt, ...) __attribute__((format(printf, 6, 7)));
    ~PnpProfScope();
    PnpProfScope(const PnpProfScope&) = delete;
    // the scope owns its event pair and hands the FINISHED record over in the destructor: a pnp_prof_summary() between the constructor
    // and the destructor can neither drop the end event nor have it recorded on another scope's record
    bool on_;
    hipStream_t st_;
    hipEvent_t e0_, e1_;
    double flops_, bytes_;
    char name_[128];
};

static inline int pnp_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
