// core.hip — error channel, ABI version and device query of libpnp_hip.so.
#include <stdarg.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "pnp_common.h"

static thread_local char g_err[512] = "";

void pnp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- kernel-level timing (bench.py's roofline): HIP events on the launch stream around the dominant kernel of a call -----------
namespace {
struct ProfRec {
    std::string name;
    double flops, bytes;
    hipEvent_t e0, e1;
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof_recs;
std::atomic<int> g_prof_mask{0};      // read on every launch (relaxed): the mutex guards g_prof_recs only
constexpr size_t kProfCap = 1u << 18;
}  // namespace

PnpProfScope::PnpProfScope(int cls, hipStream_t st, double flops, double bytes, const char* fmt, ...)
    : on_(false), st_(st), e0_(nullptr), e1_(nullptr), flops_(flops), bytes_(bytes) {
    if (!(g_prof_mask.load(std::memory_order_relaxed) & cls)) return;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(name_, sizeof(name_), fmt, ap);
    va_end(ap);
    // timing only: no system-scope fence when the event completes (hipEventDisableSystemFence exists for exactly this)
    if (hipEventCreateWithFlags(&e0_, hipEventDisableSystemFence) != hipSuccess) return;
    if (hipEventCreateWithFlags(&e1_, hipEventDisableSystemFence) != hipSuccess) {
        (void)hipEventDestroy(e0_);
        return;
    }
    (void)hipEventRecord(e0_, st);
    on_ = true;
}

PnpProfScope::~PnpProfScope() {
    if (!on_) return;
    (void)hipEventRecord(e1_, st_);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_recs.size() >= kProfCap) {
        (void)hipEventDestroy(e0_);
        (void)hipEventDestroy(e1_);
        return;
    }
    g_prof_recs.push_back(ProfRec{name_, flops_, bytes_, e0_, e1_});
}

static const pnp_step_params* g_step_params = nullptr;
const pnp_step_params* pnp_step_params_ptr() { return g_step_params; }

__global__ void step_params_set_kernel(pnp_step_params* p, unsigned long long seed, float lr_t) {
    p->drop_seed = seed;
    p->adam_lr_t = lr_t;
    p->reserved = 0.f;
}

extern "C" {

int pnp_abi_version(void) { return 4; }

int pnp_step_params_bind(const void* dev_block) {
    g_step_params = (const pnp_step_params*)dev_block;
    return PNP_OK;
}

int pnp_step_params_set(void* dev_block, uint64_t drop_seed, float adam_lr_t, void* stream) {
    PNP_REQUIRE(dev_block, "pnp_step_params_set: null block");
    hipLaunchKernelGGL(step_params_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (pnp_step_params*)dev_block,
                       (unsigned long long)drop_seed, adam_lr_t);
    PNP_CHECK_LAUNCH("step_params_set_kernel");
    return PNP_OK;
}

int pnp_prof_enable(int mask) {
    g_prof_mask.store(mask, std::memory_order_relaxed);
    return PNP_OK;
}

int pnp_prof_summary(pnp_prof_row* rows, int max_rows) {
    std::vector<ProfRec> recs;
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        recs.swap(g_prof_recs);
    }
    std::map<std::string, pnp_prof_row> agg;
    for (ProfRec& r : recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.e1) == hipSuccess && hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
            pnp_prof_row& row = agg[r.name];
            if (row.launches == 0) {
                memset(&row, 0, sizeof(row));
                strncpy(row.name, r.name.c_str(), sizeof(row.name) - 1);
            }
            row.launches += 1;
            row.ms += ms;
            row.flops += r.flops;
            row.bytes += r.bytes;
        }
        (void)hipEventDestroy(r.e0);
        (void)hipEventDestroy(r.e1);
    }
    int n = 0;
    for (auto& kv : agg) {
        if (rows && n < max_rows) rows[n] = kv.second;
        ++n;
    }
    return n;
}

const char* pnp_last_error(void) { return g_err; }

int pnp_device_info(int device, int* cu_count, int* clock_khz, int* lds_bytes, char* arch, int arch_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        pnp_set_error("pnp_device_info: %s", hipGetErrorString(e));
        return PNP_ELAUNCH;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    if (lds_bytes) *lds_bytes = (int)p.maxSharedMemoryPerMultiProcessor;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return PNP_OK;
}

}  // extern "C"
