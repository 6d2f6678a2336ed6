// comm.hip — the data-parallel exchange step of the training path: in-place sum all-reduce of a device buffer over RCCL (xGMI).
//
// The reference is single-GPU (train_segmenter.py:20, train_gan.py:18 pin CUDA_VISIBLE_DEVICES); BASELINE.json asks for the
// mini-batch to be sharded over the GPUs of a node with the gradients all-reduced over RCCL on a side HIP stream.  These entry
// points are that exchange step behind the C-ABI (pnp_comm_{init,allreduce,destroy}): the caller owns the stream
// and the events that order it against the backward pass; nothing here synchronises the host.
//
// librccl.so is resolved at run time (dlopen), not at link time: libpnp_hip.so must load on hosts without RCCL (the CPU-only
// build container, single-GPU boxes), and inside a PyTorch process it must bind to the SAME copy PyTorch's wheel ships (one HIP
// runtime per process — see _lib.py): pnp_comm_load(path) lets the host side name that copy.
#include <dlfcn.h>
#include <string.h>
#include <mutex>
#include <rccl/rccl.h>      // types and enums only; no symbol of librccl is referenced at link time
#include "pnp_common.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int*) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_locked(const char* path) {
    if (g_rccl.handle) return PNP_OK;
    void* h = nullptr;
    if (path && *path) h = dlopen(path, RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);      // a copy this process already mapped
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        pnp_set_error("pnp_comm: cannot load librccl.so (%s)", dlerror());
        return PNP_ECOMM;
    }
#define PNP_SYM(field, sym)                                                          \
    g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, sym));          \
    if (!g_rccl.field) {                                                             \
        pnp_set_error("pnp_comm: librccl.so lacks %s", sym);                         \
        dlclose(h);                                                                  \
        return PNP_ECOMM;                                                            \
    }
    PNP_SYM(GetUniqueId, "ncclGetUniqueId")
    PNP_SYM(CommInitRank, "ncclCommInitRank")
    PNP_SYM(AllReduce, "ncclAllReduce")
    PNP_SYM(CommDestroy, "ncclCommDestroy")
    PNP_SYM(GetErrorString, "ncclGetErrorString")
    PNP_SYM(GetVersion, "ncclGetVersion")
#undef PNP_SYM
    g_rccl.handle = h;
    return PNP_OK;
}

int ensure_loaded() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return load_locked(nullptr);
}

#define PNP_RCCL(call, what)                                                                   \
    do {                                                                                       \
        ncclResult_t r__ = (call);                                                             \
        if (r__ != ncclSuccess) {                                                              \
            pnp_set_error("%s: RCCL error %d: %s", what, (int)r__, g_rccl.GetErrorString(r__)); \
            return PNP_ECOMM;                                                                  \
        }                                                                                      \
    } while (0)

struct Comm {
    ncclComm_t comm;
    int rank, world;
};

}  // namespace

extern "C" {

int pnp_comm_load(const char* librccl_path) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    return load_locked(librccl_path);
}

int pnp_comm_version(int* version) {
    if (int e = ensure_loaded()) return e;
    PNP_REQUIRE(version, "pnp_comm_version: null pointer");
    PNP_RCCL(g_rccl.GetVersion(version), "pnp_comm_version");
    return PNP_OK;
}

int pnp_comm_unique_id(uint8_t* id) {
    if (int e = ensure_loaded()) return e;
    PNP_REQUIRE(id, "pnp_comm_unique_id: null pointer");
    static_assert(sizeof(ncclUniqueId) == PNP_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    PNP_RCCL(g_rccl.GetUniqueId(&u), "pnp_comm_unique_id");
    memcpy(id, &u, sizeof(u));
    return PNP_OK;
}

int pnp_comm_init(int32_t rank, int32_t world, const uint8_t* id, void** comm_out) {
    if (int e = ensure_loaded()) return e;
    PNP_REQUIRE(id && comm_out && world > 0 && rank >= 0 && rank < world, "pnp_comm_init: bad argument (rank %d of %d)", rank, world);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    Comm* c = new Comm{nullptr, rank, world};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, u, rank);      // collective: every rank of the job calls it (current device)
    if (r != ncclSuccess) {
        pnp_set_error("pnp_comm_init: RCCL error %d: %s", (int)r, g_rccl.GetErrorString(r));
        delete c;
        return PNP_ECOMM;
    }
    *comm_out = c;
    return PNP_OK;
}

int pnp_comm_allreduce(void* comm, void* buf, size_t n, int32_t dtype, void* stream) {
    PNP_REQUIRE(comm && (buf || n == 0), "pnp_comm_allreduce: null pointer");
    PNP_REQUIRE(dtype == PNP_DTYPE_F32 || dtype == PNP_DTYPE_F64, "pnp_comm_allreduce: dtype %d (f32 = %d and f64 = %d only)", dtype,
                PNP_DTYPE_F32, PNP_DTYPE_F64);
    if (n == 0) return PNP_OK;
    Comm* c = static_cast<Comm*>(comm);
    PNP_RCCL(g_rccl.AllReduce(buf, buf, n, dtype == PNP_DTYPE_F32 ? ncclFloat32 : ncclFloat64, ncclSum, c->comm, (hipStream_t)stream),
             "pnp_comm_allreduce");
    return PNP_OK;
}

int pnp_comm_destroy(void* comm) {
    if (!comm) return PNP_OK;
    Comm* c = static_cast<Comm*>(comm);
    ncclResult_t r = g_rccl.CommDestroy(c->comm);
    delete c;
    if (r != ncclSuccess) {
        pnp_set_error("pnp_comm_destroy: RCCL error %d: %s", (int)r, g_rccl.GetErrorString(r));
        return PNP_ECOMM;
    }
    return PNP_OK;
}

}  // extern "C"
