// Sharded runs: the two per-iteration all-reduces issued on the ctx's own stream.
//
// With one process per GPU every rank scans its slice of the correspondences; the two sums an
// iteration needs (src/ProjectiveICP.py:301 and :313-317) are then all-reduced over the ranks.
// Here the collective is RCCL's ncclAllReduce enqueued on the same stream as the scan kernels, so
// an iteration is one in-order chain
//     stage 1 -> all-reduce(4 doubles) -> stage 2 -> all-reduce(32 doubles)
// (each scan folds its own records, the solve step runs in the next stage 1's prologue: picp_icp.hip)
// with no hop to another stream and back around each collective, and the host enqueues whole
// rounds with one call.  RCCL is loaded at run time (dlopen), so the library itself does not
// depend on it: single-GPU users never touch it.
#include <dlfcn.h>

#include "picp_common.h"

namespace picp {

// The few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes passed by value,
// ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0).
struct UniqueId {
    char internal[128];
};
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*CommDestroyFn)(void*);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef const char* (*GetErrorStringFn)(int);
constexpr int RCCL_FLOAT64 = 8, RCCL_SUM = 0;

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllReduceFn all_reduce = nullptr;
    GetErrorStringFn error_string = nullptr;
};

static Rccl load_rccl() {
    Rccl lib;
    // a copy the process has already loaded (PyTorch carries one) is found by its soname
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        lib.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (lib.handle) break;
    }
    if (lib.handle) {
        lib.get_unique_id = (GetUniqueIdFn)dlsym(lib.handle, "ncclGetUniqueId");
        lib.comm_init_rank = (CommInitRankFn)dlsym(lib.handle, "ncclCommInitRank");
        lib.comm_destroy = (CommDestroyFn)dlsym(lib.handle, "ncclCommDestroy");
        lib.all_reduce = (AllReduceFn)dlsym(lib.handle, "ncclAllReduce");
        lib.error_string = (GetErrorStringFn)dlsym(lib.handle, "ncclGetErrorString");
    }
    return lib;
}

// The loaded library, or null when it (or one of the five symbols) is missing.  Loaded once per
// process (a function-local static: safe when two threads with their own contexts get here together).
static Rccl* rccl() {
    static Rccl lib = load_rccl();
    const bool ok = lib.handle && lib.get_unique_id && lib.comm_init_rank && lib.comm_destroy && lib.all_reduce;
    return ok ? &lib : nullptr;
}

static int rccl_fail(int code, const char* what) {
    Rccl* r = rccl();
    const char* msg = (r && r->error_string) ? r->error_string(code) : "";
    set_error("RCCL error %d (%s) in %s", code, msg ? msg : "", what);
    return PICP_ERR_COMM;
}

int allreduce_sum(picp_ctx* ctx, double* buf, size_t count) {
    if (ctx->comm == nullptr) return PICP_OK; // one shard, no exchange
    const int rc = rccl()->all_reduce(buf, buf, count, RCCL_FLOAT64, RCCL_SUM, ctx->comm, ctx->stream);
    if (rc != 0) return rccl_fail(rc, "ncclAllReduce");
    return PICP_OK;
}

} // namespace picp

using namespace picp;

extern "C" {

int picp_comm_available(void) { return rccl() != nullptr ? 1 : 0; }

int picp_comm_unique_id(void* id, int64_t bytes) {
    PICP_REQUIRE(id != nullptr && bytes >= PICP_COMM_ID_BYTES, "picp_comm_unique_id: needs a buffer of %d bytes",
                 PICP_COMM_ID_BYTES);
    Rccl* r = rccl();
    if (r == nullptr) {
        const char* why = dlerror();
        set_error("picp_comm_unique_id: librccl.so could not be loaded (%s)", why ? why : "symbols missing");
        return PICP_ERR_COMM;
    }
    UniqueId uid;
    memset(&uid, 0, sizeof(uid));
    const int rc = r->get_unique_id(&uid);
    if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(id, &uid, sizeof(uid));
    return PICP_OK;
}

int picp_comm_init(picp_ctx* ctx, const void* id, int32_t rank, int32_t world) {
    PICP_REQUIRE(ctx != nullptr && id != nullptr, "picp_comm_init: null argument");
    PICP_REQUIRE(world >= 1 && rank >= 0 && rank < world, "picp_comm_init: rank %d of %d", (int)rank, (int)world);
    PICP_REQUIRE(ctx->comm == nullptr, "picp_comm_init: the ctx already has a communicator");
    Rccl* r = rccl();
    if (r == nullptr) {
        set_error("picp_comm_init: librccl.so could not be loaded");
        return PICP_ERR_COMM;
    }
    DeviceGuard g(ctx->device);
    UniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    void* comm = nullptr;
    const int rc = r->comm_init_rank(&comm, world, uid, rank);
    if (rc != 0) return rccl_fail(rc, "ncclCommInitRank");
    ctx->comm = comm;
    ctx->comm_rank = rank;
    ctx->comm_world = world;
    return PICP_OK;
}

int picp_comm_destroy(picp_ctx* ctx) {
    PICP_REQUIRE(ctx != nullptr, "picp_comm_destroy: null ctx");
    if (ctx->comm == nullptr) return PICP_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int rc = rccl()->comm_destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_world = 0;
    ctx->comm_rank = 0;
    if (rc != 0) return rccl_fail(rc, "ncclCommDestroy");
    return PICP_OK;
}

int picp_comm_world(picp_ctx* ctx) { return (ctx && ctx->comm) ? ctx->comm_world : 0; }

int picp_comm_allreduce(picp_ctx* ctx, double* buf, int64_t count) {
    PICP_REQUIRE(ctx != nullptr && buf != nullptr && count >= 0, "picp_comm_allreduce: bad argument");
    PICP_REQUIRE(ctx->comm != nullptr, "picp_comm_allreduce: no communicator (picp_comm_init)");
    DeviceGuard g(ctx->device);
    return allreduce_sum(ctx, buf, (size_t)count);
}

int picp_sharded_rounds(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                        const double* z, int64_t M, double* state, const picp_camera* cam, const picp_schedule* sched,
                        double* s1, double* s2, double* hist, int32_t hist_rows, int32_t rounds) {
    PICP_REQUIRE(ctx && state && cam && sched && s1 && s2, "picp_sharded_rounds: null argument");
    PICP_REQUIRE(rounds >= 0, "picp_sharded_rounds: negative number of rounds");
    PICP_REQUIRE(M >= 0 && (M == 0 || (u && v && x && y && z)), "picp_sharded_rounds: bad correspondence arrays");
    DeviceGuard g(ctx->device);
    return sharded_rounds(ctx, u, v, x, y, z, M, state, cam, sched, s1, s2, hist, (int)hist_rows, (int)rounds, allreduce_sum);
}

} // extern "C"
