// Context life cycle and error reporting of libpicp_hip.so.
#include <stdlib.h>

#include "picp_common.h"

namespace picp {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return PICP_ERR_HIP;
}

} // namespace picp

using namespace picp;

extern "C" {

int picp_version(void) { return PICP_VERSION; }

const char* picp_last_error(void) { return g_error; }

int picp_create(picp_ctx** out, int device, void* stream) {
    PICP_REQUIRE(out != nullptr, "picp_create: null output pointer");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("picp_create: no HIP device is visible");
        return PICP_ERR_NO_DEVICE;
    }
    PICP_REQUIRE(device >= 0 && device < ndev, "picp_create: device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    PICP_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("picp_create: device %d is %s; this library carries gfx950 (MI355X) code only", device,
                  prop.gcnArchName);
        return PICP_ERR_NO_DEVICE;
    }
    DeviceGuard g(device);
    picp_ctx* ctx = new picp_ctx();
    memset(ctx, 0, sizeof(*ctx));
    ctx->device = device;
    ctx->stream = static_cast<hipStream_t>(stream);
    ctx->num_cus = prop.multiProcessorCount;
    // Workgroups per CU of the two scans, from sweeps on MI355X (2^20 .. 2^25 correspondences):
    // both scans keep two register sets of loads in flight per lane, so more resident waves buy no
    // bandwidth and only add partial records, launch ramp and reduction work.  Stage 1: two
    // 256-thread workgroups per CU (one for very large inputs, see grid_for callers); stage 2: one
    // 512-thread workgroup per CU.
    ctx->max_blocks1 = 2 * ctx->num_cus;
    ctx->max_blocks2 = ctx->num_cus;
    // Tuning overrides (workgroups per CU), for experiments only.
    if (const char* e1 = getenv("PICP_STAGE1_WG_PER_CU")) { int k = atoi(e1); if (k >= 1 && k <= 8) ctx->max_blocks1 = k * ctx->num_cus; }
    if (const char* e2 = getenv("PICP_STAGE2_WG_PER_CU")) { int k = atoi(e2); if (k >= 1 && k <= 4) ctx->max_blocks2 = k * ctx->num_cus; }
    ctx->stage2_reverse = -1; // by size (picp_stage2)
    if (const char* e3 = getenv("PICP_STAGE2_REVERSE")) ctx->stage2_reverse = atoi(e3) != 0;
    ctx->scan_nontemporal = -1; // by size (scan_nontemporal())
    if (const char* e4 = getenv("PICP_NONTEMPORAL")) ctx->scan_nontemporal = atoi(e4) != 0;
    hipError_t e = hipMalloc(&ctx->part1, sizeof(double) * PICP_S1_DOUBLES * ctx->max_blocks1);
    if (e == hipSuccess) e = hipMalloc(&ctx->part2, sizeof(double) * PICP_S2_DOUBLES * ctx->max_blocks2);
    if (e == hipSuccess) e = hipMalloc(&ctx->scratch, sizeof(double) * SCR_DOUBLES);
    if (e == hipSuccess) e = hipHostMalloc(&ctx->pinned, sizeof(double) * PINNED_DOUBLES, hipHostMallocDefault);
    if (e == hipSuccess) e = hipMemset(ctx->scratch, 0, sizeof(double) * SCR_DOUBLES);
    if (e == hipSuccess) e = hipMalloc(&ctx->arrive, sizeof(unsigned) * 64);
    if (e == hipSuccess) e = hipMemset(ctx->arrive, 0, sizeof(unsigned) * 64);
    if (e == hipSuccess) e = hipMalloc(&ctx->rslots, sizeof(double) * 2 * PICP_S2_DOUBLES * resident_slot_rows(ctx));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->loop_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->assoc_done, hipEventDisableTiming);
    for (int k = 0; k < PICP_ASYNC_SLOTS && e == hipSuccess; ++k)
        e = hipEventCreateWithFlags(&ctx->slot_done[k], hipEventDisableTiming);
    ctx->resident_limit = INT64_MAX; // i.e. whatever fits (resident_capacity)
    ctx->resident_spin_limit = 1u << 18;
    ctx->persistent_limit = INT64_MAX;
    ctx->peer_spin_limit = 1u << 20; // about a second; sixteen times that for a launch's first exchange, since the ranks' hosts reach their launches at different times
    if (const char* ex = getenv("PICP_PEER_SPIN_LIMIT")) { int k = atoi(ex); if (k >= 0) ctx->peer_spin_limit = (unsigned)k; }
    if (const char* ep = getenv("PICP_PERSISTENT_LIMIT")) ctx->persistent_limit = atoll(ep);
    ctx->persistent_hold = 1;
    if (const char* eh = getenv("PICP_PERSISTENT_HOLD")) ctx->persistent_hold = atoi(eh) != 0;
    if (const char* er = getenv("PICP_RESIDENT")) { if (atoi(er) == 0) ctx->resident_limit = 0; }
    if (const char* es = getenv("PICP_RESIDENT_SPIN_LIMIT")) { int k = atoi(es); if (k >= 0) ctx->resident_spin_limit = (unsigned)k; }
    if (e != hipSuccess) {
        picp_destroy(ctx);
        return hip_fail(e, "picp_create allocations");
    }
    if (int rc = noalloc_bind(ctx)) { // address of the static workspace on this device
        picp_destroy(ctx);
        return rc;
    }
    *out = ctx;
    return PICP_OK;
}

int picp_destroy(picp_ctx* ctx) {
    if (ctx == nullptr) return PICP_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)picp_comm_destroy(ctx);
    if (ctx->peer_world) (void)picp_peer_destroy(ctx);
    if (ctx->part1) (void)hipFree(ctx->part1);
    if (ctx->part2) (void)hipFree(ctx->part2);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->arrive) (void)hipFree(ctx->arrive);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->rslots) (void)hipFree(ctx->rslots);
    if (ctx->loop_done) (void)hipEventDestroy(ctx->loop_done);
    if (ctx->assoc_done) (void)hipEventDestroy(ctx->assoc_done);
    if (ctx->prof0) (void)hipEventDestroy(ctx->prof0);
    if (ctx->prof1) (void)hipEventDestroy(ctx->prof1);
    for (int k = 0; k < PICP_ASYNC_SLOTS; ++k)
        if (ctx->slot_done[k]) (void)hipEventDestroy(ctx->slot_done[k]);
    if (ctx->assoc_count) (void)hipFree(ctx->assoc_count);
    if (ctx->assoc_first) (void)hipFree(ctx->assoc_first);
    if (ctx->assoc_offset) (void)hipFree(ctx->assoc_offset);
    if (ctx->assoc_blocksum) (void)hipFree(ctx->assoc_blocksum);
    if (ctx->assoc_key) (void)hipFree(ctx->assoc_key);
    if (ctx->assoc_table) (void)hipFree(ctx->assoc_table);
    if (ctx->assoc_hash) (void)hipFree(ctx->assoc_hash);
    delete ctx;
    return PICP_OK;
}

int picp_set_stream(picp_ctx* ctx, void* stream) {
    PICP_REQUIRE(ctx != nullptr, "picp_set_stream: null ctx");
    if (ctx->stream == static_cast<hipStream_t>(stream)) return PICP_OK;
    // A single-launch loop may still be running on the old stream, and it cleans the set of exchange slots
    // that the next loop uses: nothing orders that with work on the new stream.  Wait for it.
    DeviceGuard g(ctx->device);
    PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->stream = static_cast<hipStream_t>(stream);
    return PICP_OK;
}

int picp_set_resident_limit(picp_ctx* ctx, int64_t max_correspondences) {
    PICP_REQUIRE(ctx != nullptr, "picp_set_resident_limit: null ctx");
    PICP_REQUIRE(max_correspondences >= 0, "picp_set_resident_limit: negative limit");
    ctx->resident_limit = max_correspondences;
    return PICP_OK;
}

int picp_set_stage2_order(picp_ctx* ctx, int32_t order) {
    PICP_REQUIRE(ctx != nullptr, "picp_set_stage2_order: null ctx");
    PICP_REQUIRE(order >= -1 && order <= 1, "picp_set_stage2_order: order %d is not -1, 0 or 1", (int)order);
    ctx->stage2_reverse = order;
    return PICP_OK;
}

int64_t picp_resident_capacity(picp_ctx* ctx) { return ctx ? resident_capacity(ctx) : 0; }
int64_t picp_streaming_held(picp_ctx* ctx, int64_t M) { return (ctx && M > 0) ? persistent_held(ctx, M) : 0; }

int picp_resident_aborts(picp_ctx* ctx) { return ctx ? ctx->resident_aborts : 0; }
int picp_immoderate_exits(picp_ctx* ctx) { return ctx ? ctx->immoderate_exits : 0; }

int picp_synchronize(picp_ctx* ctx) {
    PICP_REQUIRE(ctx != nullptr, "picp_synchronize: null ctx");
    DeviceGuard g(ctx->device);
    PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PICP_OK;
}

} // extern "C"
