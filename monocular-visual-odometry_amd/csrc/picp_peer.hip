// Sharded runs with the exchange inside the kernel: set-up of the exchange memory between the GPUs.
//
// Every rank owns a few kilobytes of device memory that its peers write into while the loops run
// (picp_resident.hip: cross_total).  It is allocated uncached, so that neither the writers' nor the
// readers' L2 keeps a copy, and handed to the other processes as a HIP IPC handle.
#include <stdlib.h>

#include "picp_common.h"

using namespace picp;

extern "C" {

int picp_peer_create(picp_ctx* ctx, int32_t rank, int32_t world, void* handle_host) {
    PICP_REQUIRE(ctx != nullptr && handle_host != nullptr, "picp_peer_create: null argument");
    PICP_REQUIRE(world >= 1 && world <= PICP_PEER_MAX_RANKS && rank >= 0 && rank < world,
                 "picp_peer_create: rank %d of %d (at most %d ranks)", (int)rank, (int)world, PICP_PEER_MAX_RANKS);
    PICP_REQUIRE(ctx->peer_world == 0, "picp_peer_create: the ctx already has an exchange (picp_peer_destroy first)");
    static_assert(sizeof(hipIpcMemHandle_t) == PICP_PEER_HANDLE_BYTES, "handle size");
    DeviceGuard g(ctx->device);
    double* own = nullptr;
    const size_t bytes = sizeof(double) * (size_t)peer_slot_doubles();
    // PICP_PEER_MEMORY=finegrained skips the uncached attempt (to try the other flavour on a machine where both work)
    const char* want = getenv("PICP_PEER_MEMORY");
    const bool try_uncached = !(want && strcmp(want, "finegrained") == 0);
    hipError_t e = try_uncached ? hipExtMallocWithFlags(reinterpret_cast<void**>(&own), bytes, hipDeviceMallocUncached)
                                : hipErrorInvalidValue;
    int kind = PICP_PEER_MEMORY_UNCACHED;
    if (e != hipSuccess) {
        (void)hipGetLastError();
        e = hipExtMallocWithFlags(reinterpret_cast<void**>(&own), bytes, hipDeviceMallocFinegrained);
        kind = PICP_PEER_MEMORY_FINEGRAINED;
    }
    if (e != hipSuccess) return hip_fail(e, "picp_peer_create: allocation of the exchange memory");
    peer_fill(ctx, own);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, own);
    if (e != hipSuccess) {
        (void)hipFree(own);
        return hip_fail(e, "picp_peer_create: hipIpcGetMemHandle");
    }
    memcpy(handle_host, &h, sizeof(h));
    memset(ctx->peer_slots, 0, sizeof(ctx->peer_slots));
    ctx->peer_slots[rank] = own;
    ctx->peer_rank = rank;
    ctx->peer_world = world;
    ctx->peer_connected = (world == 1);
    ctx->peer_broken = 0;
    ctx->peer_epoch = 0;
    ctx->peer_memory_kind = kind;
    return PICP_OK;
}

int picp_peer_memory_kind(picp_ctx* ctx) { return (ctx && ctx->peer_world >= 1) ? ctx->peer_memory_kind : PICP_PEER_MEMORY_NONE; }

int picp_device_bus_id(picp_ctx* ctx, char* out, int64_t bytes) {
    PICP_REQUIRE(ctx != nullptr && out != nullptr && bytes >= PICP_BUS_ID_BYTES, "picp_device_bus_id: needs a buffer of %d bytes",
                 PICP_BUS_ID_BYTES);
    memset(out, 0, (size_t)bytes);
    PICP_HIP_CHECK(hipDeviceGetPCIBusId(out, (int)bytes, ctx->device));
    return PICP_OK;
}

int picp_peer_can_access(picp_ctx* ctx, const char* peer_bus_id, int32_t* can) {
    PICP_REQUIRE(ctx != nullptr && peer_bus_id != nullptr && can != nullptr, "picp_peer_can_access: null argument");
    *can = -1;
    int peer = -1;
    if (hipDeviceGetByPCIBusId(&peer, peer_bus_id) != hipSuccess) { // the peer's GPU is not visible to this process
        (void)hipGetLastError();
        return PICP_OK;
    }
    if (peer == ctx->device) { // ranks sharing one GPU (rehearsals): its own memory
        *can = 1;
        return PICP_OK;
    }
    int yes = 0;
    if (hipDeviceCanAccessPeer(&yes, ctx->device, peer) != hipSuccess) {
        (void)hipGetLastError();
        return PICP_OK;
    }
    *can = yes ? 1 : 0;
    return PICP_OK;
}

int picp_peer_connect(picp_ctx* ctx, const void* handles_host) {
    PICP_REQUIRE(ctx != nullptr && handles_host != nullptr, "picp_peer_connect: null argument");
    PICP_REQUIRE(ctx->peer_world >= 1, "picp_peer_connect: picp_peer_create first");
    PICP_REQUIRE(!ctx->peer_connected || ctx->peer_world == 1, "picp_peer_connect: already connected");
    DeviceGuard g(ctx->device);
    const char* hs = static_cast<const char*>(handles_host);
    for (int r = 0; r < ctx->peer_world; ++r) {
        if (r == ctx->peer_rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, hs + (size_t)r * PICP_PEER_HANDLE_BYTES, sizeof(h));
        void* p = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) {
            for (int k = 0; k < r; ++k)
                if (k != ctx->peer_rank && ctx->peer_slots[k]) {
                    (void)hipIpcCloseMemHandle(ctx->peer_slots[k]);
                    ctx->peer_slots[k] = nullptr;
                }
            return hip_fail(e, "picp_peer_connect: hipIpcOpenMemHandle");
        }
        ctx->peer_slots[r] = static_cast<double*>(p);
    }
    ctx->peer_connected = 1;
    return PICP_OK;
}

int picp_peer_destroy(picp_ctx* ctx) {
    PICP_REQUIRE(ctx != nullptr, "picp_peer_destroy: null ctx");
    if (ctx->peer_world == 0) return PICP_OK;
    DeviceGuard g(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (int r = 0; r < ctx->peer_world; ++r) {
        if (ctx->peer_slots[r] == nullptr) continue;
        if (r == ctx->peer_rank) (void)hipFree(ctx->peer_slots[r]);
        else (void)hipIpcCloseMemHandle(ctx->peer_slots[r]);
        ctx->peer_slots[r] = nullptr;
    }
    ctx->peer_world = 0;
    ctx->peer_connected = 0;
    ctx->peer_broken = 0;
    ctx->peer_epoch = 0;
    ctx->peer_memory_kind = PICP_PEER_MEMORY_NONE;
    return PICP_OK;
}

int picp_peer_world(picp_ctx* ctx) {
    return (ctx && ctx->peer_connected && !ctx->peer_broken) ? ctx->peer_world : 0;
}

int picp_set_grid_limit(picp_ctx* ctx, int32_t max_workgroups) {
    PICP_REQUIRE(ctx != nullptr && max_workgroups >= 0, "picp_set_grid_limit: bad argument");
    ctx->grid_limit = max_workgroups;
    return PICP_OK;
}

} // extern "C"
