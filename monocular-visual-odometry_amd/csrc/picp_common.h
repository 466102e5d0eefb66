// Host-side plumbing shared by the translation units of libpicp_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/picp_hip.h"

struct picp_ctx {
    int device;
    hipStream_t stream;
    int num_cus;
    int max_blocks1;  // workgroups (= partial records) of a stage-1 launch
    int max_blocks2;  // workgroups (= partial records) of a stage-2 launch
    double* part1;    // [max_blocks1][PICP_S1_DOUBLES]
    double* part2;    // [max_blocks2][PICP_S2_DOUBLES]
    int rows1, rows2; // rows written by the last stage-1 / stage-2 launch
    int stage2_reverse; // stage 2 walks the arrays against stage 1's direction: 1 always, 0 never, -1 by size
    int scan_nontemporal; // the scans' loads carry the non-temporal hint: 1 always, 0 never, -1 by size
    double* scratch;  // small device scratch: state block + s1 + s2 + misc (see offsets below)
    unsigned* arrive; // [2] arrival counters of the sharded rounds' scans (zero between launches)
    // association workspace, grown on demand
    int64_t* assoc_count;  // [cap] matches per query row
    int64_t* assoc_first;  // [cap] first matching row of set 2
    int64_t* assoc_offset; // [cap + 1] exclusive prefix sum of assoc_count
    int64_t* assoc_blocksum;
    uint32_t* assoc_key;   // [key_cap] low words of component 0 of set 2
    int64_t assoc_cap;
    int64_t assoc_key_cap;
    unsigned long long* assoc_table; // [table_cap] hash-join table and the join's control words behind it
    int64_t assoc_table_cap;
    int64_t assoc_table_clean; // leading words of the table known to be zero (cleared behind the last join)
    hipEvent_t assoc_done;     // recorded behind a join's last kernel, before the table is cleared again
    uint64_t* assoc_hash;  // [hash_cap] hashes of the rows of set 2, between the two launches of the table build
    int64_t assoc_hash_cap;
    int assoc_mode;           // 0 auto, 1 all-pairs scan, 2 hash join
    int assoc_hash_fallbacks; // hash joins abandoned for the all-pairs scan (heavy duplication)
    double* pinned; // host-pinned staging for small read-backs
    double* noalloc; // this device's address of the static workspace of picp_linearize_noalloc
    // register-resident loop (picp_resident.hip)
    double* rslots;                // [2][resident_slot_rows][PICP_S2_DOUBLES] exchange records: two sets of 3 slots of a launch's grid
    int slot_set;                  // the set the next single-launch loop exchanges through
    int slots_clean[2];            // the set holds sentinels everywhere
    int64_t resident_limit;        // largest M the resident loop is used for (0: never)
    unsigned resident_spin_limit;  // polls before a waiting workgroup abandons the launch
    int resident_aborts;           // launches abandoned so far (each was repeated with the streaming kernels)
    int abort_streak;              // consecutive single-launch loops that gave up
    int single_launch_holdoff;     // picp_run calls that still go straight to the stage launches after a streak
    int64_t persistent_limit;      // largest M the streaming single-launch loop is used for (0: never)
    int persistent_hold;           // the streaming single-launch loop keeps part of its input in LDS (PICP_PERSISTENT_HOLD=0: never)
    hipEvent_t loop_done;          // recorded behind a single-launch loop; the host waits for it, not for the stream
    hipEvent_t slot_done[PICP_ASYNC_SLOTS]; // the same for loops enqueued with picp_run_enqueue, one per slot
    int grid_limit;                // cap on the workgroups of the single-launch loops (0: one per CU)
    int full_select_calls;         // on-chip loops that still use the form without assumptions on magnitudes
    int immoderate_exits;          // on-chip loops that left because a value was not of moderate magnitude
    // sharded runs (picp_comm.hip)
    void* comm;                    // RCCL communicator of this rank, or null
    int comm_rank, comm_world;
    // sharded runs with the exchange inside the kernel (picp_peer.hip)
    double* peer_slots[PICP_PEER_MAX_RANKS]; // exchange memory of every rank, mapped here; [peer_rank] is this GPU's own
    int peer_rank, peer_world;     // peer_world == 0: not set up
    int peer_connected;            // the peers' memory is mapped
    int peer_broken;               // a launch gave up at an exchange: the slots are in an unknown state
    int peer_memory_kind;          // how the exchange memory was allocated: 0 none, 1 uncached, 2 fine-grained
    unsigned peer_epoch;           // exchanges between the GPUs completed by earlier launches
    unsigned peer_spin_limit;      // polls before a workgroup waiting for a peer gives up
    int time_launches;             // picp_run_sharded brackets its launch with events (picp_set_launch_timing)
    hipEvent_t prof0, prof1;
    double last_launch_ms;
};

namespace picp {

// offsets (in doubles) into ctx->scratch
enum : int {
    SCR_STATE = 0,                              // PICP_STATE_DOUBLES
    SCR_S1 = 128,                               // PICP_S1_DOUBLES
    SCR_S2 = 192,                               // PICP_S2_DOUBLES
    SCR_MISC = 256,                             // 64 doubles for tiny test kernels
    SCR_STATE2 = 384,                           // second state block of the sharded rounds (PICP_STATE_DOUBLES)
    SCR_DOUBLES = 512,
};
enum : int { PINNED_DOUBLES = 4096, PINNED_SLOT0 = 2048, PINNED_SLOT_STRIDE = 128 }; // final states of enqueued loops

// Value of the host-visible stop flag after an on-chip loop in its lean form met a value that is not
// finite and of moderate magnitude: nothing was computed, the loop runs again in its other form.
constexpr double STOP_IMMODERATE = 2.0;

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

// picp_icp.hip
int noalloc_bind(picp_ctx* ctx);
int sharded_rounds(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                   int64_t M, double* state, const picp_camera* cam, const picp_schedule* sched, double* s1, double* s2,
                   double* hist, int hist_rows, int rounds, int (*allreduce)(picp_ctx*, double*, size_t));

// picp_resident.hip
int64_t resident_capacity(const picp_ctx* ctx);
int resident_slot_rows(const picp_ctx* ctx); // records to allocate for the in-kernel exchange (rslots)
int enqueue_resident(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, int64_t M, const double* w_T_c0_host, double* state, const picp_camera* cam,
                     const picp_schedule* sched, double* hist, int hist_rows, double* host_state, hipEvent_t e0,
                     hipEvent_t e1, bool sharded = false);

int enqueue_persistent(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                       const double* z, int64_t M, const double* w_T_c0_host, double* state, const picp_camera* cam,
                       const picp_schedule* sched, double* hist, int hist_rows, double* host_state, hipEvent_t e0,
                       hipEvent_t e1, bool sharded = false);
void peer_fill(picp_ctx* ctx, double* slots);
int peer_slot_doubles();
int64_t persistent_held(const picp_ctx* ctx, int64_t M);

// Whether the second scan of an iteration walks the arrays against the first one's direction: it
// then starts with what stage 1 read last and ends with what the next stage 1 reads first, which
// the 256 MB memory-side cache still holds when the arrays are not much larger than it (measured per
// iteration: 2^23 correspondences 127 -> 112 us, 12 M 181 -> 169 us; no gain from 2^24 on, a 3 % loss
// at 2^25; nothing to gain while the arrays fit the L2s).
inline bool stage2_reverse(const picp_ctx* ctx, int64_t M) {
    const int64_t bytes = M * 5 * (int64_t)sizeof(double);
    return ctx->stage2_reverse < 0 ? (bytes > (int64_t(64) << 20) && bytes <= (int64_t(512) << 20))
                                   : ctx->stage2_reverse != 0;
}
// Whether the scans of an M-correspondence input use non-temporal loads: arrays beyond what the memory-side
// cache can help with (the same 512 MB up to which the second scan runs backwards; swept in 48 MB steps around
// it: 503 MB a tie, 545 MB 169 -> 155 us per iteration, 671 MB 221 -> 199 us).  PICP_NONTEMPORAL=0/1 overrides.
inline bool scan_nontemporal(const picp_ctx* ctx, int64_t M) {
    const int64_t bytes = M * 5 * (int64_t)sizeof(double);
    return ctx->scan_nontemporal < 0 ? (bytes > (int64_t(512) << 20) && !stage2_reverse(ctx, M)) : ctx->scan_nontemporal != 0;
}

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = (hipSetDevice(dev) == hipSuccess);
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

int64_t batched_resident_capacity();
int enqueue_batched_resident(picp_ctx* ctx, int B, int64_t max_count, const double* u, const double* v, const double* x,
                             const double* y, const double* z, const int64_t* offsets, const double* poses0,
                             const picp_camera* cam, const picp_schedule* sched, picp_result* results, double* hist,
                             int hist_rows);

} // namespace picp

#define PICP_HIP_CHECK(expr)                                           \
    do {                                                               \
        hipError_t _e = (expr);                                        \
        if (_e != hipSuccess) return picp::hip_fail(_e, #expr);        \
    } while (0)

#define PICP_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            picp::set_error(__VA_ARGS__);  \
            return PICP_ERR_INVALID;       \
        }                                  \
    } while (0)
