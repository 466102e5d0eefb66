// Projective-ICP pose loop on gfx950: the two streaming scans over the correspondences, the
// single-lane solve + schedule step, the whole-loop-in-one-workgroup batched kernel, and the
// C ABI around them (include/picp_hip.h).
//
// Per iteration the reference needs a global mean of chi before it can decide which
// correspondences enter H and b (src/ProjectiveICP.py:301-310), so an iteration is two scans:
//   stage 1: in-image test, chi, {sum chi, count} of chi <= threshold            40 B / pair
//   stage 2: mean filter, Jacobian, H/b sums of both candidate inlier sets       40 B / pair
// Both are HBM-bound streams of five float64 arrays; every workgroup leaves one record of
// partial sums, and the next kernel in the chain folds those records in its prologue (a kernel
// boundary is cheaper here than an in-kernel device-wide barrier).
#include <hip/hip_ext.h>
#include <stdlib.h>

#include "picp_common.h"
#include "picp_device.h"

namespace picp {

constexpr int NT = 256;  // threads per workgroup of the stage-1 scan, the batched kernel and the small kernels
constexpr int NT2 = 512; // stage-2 scan: one workgroup of eight waves per CU, so a launch leaves at most one partial
                         // record per CU to fold

struct Pose16 {
    double m[16];
};

__global__ void state_init_kernel(double* st, Pose16 pose, double thr, double lambda, int num_iterations) {
    if (threadIdx.x == 0 && blockIdx.x == 0) init_state(st, pose.m, thr, lambda, num_iterations);
}

// ---- stage 1 --------------------------------------------------------------------------------
template <int VEC, bool PINHOLE, int THREADS = NT, int ORDER = SCAN_FORWARD>
__global__ __launch_bounds__(THREADS) void stage1_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                    const double* __restrict__ x, const double* __restrict__ y,
                                                    const double* __restrict__ z, int64_t M,
                                                    const double* __restrict__ state, CameraK cam,
                                                    double* __restrict__ part) {
    __shared__ double lds[(THREADS / 64) * 64];
    CorrStream<VEC, THREADS, ORDER> stream;
    stream.start(u, v, x, y, z, M); // first loads in flight before the state block is read
    if (state[ST_STOP] != 0.0) return;
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = state[ST_C_T_W + i];
    const double thr = state[ST_THR];
    Stage1Sums s{0.0, 0u, 0u};
    stream.run([&](double mu, double mv, double px, double py, double pz) {
        stage1_point<PINHOLE>(T, cam, thr, mu, mv, px, py, pz, s);
    });
    const double vals[3] = {s.sum_chi, (double)s.n1, (double)s.n_inside};
    const double tot = block_sum<3, THREADS>(vals, lds);
    if (threadIdx.x < 3) part[(size_t)blockIdx.x * PICP_S1_DOUBLES + threadIdx.x] = tot;
    if (threadIdx.x == 3) part[(size_t)blockIdx.x * PICP_S1_DOUBLES + 3] = 0.0;
}

// ---- stage 2 --------------------------------------------------------------------------------
template <int VEC, bool PINHOLE, int REVERSE = SCAN_FORWARD> // REVERSE: the scan's ORDER (CorrStream)
__global__ __launch_bounds__(NT2) void stage2_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                    const double* __restrict__ x, const double* __restrict__ y,
                                                    const double* __restrict__ z, int64_t M,
                                                    const double* __restrict__ state, CameraK cam,
                                                    const double* __restrict__ s1part, int s1rows,
                                                    double* __restrict__ part) {
    __shared__ double lds[(NT2 / 64) * 64];
    __shared__ double s1[PICP_S1_DOUBLES];
    // first correspondences and the stage-1 records are requested together, before anything waits
    CorrStream<VEC, NT2, REVERSE> stream;
    stream.start(u, v, x, y, z, M);
    PartialFold<PICP_S1_DOUBLES, NT2, 8> f1;
    f1.issue(s1part, s1rows, PICP_S1_DOUBLES);
    if (state[ST_STOP] != 0.0) return;
    {
        const double tot = finish_fold(f1, s1part, s1rows, PICP_S1_DOUBLES, lds);
        if (threadIdx.x < PICP_S1_DOUBLES) s1[threadIdx.x] = tot;
        __syncthreads();
    }
    const double n1 = s1[S1_COUNT];
    double* row = part + (size_t)blockIdx.x * PICP_S2_DOUBLES;
    if (n1 == 0.0) { // nothing survived stage 1 (src/ProjectiveICP.py:300): the sums are zero
        if (threadIdx.x < PICP_S2_DOUBLES) row[threadIdx.x] = 0.0;
        return;
    }
    const double mean = s1[S1_SUM_CHI] / n1; // np.mean, :301
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = state[ST_C_T_W + i];
    const double thr = state[ST_THR];
    const bool all = (state[ST_MODE] != 0.0);
    Stage2Sums s;
    s.clear();
    stream.run([&](double mu, double mv, double px, double py, double pz) {
        stage2_point<PINHOLE>(T, cam, thr, mean, all, mu, mv, px, py, pz, s);
    });
    double vals[S2_USED];
    pack_sets(s, vals);
    const double tot = block_sum<S2_USED, NT2>(vals, lds);
    if (threadIdx.x < S2_USED) row[threadIdx.x] = tot;
    else if (threadIdx.x < PICP_S2_DOUBLES) row[threadIdx.x] = 0.0;
}

// Folds partial records into one record in caller memory (the all-reduce payload when sharded).
template <int NV>
__global__ __launch_bounds__(NT) void fold_kernel(const double* __restrict__ part, int rows,
                                                  const double* __restrict__ state, double* __restrict__ out) {
    __shared__ double lds[NT];
    if (state != nullptr && state[ST_STOP] != 0.0) return;
    const double tot = fold_partials<NV, NT>(part, rows, NV, lds);
    if (threadIdx.x < NV) out[threadIdx.x] = tot;
}

// ---- solve + schedules -------------------------------------------------------------------------
constexpr int NT_SOLVE = 512; // more lanes than the scans: the prologue folds up to max_blocks partial records
__global__ __launch_bounds__(NT_SOLVE) void solve_kernel(double* __restrict__ state, picp_schedule sc,
                                                   const double* __restrict__ s1part, int s1rows,
                                                   const double* __restrict__ s2part, int s2rows,
                                                   double* __restrict__ hist, int hist_rows) {
    __shared__ double lds[NT_SOLVE];
    __shared__ double s1[PICP_S1_DOUBLES];
    __shared__ double s2[PICP_S2_DOUBLES];
    // The tail (solve_step) runs against copies of the state block and the history row in LDS;
    // both are moved to and from global memory by the whole workgroup.  All global loads of
    // the prologue -- both sets of partial records and the state block -- are issued before the
    // first wait, so the kernel pays one memory round trip, not three.
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double row[PICP_HIST_STRIDE];
    PartialFold<PICP_S1_DOUBLES, NT_SOLVE, 4> f1;
    PartialFold<PICP_S2_DOUBLES, NT_SOLVE, 32> f2;
    f1.issue(s1part, s1rows, PICP_S1_DOUBLES);
    f2.issue(s2part, s2rows, PICP_S2_DOUBLES); // stage 2 always leaves valid records (zeros if stage 1 kept nothing)
    const double mine = (threadIdx.x < PICP_STATE_DOUBLES) ? state[threadIdx.x] : 0.0;
    if (threadIdx.x < PICP_STATE_DOUBLES) st[threadIdx.x] = mine;
    __syncthreads();
    if (st[ST_STOP] != 0.0) return;
    {
        const double t1 = finish_fold(f1, s1part, s1rows, PICP_S1_DOUBLES, lds);
        if (threadIdx.x < PICP_S1_DOUBLES) s1[threadIdx.x] = t1;
        const double t2 = finish_fold(f2, s2part, s2rows, PICP_S2_DOUBLES, lds);
        if (threadIdx.x < PICP_S2_DOUBLES) s2[threadIdx.x] = t2;
        __syncthreads();
    }
    __shared__ StepShared sh;
    const int it = (int)st[ST_IT];
    const bool record = (hist != nullptr && it < hist_rows);
    const bool advanced = solve_step(st, s1, s2, sc, record ? row : nullptr, &sh);
    if (threadIdx.x < PICP_STATE_DOUBLES) state[threadIdx.x] = st[threadIdx.x];
    if (record && advanced && threadIdx.x < PICP_HIST_STRIDE)
        hist[(size_t)it * PICP_HIST_STRIDE + threadIdx.x] = row[threadIdx.x];
}

// ---- sharded rounds: two launches per round ---------------------------------------------------
// A round of a sharded run is  stage 1 -> all-reduce -> stage 2 -> all-reduce  and nothing else:
//  * each scan folds its own per-workgroup records: a workgroup stores its record (write-through),
//    waits for the stores, and takes a ticket from an arrival counter; the workgroup that draws the
//    last ticket reads all records (in index order, so the sum does not depend on who was last) and
//    leaves the shard's record where the all-reduce expects it -- no fold launch;
//  * the solve step of round k runs in the prologue of round k+1's stage 1, in every workgroup alike
//    (identical inputs, identical result; the first loads of the scan are already in flight), and
//    workgroup 0 writes the new state block and the history row -- no solve launch.  The state
//    alternates between two blocks, so no workgroup reads a block that another one is rewriting.
// Only the last round of a call is followed by a solve launch, which leaves the caller's state
// block complete.
template <int NV, int NVW, int NT_>
__device__ __forceinline__ void fold_by_last_arrival(double tot, double* __restrict__ part, unsigned* __restrict__ counter,
                                                     double* __restrict__ out, double* lds /* NT_ */, int* lds_last) {
    static_assert(NVW <= 64 && NT_ % NVW == 0, "the record is stored by the first wave");
    const int G = gridDim.x;
    const int tid = threadIdx.x;
    if (tid < NVW)
        __hip_atomic_store(part + (size_t)blockIdx.x * NVW + tid, (tid < NV) ? tot : 0.0, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    if (tid < 64) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the record is in memory before the ticket is drawn
        if (tid == 0) {
            const unsigned ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *lds_last = (ticket == (unsigned)(G - 1)) ? 1 : 0;
        }
    }
    __syncthreads();
    if (*lds_last == 0) return;
    constexpr int GR = NT_ / NVW; // row groups
    const int col = tid % NVW, grp = tid / NVW;
    double acc = 0.0;
    for (int r = grp; r < G; r += GR)
        acc += __hip_atomic_load(part + (size_t)r * NVW + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds[tid] = acc;
    __syncthreads();
    if (tid < NVW) {
        double t = 0.0;
        for (int g = 0; g < GR; ++g) t += lds[g * NVW + tid];
        out[tid] = t;
    }
    if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // for the next launch
}

template <int VEC, bool PINHOLE, int THREADS, int ORDER = SCAN_FORWARD>
__global__ __launch_bounds__(THREADS) void stage1_round_kernel(
    const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ z, int64_t M, const double* __restrict__ state_in, double* __restrict__ state_out,
    picp_schedule sc, const double* __restrict__ s1_prev, const double* __restrict__ s2_prev, int do_solve,
    double* __restrict__ hist, int hist_rows, CameraK cam, double* __restrict__ part, unsigned* __restrict__ counter,
    double* __restrict__ s1_out) {
    static_assert(THREADS >= 192, "solve_step needs three waves");
    __shared__ double lds[THREADS];
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double s1[PICP_S1_DOUBLES];
    __shared__ double s2[PICP_S2_DOUBLES];
    __shared__ double row[PICP_HIST_STRIDE];
    __shared__ StepShared sh;
    __shared__ int last;
    CorrStream<VEC, THREADS, ORDER> stream;
    stream.start(u, v, x, y, z, M); // first loads in flight before anything else
    if (threadIdx.x < PICP_STATE_DOUBLES) st[threadIdx.x] = state_in[threadIdx.x];
    if (do_solve) {
        if (threadIdx.x < PICP_S1_DOUBLES) s1[threadIdx.x] = s1_prev[threadIdx.x];
        if (threadIdx.x >= 64 && threadIdx.x < 64 + PICP_S2_DOUBLES) s2[threadIdx.x - 64] = s2_prev[threadIdx.x - 64];
    }
    __syncthreads();
    if (do_solve) {
        if (st[ST_STOP] == 0.0) { // the previous round's solve step (uniform over the grid)
            const int it = (int)st[ST_IT];
            const bool record = (blockIdx.x == 0 && hist != nullptr && it < hist_rows);
            const bool advanced = solve_step(st, s1, s2, sc, record ? row : nullptr, &sh);
            if (record && advanced && threadIdx.x < PICP_HIST_STRIDE)
                hist[(size_t)it * PICP_HIST_STRIDE + threadIdx.x] = row[threadIdx.x];
        }
        if (blockIdx.x == 0 && threadIdx.x < PICP_STATE_DOUBLES) state_out[threadIdx.x] = st[threadIdx.x];
    }
    if (st[ST_STOP] != 0.0) return;
    double T[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = st[ST_C_T_W + i];
    const double thr = st[ST_THR];
    Stage1Sums s{0.0, 0u, 0u};
    stream.run([&](double mu, double mv, double px, double py, double pz) {
        stage1_point<PINHOLE>(T, cam, thr, mu, mv, px, py, pz, s);
    });
    const double vals[3] = {s.sum_chi, (double)s.n1, (double)s.n_inside};
    const double tot = block_sum<3, THREADS>(vals, lds);
    fold_by_last_arrival<3, PICP_S1_DOUBLES, THREADS>(tot, part, counter, s1_out, lds, &last);
}

template <int VEC, bool PINHOLE, int REVERSE> // REVERSE: the scan's ORDER (CorrStream)
__global__ __launch_bounds__(NT2) void stage2_round_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                          const double* __restrict__ x, const double* __restrict__ y,
                                                          const double* __restrict__ z, int64_t M,
                                                          const double* __restrict__ state, CameraK cam,
                                                          const double* __restrict__ s1g, double* __restrict__ part,
                                                          unsigned* __restrict__ counter, double* __restrict__ s2_out) {
    __shared__ double lds[NT2];
    __shared__ int last;
    CorrStream<VEC, NT2, REVERSE> stream;
    stream.start(u, v, x, y, z, M);
    if (state[ST_STOP] != 0.0) return;
    const double n1 = s1g[S1_COUNT];
    Stage2Sums s;
    s.clear();
    if (n1 != 0.0) { // else nothing survived stage 1 (src/ProjectiveICP.py:300): the sums are zero
        const double mean = s1g[S1_SUM_CHI] / n1; // np.mean, :301
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = state[ST_C_T_W + i];
        const double thr = state[ST_THR];
        const bool all = (state[ST_MODE] != 0.0);
        stream.run([&](double mu, double mv, double px, double py, double pz) {
            stage2_point<PINHOLE>(T, cam, thr, mean, all, mu, mv, px, py, pz, s);
        });
    }
    double vals[S2_USED];
    pack_sets(s, vals);
    const double tot = block_sum<S2_USED, NT2>(vals, lds);
    fold_by_last_arrival<S2_USED, PICP_S2_DOUBLES, NT2>(tot, part, counter, s2_out, lds, &last);
}

// The solve step as a launch of its own, from reduced records, reading one state block and writing
// another (or the same): closes the last round of a call.
__global__ __launch_bounds__(256) void solve_round_kernel(const double* __restrict__ state_in, double* __restrict__ state_out,
                                                         picp_schedule sc, const double* __restrict__ s1g,
                                                         const double* __restrict__ s2g, double* __restrict__ hist,
                                                         int hist_rows) {
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double s1[PICP_S1_DOUBLES];
    __shared__ double s2[PICP_S2_DOUBLES];
    __shared__ double row[PICP_HIST_STRIDE];
    __shared__ StepShared sh;
    if (threadIdx.x < PICP_STATE_DOUBLES) st[threadIdx.x] = state_in[threadIdx.x];
    if (threadIdx.x < PICP_S1_DOUBLES) s1[threadIdx.x] = s1g[threadIdx.x];
    if (threadIdx.x >= 128 && threadIdx.x < 128 + PICP_S2_DOUBLES) s2[threadIdx.x - 128] = s2g[threadIdx.x - 128];
    __syncthreads();
    if (st[ST_STOP] == 0.0) {
        const int it = (int)st[ST_IT];
        const bool record = (hist != nullptr && it < hist_rows);
        const bool advanced = solve_step(st, s1, s2, sc, record ? row : nullptr, &sh);
        if (record && advanced && threadIdx.x < PICP_HIST_STRIDE)
            hist[(size_t)it * PICP_HIST_STRIDE + threadIdx.x] = row[threadIdx.x];
    }
    if (threadIdx.x < PICP_STATE_DOUBLES) state_out[threadIdx.x] = st[threadIdx.x];
}

// ---- whole loop in one workgroup, one problem per workgroup -------------------------------------
// Problems of any size: the correspondences are re-read (from cache) every iteration.  Problems
// with a known bound on their size are held on chip instead (picp_resident.hip).
template <bool PINHOLE>
__global__ __launch_bounds__(NT) void icp_batched_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                         const double* __restrict__ x, const double* __restrict__ y,
                                                         const double* __restrict__ z,
                                                         const int64_t* __restrict__ offsets,
                                                         const double* __restrict__ poses0, CameraK cam,
                                                         picp_schedule sc, picp_result* __restrict__ results,
                                                         double* __restrict__ hist, int hist_rows) {
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double lds[(NT / 64) * 64];
    __shared__ double s1[PICP_S1_DOUBLES];
    __shared__ double s2[PICP_S2_DOUBLES];
    __shared__ StepShared sh;
    const int prob = blockIdx.x;
    const int64_t beg = offsets[prob];
    const int64_t cnt = offsets[prob + 1] - beg;
    if (threadIdx.x == 0) init_state(st, poses0 + (size_t)prob * 16, sc.kernel_threshold, sc.damping, sc.num_iterations);
    __syncthreads();
    double* myhist = (hist != nullptr) ? hist + (size_t)prob * hist_rows * PICP_HIST_STRIDE : nullptr;
    while (st[ST_STOP] == 0.0) {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = st[ST_C_T_W + i];
        const double thr = st[ST_THR];
        Stage1Sums a{0.0, 0u, 0u};
        for (int64_t i = threadIdx.x; i < cnt; i += NT) {
            double px = x[beg + i], py = y[beg + i], pz = z[beg + i];
            homogeneous_nan(px, py, pz);
            stage1_point<PINHOLE>(T, cam, thr, u[beg + i], v[beg + i], px, py, pz, a);
        }
        {
            const double vals[3] = {a.sum_chi, (double)a.n1, (double)a.n_inside};
            const double tot = block_sum<3, NT>(vals, lds);
            if (threadIdx.x < 3) s1[threadIdx.x] = tot;
            __syncthreads();
        }
        const double n1 = s1[S1_COUNT];
        // Stage 2 sums the mean-filtered set; if the solve step finds it too small (block-uniform,
        // rare) the scan is repeated over the whole stage-1 set for the same pose.
        for (int pass = 0; pass < 2; ++pass) {
            if (n1 != 0.0) {
                const double mean = s1[S1_SUM_CHI] / n1;
                const bool all = (st[ST_MODE] != 0.0);
                Stage2Sums s;
                s.clear();
                for (int64_t i = threadIdx.x; i < cnt; i += NT) {
                    double px = x[beg + i], py = y[beg + i], pz = z[beg + i];
                    homogeneous_nan(px, py, pz);
                    stage2_point<PINHOLE>(T, cam, thr, mean, all, u[beg + i], v[beg + i], px, py, pz, s);
                }
                double vals[S2_USED];
                pack_sets(s, vals);
                const double tot = block_sum<S2_USED, NT>(vals, lds);
                if (threadIdx.x < S2_USED) s2[threadIdx.x] = tot;
                __syncthreads();
            }
            {
                const int it = (int)st[ST_IT];
                double* row = (myhist != nullptr && it < hist_rows) ? myhist + (size_t)it * PICP_HIST_STRIDE : nullptr;
                (void)solve_step(st, s1, s2, sc, row, &sh);
            }
            if (st[ST_MODE] == 0.0) break;
        }
    }
    if (threadIdx.x == 0) {
        picp_result r;
        const bool valid = (st[ST_IT] > 0.0) && !(st[ST_BEST_ERR] > 5.0); // src/ProjectiveICP.py:249
        for (int i = 0; i < 16; ++i) r.T[i] = valid ? st[ST_BEST_T + i] : st[ST_START_T + i];
        r.valid = valid ? 1 : 0;
        r.num_iterations = (int)st[ST_IT];
        r.best_index = (int)st[ST_BEST_IT];
        r.reserved = 0;
        r.best_error = st[ST_BEST_ERR];
        results[prob] = r;
    }
}

__global__ void invert_kernel(const double* in, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) invert_pose(in, out);
}
__global__ void v2T_kernel(const double* in, double* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) v2T_device(in, out);
}

// ---- host helpers ------------------------------------------------------------------------------
static CameraK to_device_camera(const picp_camera* cam) {
    CameraK c;
    for (int i = 0; i < 9; ++i) c.k[i] = cam->K[i];
    c.width = cam->width;
    c.height = cam->height;
    return c;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static int grid_for(int64_t items, int threads, int max_blocks) {
    const int64_t want = (items + threads - 1) / threads;
    int64_t g = want < 1 ? 1 : want;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

static int check_corr(const double* u, const double* v, const double* x, const double* y, const double* z, int64_t M) {
    PICP_REQUIRE(M >= 0, "negative correspondence count %lld", (long long)M);
    PICP_REQUIRE(M == 0 || (u && v && x && y && z), "null correspondence array with M = %lld", (long long)M);
    return PICP_OK;
}

// The best-iteration rule on a host copy of the state block (src/ProjectiveICP.py:249-257):
// invalid when the smallest error is > 5 (NaN-free by construction); then the start pose is returned.
static void result_from_state(const double* h, const double* w_T_c0_host, picp_result* result) {
    const int n_it = (int)h[ST_IT];
    const bool valid = n_it > 0 && !(h[ST_BEST_ERR] > 5.0);
    for (int i = 0; i < 16; ++i)
        result->T[i] = valid ? h[ST_BEST_T + i] : (w_T_c0_host ? w_T_c0_host[i] : h[ST_START_T + i]);
    result->valid = valid ? 1 : 0;
    result->num_iterations = n_it;
    result->best_index = (int)h[ST_BEST_IT];
    result->reserved = 0;
    result->best_error = h[ST_BEST_ERR];
}

// Waits for a single-launch loop that has just been enqueued with ctx->loop_done as its completion
// event (bound to the dispatch: an event recorded behind it would be a packet of its own on the stream).
static int wait_for_loop(picp_ctx* ctx) {
    PICP_HIP_CHECK(hipEventSynchronize(ctx->loop_done)); // (polling hipEventQuery instead measured slower)
    return PICP_OK;
}

static int check_pose(const double* T) {
    PICP_REQUIRE(T != nullptr, "null pose");
    PICP_REQUIRE(T[12] == 0.0 && T[13] == 0.0 && T[14] == 0.0 && T[15] == 1.0,
                 "pose last row must be (0,0,0,1), got (%g,%g,%g,%g)", T[12], T[13], T[14], T[15]);
    return PICP_OK;
}

// Launches the stage-1 scan with its per-workgroup records going to `part` (room for `max_rows`
// records); *rows receives the number of records written.
static int launch_stage1(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                         const double* z, int64_t M, const double* state, const picp_camera* cam, double* part,
                         int max_rows, int* rows) {
    const CameraK c = to_device_camera(cam);
    const bool vec = aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z);
    // above ~4M correspondences one workgroup per CU streams as fast and leaves half the records
    const int cap1 = (M > (int64_t(1) << 22) && max_rows >= 2 * ctx->num_cus) ? ctx->num_cus : max_rows;
    const int grid = grid_for(vec ? (M + 1) / 2 : M, NT, cap1);
    auto kern = vec ? (is_pinhole(cam->K) ? stage1_kernel<2, true> : stage1_kernel<2, false>)
                    : (is_pinhole(cam->K) ? stage1_kernel<1, true> : stage1_kernel<1, false>);
    // Where stage 2 runs against stage 1's direction and the arrays are not much larger than the
    // memory-side cache, stage 1 takes stage 2's workgroup shape and grid: every workgroup then scans
    // the same share of the arrays in both stages (and sits on the same XCD), so stage 2 also finds
    // the end of its share in that XCD's L2 (per iteration: 2^21 correspondences 40.3 -> 36.6 us,
    // 2^22 62 -> 59, 2^23 111.8 -> 109.4; nothing from 12 M on).
    if (vec && stage2_reverse(ctx, M) && M * 5 * (int64_t)sizeof(double) <= (int64_t(384) << 20)) {
        const int wide_cap = ctx->max_blocks2 < max_rows ? ctx->max_blocks2 : max_rows; // `part` holds max_rows records
        const int gridw = grid_for((M + 1) / 2, NT2, wide_cap);
        *rows = gridw;
        auto kw = is_pinhole(cam->K) ? stage1_kernel<2, true, NT2> : stage1_kernel<2, false, NT2>;
        hipLaunchKernelGGL(kw, dim3(gridw), dim3(NT2), 0, ctx->stream, u, v, x, y, z, M, state, c, part);
        PICP_HIP_CHECK(hipGetLastError());
        return PICP_OK;
    }
    if (vec && scan_nontemporal(ctx, M)) // arrays far beyond the memory-side cache: non-temporal loads
        kern = is_pinhole(cam->K) ? stage1_kernel<2, true, NT, SCAN_FORWARD_NONTEMPORAL>
                                  : stage1_kernel<2, false, NT, SCAN_FORWARD_NONTEMPORAL>;
    *rows = grid;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), 0, ctx->stream, u, v, x, y, z, M, state, c, part);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

// Launches the stage-2 scan: folds the `s1rows` stage-1 records at `s1part`, leaves its own
// per-workgroup records in `part` (room for `max_rows`); *rows receives their number.
static int launch_stage2(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                         const double* z, int64_t M, const double* state, const picp_camera* cam, const double* s1part,
                         int s1rows, double* part, int max_rows, int* rows) {
    const CameraK c = to_device_camera(cam);
    const bool vec = aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z);
    const int grid = grid_for(vec ? (M + 1) / 2 : M, NT2, max_rows);
    *rows = grid;
    auto kern = vec ? (is_pinhole(cam->K) ? stage2_kernel<2, true> : stage2_kernel<2, false>)
                    : (is_pinhole(cam->K) ? stage2_kernel<1, true> : stage2_kernel<1, false>);
    const bool reverse = stage2_reverse(ctx, M);
    if (vec && reverse) kern = is_pinhole(cam->K) ? stage2_kernel<2, true, SCAN_REVERSE> : stage2_kernel<2, false, SCAN_REVERSE>;
    else if (vec && scan_nontemporal(ctx, M))
        kern = is_pinhole(cam->K) ? stage2_kernel<2, true, SCAN_FORWARD_NONTEMPORAL> : stage2_kernel<2, false, SCAN_FORWARD_NONTEMPORAL>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT2), 0, ctx->stream, u, v, x, y, z, M, state, c, s1part, s1rows, part);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

// ---- picp_linearize for callers that cannot allocate ------------------------------------------
// Everything picp_linearize keeps in the ctx's hipMalloc'd workspace (the loop state block, the
// per-workgroup records of both scans and the two folded records) lives in one file-scope device
// array instead, so the entry point needs no allocation of any kind.  Every GPU the library runs
// on holds its own copy of the array until the process ends; its address on a device is looked up
// once, when a ctx is created on that device (noalloc_bind), and that address -- never the
// array's name, which in host code is only a handle for the lookup -- is what kernels receive.
// Per-workgroup records of one scan that fit the array: the largest grids picp_create can configure on a
// 256-CU device (8 workgroups per CU for stage 1, 4 for stage 2), so that the grids -- and with them the
// order of additions -- are picp_linearize's whatever the tuning.
constexpr int NOALLOC_MAX_WG1 = 2048;
constexpr int NOALLOC_MAX_WG2 = 1024;
enum : int {
    NA_STATE = 0,                                      // PICP_STATE_DOUBLES
    NA_S1 = 128,                                       // folded stage-1 record
    NA_S2 = 160,                                       // folded stage-2 record
    NA_PART1 = 256,                                    // [NOALLOC_MAX_WG1][PICP_S1_DOUBLES]
    NA_PART2 = NA_PART1 + NOALLOC_MAX_WG1 * PICP_S1_DOUBLES, // [NOALLOC_MAX_WG2][PICP_S2_DOUBLES]
    NA_DOUBLES = NA_PART2 + NOALLOC_MAX_WG2 * PICP_S2_DOUBLES, // 41 216 doubles = 322 KiB per GPU
};
static __device__ __align__(256) double scratch[NA_DOUBLES];

int noalloc_bind(picp_ctx* ctx) { // called by picp_create with the ctx's device current
    double* p = nullptr;
    PICP_HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
    ctx->noalloc = p;
    return PICP_OK;
}

// The set choice of __linearize (src/ProjectiveICP.py:300-310, :319-321) on the folded records, on
// the device: `out` receives H[36], b[6], num_inliers, error, n_inside (PICP_LINEARIZE_OUT_DOUBLES).
// When the record holds the mean-filtered set and that set is too small (:307), nothing is written:
// the state's mode flag is set and the caller's second stage-2 launch re-sums the stage-1 set;
// otherwise the state's stop flag is set, which turns that second round into no-ops.
__global__ void linearize_decide_kernel(double* __restrict__ state, const double* __restrict__ s1,
                                        const double* __restrict__ s2, int min_inliers, double* __restrict__ out) {
    if (state[ST_STOP] != 0.0) return;
    const int tid = threadIdx.x;
    const bool none = (s1[S1_COUNT] == 0.0); // :300
    const bool redo_mode = (state[ST_MODE] != 0.0);
    if (!none && !redo_mode && s2[S2_N] < (double)min_inliers) {
        if (tid == 0) state[ST_MODE] = 1.0;
        return;
    }
    if (tid < 36) {
        const int i = tid / 6, j = tid % 6;
        const int lo = i < j ? i : j, hi = i < j ? j : i;
        const int tri = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo);
        out[tid] = none ? 0.0 : s2[S2_H + tri];
    } else if (tid < 42) {
        out[tid] = none ? 0.0 : s2[S2_B + (tid - 36)];
    } else if (tid == 42) {
        out[42] = none ? 0.0 : s2[S2_N];
    } else if (tid == 43) {
        out[43] = (!none && s2[S2_N] > 0.0) ? s2[S2_CHI] / s2[S2_N] : d_inf();
    } else if (tid == 44) {
        out[44] = s1[S1_N_INSIDE];
    } else if (tid < PICP_LINEARIZE_OUT_DOUBLES) {
        out[tid] = 0.0;
    }
    if (tid == 63) state[ST_STOP] = 1.0;
}

// One call's worth of sharded rounds with two launches per round (see stage1_round_kernel).
// allreduce(buf, count): the caller's exchange, enqueued on the ctx's stream (a no-op for one shard).
int sharded_rounds(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                   int64_t M, double* state, const picp_camera* cam, const picp_schedule* sched, double* s1, double* s2,
                   double* hist, int hist_rows, int rounds, int (*allreduce)(picp_ctx*, double*, size_t)) {
    if (rounds <= 0) return PICP_OK;
    const CameraK c = to_device_camera(cam);
    const bool pin = is_pinhole(cam->K);
    const bool vec = aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z);
    const bool reverse = vec && stage2_reverse(ctx, M);
    const int grid1 = grid_for(vec ? (M + 1) / 2 : M, NT2, ctx->max_blocks2 < ctx->max_blocks1 ? ctx->max_blocks2 : ctx->max_blocks1);
    const int grid2 = grid_for(vec ? (M + 1) / 2 : M, NT2, ctx->max_blocks2);
    auto k1 = vec ? (pin ? stage1_round_kernel<2, true, NT2> : stage1_round_kernel<2, false, NT2>)
                  : (pin ? stage1_round_kernel<1, true, NT2> : stage1_round_kernel<1, false, NT2>);
    auto k2 = vec ? (reverse ? (pin ? stage2_round_kernel<2, true, SCAN_REVERSE> : stage2_round_kernel<2, false, SCAN_REVERSE>)
                             : (pin ? stage2_round_kernel<2, true, SCAN_FORWARD> : stage2_round_kernel<2, false, SCAN_FORWARD>))
                  : (pin ? stage2_round_kernel<1, true, SCAN_FORWARD> : stage2_round_kernel<1, false, SCAN_FORWARD>);
    if (vec && !reverse && scan_nontemporal(ctx, M)) { // arrays far beyond the memory-side cache
        k1 = pin ? stage1_round_kernel<2, true, NT2, SCAN_FORWARD_NONTEMPORAL> : stage1_round_kernel<2, false, NT2, SCAN_FORWARD_NONTEMPORAL>;
        k2 = pin ? stage2_round_kernel<2, true, SCAN_FORWARD_NONTEMPORAL> : stage2_round_kernel<2, false, SCAN_FORWARD_NONTEMPORAL>;
    }
    double* other = ctx->scratch + SCR_STATE2; // the state alternates between the caller's block and this one
    double* cur = state;
    hipStream_t st = ctx->stream;
    for (int k = 0; k < rounds; ++k) {
        double* nxt = (cur == state) ? other : state;
        hipLaunchKernelGGL(k1, dim3(grid1), dim3(NT2), 0, st, u, v, x, y, z, M, (const double*)cur, nxt, *sched,
                           (const double*)s1, (const double*)s2, k > 0 ? 1 : 0, hist, hist_rows, c, ctx->part1,
                           ctx->arrive, s1);
        if (k > 0) cur = nxt;
        if (int rc = allreduce(ctx, s1, PICP_S1_DOUBLES)) return rc;
        hipLaunchKernelGGL(k2, dim3(grid2), dim3(NT2), 0, st, u, v, x, y, z, M, (const double*)cur, c, (const double*)s1,
                           ctx->part2, ctx->arrive + 1, s2);
        if (int rc = allreduce(ctx, s2, PICP_S2_DOUBLES)) return rc;
    }
    hipLaunchKernelGGL(solve_round_kernel, dim3(1), dim3(256), 0, st, (const double*)cur, state, *sched, (const double*)s1,
                       (const double*)s2, hist, hist_rows);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

} // namespace picp

using namespace picp;

extern "C" {

int picp_state_init(picp_ctx* ctx, double* state, const double* w_T_c0_host, const picp_schedule* sched) {
    PICP_REQUIRE(ctx && state && sched, "picp_state_init: null argument");
    if (int rc = check_pose(w_T_c0_host)) return rc;
    DeviceGuard g(ctx->device);
    Pose16 p;
    memcpy(p.m, w_T_c0_host, sizeof(p.m));
    hipLaunchKernelGGL(state_init_kernel, dim3(1), dim3(64), 0, ctx->stream, state, p, sched->kernel_threshold,
                       sched->damping, sched->num_iterations);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_stage1(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                int64_t M, const double* state, const picp_camera* cam) {
    PICP_REQUIRE(ctx && state && cam, "picp_stage1: null argument");
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    DeviceGuard g(ctx->device);
    return launch_stage1(ctx, u, v, x, y, z, M, state, cam, ctx->part1, ctx->max_blocks1, &ctx->rows1);
}

int picp_stage1_reduce(picp_ctx* ctx, double* s1) {
    PICP_REQUIRE(ctx && s1, "picp_stage1_reduce: null argument");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(fold_kernel<PICP_S1_DOUBLES>, dim3(1), dim3(NT), 0, ctx->stream, ctx->part1, ctx->rows1,
                       (const double*)nullptr, s1);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_stage2(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                int64_t M, const double* state, const picp_camera* cam, const double* s1) {
    PICP_REQUIRE(ctx && state && cam, "picp_stage2: null argument");
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    DeviceGuard g(ctx->device);
    return launch_stage2(ctx, u, v, x, y, z, M, state, cam, s1 ? s1 : ctx->part1, s1 ? 1 : ctx->rows1, ctx->part2,
                         ctx->max_blocks2, &ctx->rows2);
}

int picp_profile_stage2(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                        const double* z, int64_t M, const double* state, const picp_camera* cam, int32_t reps,
                        double* mean_ms, double* min_ms) {
    PICP_REQUIRE(ctx && state && cam && mean_ms && min_ms && reps > 0, "picp_profile_stage2: bad argument");
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    DeviceGuard g(ctx->device);
    const CameraK c = to_device_camera(cam);
    const bool vec = aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z);
    const int grid = grid_for(vec ? (M + 1) / 2 : M, NT2, ctx->max_blocks2);
    ctx->rows2 = grid;
    auto kern = vec ? (is_pinhole(cam->K) ? stage2_kernel<2, true> : stage2_kernel<2, false>)
                    : (is_pinhole(cam->K) ? stage2_kernel<1, true> : stage2_kernel<1, false>);
    if (vec && scan_nontemporal(ctx, M))
        kern = is_pinhole(cam->K) ? stage2_kernel<2, true, SCAN_FORWARD_NONTEMPORAL> : stage2_kernel<2, false, SCAN_FORWARD_NONTEMPORAL>;
    hipEvent_t e0, e1;
    PICP_HIP_CHECK(hipEventCreate(&e0));
    PICP_HIP_CHECK(hipEventCreate(&e1));
    double sum = 0.0, best = 1e30;
    for (int r = 0; r < reps; ++r) {
        // start/stop events bound to this one dispatch: they carry the kernel's own begin and end
        // timestamps, the same interval a kernel trace reports
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NT2), 0, ctx->stream, e0, e1, 0, u, v, x, y, z, M, state, c,
                              (const double*)ctx->part1, ctx->rows1, ctx->part2);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return hip_fail(e, "picp_profile_stage2");
        }
        sum += ms;
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *mean_ms = sum / reps;
    *min_ms = best;
    return PICP_OK;
}

int picp_profile_run(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                     int64_t M, const double* w_T_c0_host, const picp_camera* cam, const picp_schedule* sched,
                     double* state, int32_t reps, double* mean_ms, double* min_ms) {
    PICP_REQUIRE(ctx && cam && sched && state && mean_ms && min_ms && reps > 0, "picp_profile_run: bad argument");
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    if (int rc = check_pose(w_T_c0_host)) return rc;
    const bool on_chip = M > 0 && M <= resident_capacity(ctx);
    PICP_REQUIRE(on_chip || (M > 0 && ctx->resident_limit > 0 && aligned16(u) && aligned16(v) && aligned16(x) &&
                             aligned16(y) && aligned16(z)),
                 "picp_profile_run: %lld correspondences do not run as a single launch (on-chip capacity %lld; larger "
                 "inputs need 16-byte aligned arrays)", (long long)M, (long long)resident_capacity(ctx));
    DeviceGuard g(ctx->device);
    hipEvent_t e0, e1;
    PICP_HIP_CHECK(hipEventCreate(&e0));
    PICP_HIP_CHECK(hipEventCreate(&e1));
    double sum = 0.0, best = 1e30;
    int rc = PICP_OK;
    for (int r = 0; r < reps && rc == PICP_OK; ++r) {
        rc = on_chip ? enqueue_resident(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, nullptr, 0, nullptr, e0, e1)
                     : enqueue_persistent(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, nullptr, 0, nullptr, e0, e1);
        if (rc != PICP_OK) break;
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            rc = hip_fail(e, "picp_profile_run");
            break;
        }
        sum += ms;
        if (ms < best) best = ms;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != PICP_OK) return rc;
    *mean_ms = sum / reps;
    *min_ms = best;
    return PICP_OK;
}

int picp_stage2_reduce(picp_ctx* ctx, double* s2) {
    PICP_REQUIRE(ctx && s2, "picp_stage2_reduce: null argument");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(fold_kernel<PICP_S2_DOUBLES>, dim3(1), dim3(NT), 0, ctx->stream, ctx->part2, ctx->rows2,
                       (const double*)nullptr, s2);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_solve(picp_ctx* ctx, double* state, const picp_schedule* sched, const double* s1, const double* s2,
               double* hist, int32_t hist_rows) {
    PICP_REQUIRE(ctx && state && sched, "picp_solve: null argument");
    PICP_REQUIRE(hist == nullptr || hist_rows >= 0, "picp_solve: negative hist_rows");
    DeviceGuard g(ctx->device);
    const double* s1p = s1 ? s1 : ctx->part1;
    const int s1rows = s1 ? 1 : ctx->rows1;
    const double* s2p = s2 ? s2 : ctx->part2;
    const int s2rows = s2 ? 1 : ctx->rows2;
    hipLaunchKernelGGL(solve_kernel, dim3(1), dim3(NT_SOLVE), 0, ctx->stream, state, *sched, s1p, s1rows, s2p, s2rows, hist,
                       (int)hist_rows);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_finish(picp_ctx* ctx, const double* state, const double* hist, int32_t hist_rows,
                const double* w_T_c0_host, picp_result* result) {
    (void)hist;
    (void)hist_rows;
    PICP_REQUIRE(ctx && state && result, "picp_finish: null argument");
    DeviceGuard g(ctx->device);
    double* h = ctx->pinned;
    PICP_HIP_CHECK(hipMemcpyAsync(h, state, PICP_STATE_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    result_from_state(h, w_T_c0_host, result);
    return PICP_OK;
}

// After an on-chip loop in its lean form left because of an immoderate value: the next
// PICP_FULL_FORM_CALLS on-chip loops of this context use the other form (such inputs come in runs,
// and a lean launch that leaves costs a few microseconds).
static void immoderate_exit(picp_ctx* ctx) {
    ctx->immoderate_exits += 1;
    ctx->full_select_calls = PICP_FULL_FORM_CALLS;
}

int picp_run(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
             int64_t M, const double* w_T_c0_host, const picp_camera* cam, const picp_schedule* sched, double* state,
             double* hist, int32_t hist_rows, picp_result* result) {
    PICP_REQUIRE(ctx && cam && sched && state && result, "picp_run: null argument");
    PICP_REQUIRE(sched->num_iterations >= 0, "picp_run: negative num_iterations");
    PICP_REQUIRE(hist == nullptr || hist_rows >= sched->num_iterations,
                 "picp_run: history has %d rows, num_iterations is %d", hist_rows, sched->num_iterations);
    if (int rc = check_pose(w_T_c0_host)) return rc;
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    DeviceGuard g(ctx->device);
    // A single-launch loop that gives up (device shared with other work) has burnt its whole bounded
    // wait, a few hundred milliseconds, before the stage launches take over.  After three give-ups in
    // a row the single launches are left out for a number of calls (32, doubling with every further
    // give-up, at most 1 024), then tried again; one completed loop clears the streak.
    const bool single_launch = (ctx->single_launch_holdoff == 0);
    if (!single_launch) ctx->single_launch_holdoff -= 1;
    auto gave_up = [&]() {
        ctx->resident_aborts += 1;
        ctx->abort_streak += 1;
        if (ctx->abort_streak >= 3) {
            const int shift = ctx->abort_streak - 3 < 5 ? ctx->abort_streak - 3 : 5;
            ctx->single_launch_holdoff = 32 << shift;
        }
    };
    // Inputs that fit the register files run the whole loop in one launch (picp_resident.hip).
    if (single_launch && M > 0 && M <= resident_capacity(ctx) && sched->num_iterations > 0) {
        if (int rc = enqueue_resident(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist, (int)hist_rows,
                                      ctx->pinned, nullptr, ctx->loop_done))
            return rc;
        if (int rc = wait_for_loop(ctx)) return rc; // the kernel left the final state in ctx->pinned
        if (ctx->pinned[ST_STOP] == STOP_IMMODERATE) {
            // The lean form met a value that is not finite and moderate and left before computing anything:
            // run the form without that assumption, now and for the next calls (such inputs come in runs).
            immoderate_exit(ctx);
            if (int rc = enqueue_resident(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist, (int)hist_rows,
                                          ctx->pinned, nullptr, ctx->loop_done))
                return rc;
            if (int rc = wait_for_loop(ctx)) return rc;
        }
        if (ctx->pinned[ST_STOP] != 0.0) {
            ctx->abort_streak = 0;
            result_from_state(ctx->pinned, w_T_c0_host, result);
            return PICP_OK;
        }
        // the launch was abandoned at a grid-wide wait (device shared with other work): start over
        gave_up();
    }
    // Larger inputs: the same single launch with the arrays streamed from memory every iteration.
    if (single_launch && M > resident_capacity(ctx) && ctx->resident_limit > 0 && M <= ctx->persistent_limit &&
        sched->num_iterations > 0 && aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z)) {
        if (int rc = enqueue_persistent(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist, (int)hist_rows,
                                        ctx->pinned, nullptr, ctx->loop_done))
            return rc;
        if (int rc = wait_for_loop(ctx)) return rc; // the kernel left the final state in ctx->pinned
        if (ctx->pinned[ST_STOP] != 0.0) {
            ctx->abort_streak = 0;
            result_from_state(ctx->pinned, w_T_c0_host, result);
            return PICP_OK;
        }
        gave_up();
    }
    if (int rc = picp_state_init(ctx, state, w_T_c0_host, sched)) return rc;
    // Rounds of {stage 1, stage 2, solve} are enqueued in chunks; between chunks the stop flag is
    // read back, so a loop that converged after a few iterations does not pay for launches that
    // would do nothing (every kernel returns at once when the flag is set, so overshooting is
    // safe).  A round normally completes one iteration; in the rare case that the mean-filtered
    // set is too small it only switches stage 2 to the stage-1 set and the next round completes
    // the iteration, so at most 2 * num_iterations rounds are ever needed.  The stop flag is set
    // by the solve step when the stop test fires or num_iterations is reached.
    const int total = sched->num_iterations;
    int chunk = sched->disable_stop ? total : 8;
    int issued = 0;
    while (total > 0 && issued < 2 * total) {
        const int n = chunk < 1 ? 1 : chunk;
        for (int k = 0; k < n; ++k) {
            if (int rc = picp_stage1(ctx, u, v, x, y, z, M, state, cam)) return rc;
            if (int rc = picp_stage2(ctx, u, v, x, y, z, M, state, cam, nullptr)) return rc;
            if (int rc = picp_solve(ctx, state, sched, nullptr, nullptr, hist, hist_rows)) return rc;
        }
        issued += n;
        double* h = ctx->pinned;
        PICP_HIP_CHECK(hipMemcpyAsync(h, state + ST_IT, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (h[1] != 0.0) break; // ST_STOP follows ST_IT
        const int left = total - (int)h[0];
        chunk = sched->disable_stop ? left : (chunk < 64 ? chunk * 2 : chunk);
        if (chunk > left) chunk = left;
    }
    return picp_finish(ctx, state, hist, hist_rows, w_T_c0_host, result);
}

int picp_run_sharded(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                     int64_t M, const double* w_T_c0_host, const picp_camera* cam, const picp_schedule* sched,
                     double* state, double* hist, int32_t hist_rows, picp_result* result) {
    PICP_REQUIRE(ctx && cam && sched && state && result, "picp_run_sharded: null argument");
    PICP_REQUIRE(sched->num_iterations > 0, "picp_run_sharded: num_iterations must be positive");
    PICP_REQUIRE(hist == nullptr || hist_rows >= sched->num_iterations,
                 "picp_run_sharded: history has %d rows, num_iterations is %d", hist_rows, sched->num_iterations);
    if (int rc = check_pose(w_T_c0_host)) return rc;
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    PICP_REQUIRE(ctx->peer_connected, "picp_run_sharded: no exchange between the GPUs (picp_peer_create / picp_peer_connect)");
    if (ctx->peer_broken) {
        set_error("picp_run_sharded: an earlier launch gave up at an exchange; destroy the exchange and set it up again");
        return PICP_ERR_RETRY;
    }
    PICP_REQUIRE(M == 0 || (aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z)),
                 "picp_run_sharded: the slices must be 16-byte aligned");
    DeviceGuard g(ctx->device);
    if (ctx->time_launches && ctx->prof0 == nullptr) {
        PICP_HIP_CHECK(hipEventCreate(&ctx->prof0));
        PICP_HIP_CHECK(hipEventCreate(&ctx->prof1));
    }
    hipEvent_t e0 = ctx->time_launches ? ctx->prof0 : nullptr, e1 = ctx->time_launches ? ctx->prof1 : ctx->loop_done;
    // A slice that fits the register files stays on chip for the whole loop, a larger (or empty) one is streamed:
    // both kernels run the same exchanges between the GPUs, so every rank chooses for its own slice.
    const bool on_chip = M > 0 && M <= resident_capacity(ctx);
    if (int rc = on_chip ? enqueue_resident(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist, (int)hist_rows,
                                            ctx->pinned, e0, e1, true)
                         : enqueue_persistent(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist,
                                              (int)hist_rows, ctx->pinned, e0, e1, true))
        return rc;
    PICP_HIP_CHECK(hipEventSynchronize(e1)); // the kernel left the final state in ctx->pinned
    if (e0 != nullptr) {
        float ms = 0.f;
        PICP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        ctx->last_launch_ms = ms;
    }
    if (ctx->pinned[ST_STOP] == 0.0) {
        ctx->peer_broken = 1;
        ctx->resident_aborts += 1;
        set_error("picp_run_sharded: the launch gave up waiting for a record (of a workgroup or of another rank)");
        return PICP_ERR_RETRY;
    }
    ctx->peer_epoch = (unsigned)ctx->pinned[ST_XEPOCH];
    result_from_state(ctx->pinned, w_T_c0_host, result);
    return PICP_OK;
}

int picp_set_launch_timing(picp_ctx* ctx, int32_t on) {
    PICP_REQUIRE(ctx != nullptr, "picp_set_launch_timing: null ctx");
    ctx->time_launches = on ? 1 : 0;
    return PICP_OK;
}

int picp_last_launch_ms(picp_ctx* ctx, double* ms) {
    PICP_REQUIRE(ctx != nullptr && ms != nullptr, "picp_last_launch_ms: null argument");
    *ms = ctx->last_launch_ms;
    return PICP_OK;
}

int picp_run_enqueue(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y, const double* z,
                     int64_t M, const double* w_T_c0_host, const picp_camera* cam, const picp_schedule* sched,
                     double* state, double* hist, int32_t hist_rows, int32_t slot) {
    PICP_REQUIRE(ctx && cam && sched && state, "picp_run_enqueue: null argument");
    PICP_REQUIRE(slot >= 0 && slot < PICP_ASYNC_SLOTS, "picp_run_enqueue: slot %d is not in 0..%d", (int)slot,
                 PICP_ASYNC_SLOTS - 1);
    PICP_REQUIRE(sched->num_iterations > 0, "picp_run_enqueue: num_iterations must be positive");
    PICP_REQUIRE(hist == nullptr || hist_rows >= sched->num_iterations,
                 "picp_run_enqueue: history has %d rows, num_iterations is %d", hist_rows, sched->num_iterations);
    if (int rc = check_pose(w_T_c0_host)) return rc;
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    const bool on_chip = M > 0 && M <= resident_capacity(ctx);
    const bool streamed = M > resident_capacity(ctx) && ctx->resident_limit > 0 && M <= ctx->persistent_limit &&
                          aligned16(u) && aligned16(v) && aligned16(x) && aligned16(y) && aligned16(z);
    PICP_REQUIRE(on_chip || streamed, "picp_run_enqueue: %lld correspondences do not run as a single launch here "
                 "(empty input, unaligned arrays, or single launches switched off): use picp_run", (long long)M);
    DeviceGuard g(ctx->device);
    double* host_state = ctx->pinned + PINNED_SLOT0 + (size_t)slot * PINNED_SLOT_STRIDE;
    const int rc = on_chip ? enqueue_resident(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist, (int)hist_rows,
                                              host_state, nullptr, ctx->slot_done[slot])
                           : enqueue_persistent(ctx, u, v, x, y, z, M, w_T_c0_host, state, cam, sched, hist,
                                                (int)hist_rows, host_state, nullptr, ctx->slot_done[slot]);
    return rc; // (the slot's event is the dispatch's completion event: no packet of its own behind the loop)
}

int picp_run_collect(picp_ctx* ctx, int32_t slot, const double* w_T_c0_host, picp_result* result) {
    PICP_REQUIRE(ctx && result, "picp_run_collect: null argument");
    PICP_REQUIRE(slot >= 0 && slot < PICP_ASYNC_SLOTS, "picp_run_collect: slot %d is not in 0..%d", (int)slot,
                 PICP_ASYNC_SLOTS - 1);
    DeviceGuard g(ctx->device);
    PICP_HIP_CHECK(hipEventSynchronize(ctx->slot_done[slot]));
    const double* h = ctx->pinned + PINNED_SLOT0 + (size_t)slot * PINNED_SLOT_STRIDE;
    if (h[ST_STOP] == STOP_IMMODERATE) { // nothing computed; the next launches use the form without assumptions
        immoderate_exit(ctx);
        set_error("picp_run_collect: the launch of slot %d met values that are not finite and of moderate magnitude and "
                  "left; run it again (picp_run or picp_run_enqueue)", (int)slot);
        return PICP_ERR_RETRY;
    }
    if (h[ST_STOP] == 0.0) { // the launch was abandoned at a grid-wide wait
        ctx->resident_aborts += 1;
        set_error("picp_run_collect: the launch of slot %d gave up at an exchange (device shared with other work)", (int)slot);
        return PICP_ERR_RETRY;
    }
    result_from_state(h, w_T_c0_host, result);
    return PICP_OK;
}

int picp_run_batched(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, const int64_t* offsets, int32_t B, const double* poses0,
                     const picp_camera* cam, const picp_schedule* sched, picp_result* results, double* hist,
                     int32_t hist_rows, int64_t max_count) {
    PICP_REQUIRE(ctx && offsets && poses0 && cam && sched && results, "picp_run_batched: null argument");
    PICP_REQUIRE(B >= 0 && max_count >= 0, "picp_run_batched: negative batch or max_count");
    PICP_REQUIRE(hist == nullptr || hist_rows >= sched->num_iterations,
                 "picp_run_batched: history has %d rows, num_iterations is %d", hist_rows, sched->num_iterations);
    if (B == 0) return PICP_OK;
    DeviceGuard g(ctx->device);
    // With a bound on the problem size that fits, the problems are held on chip (picp_resident.hip).
    if (max_count > 0 && max_count <= batched_resident_capacity() && ctx->resident_limit > 0)
        return enqueue_batched_resident(ctx, B, max_count, u, v, x, y, z, offsets, poses0, cam, sched, results, hist,
                                        (int)hist_rows);
    const CameraK c = to_device_camera(cam);
    auto kern = is_pinhole(cam->K) ? icp_batched_kernel<true> : icp_batched_kernel<false>;
    hipLaunchKernelGGL(kern, dim3(B), dim3(NT), 0, ctx->stream, u, v, x, y, z, offsets, poses0, c, *sched, results,
                       hist, (int)hist_rows);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_linearize(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                   const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                   double kernel_threshold, int32_t min_inliers, double* H, double* b, int64_t* num_inliers,
                   double* error, int64_t* n_inside) {
    PICP_REQUIRE(ctx && cam && H && b && num_inliers && error, "picp_linearize: null argument");
    picp_schedule sc;
    memset(&sc, 0, sizeof(sc));
    sc.num_iterations = 1;
    sc.min_inliers = min_inliers;
    sc.kernel_threshold = kernel_threshold;
    sc.damping = 1.0;
    double* state = ctx->scratch + SCR_STATE;
    double* s1 = ctx->scratch + SCR_S1;
    double* s2 = ctx->scratch + SCR_S2;
    if (int rc = picp_state_init(ctx, state, w_T_c0_host, &sc)) return rc;
    DeviceGuard g(ctx->device);
    PICP_HIP_CHECK(hipMemsetAsync(s2, 0, PICP_S2_DOUBLES * sizeof(double), ctx->stream));
    if (int rc = picp_stage1(ctx, u, v, x, y, z, M, state, cam)) return rc;
    if (int rc = picp_stage1_reduce(ctx, s1)) return rc;
    double* h = ctx->pinned;
    const double* hs1 = h;
    const double* hs2 = h + (SCR_S2 - SCR_S1);
    for (int i = 0; i < 36; ++i) H[i] = 0.0;
    for (int i = 0; i < 6; ++i) b[i] = 0.0;
    // mode 0 sums the mean-filtered set; if it has fewer than min_inliers members the scan is
    // repeated in mode 1 over the stage-1 set (src/ProjectiveICP.py:307-310)
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 1) {
            const double one = 1.0;
            PICP_HIP_CHECK(hipMemcpyAsync(state + ST_MODE, &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
        if (int rc = picp_stage2(ctx, u, v, x, y, z, M, state, cam, s1)) return rc;
        if (int rc = picp_stage2_reduce(ctx, s2)) return rc;
        PICP_HIP_CHECK(hipMemcpyAsync(h, s1, (SCR_S2 - SCR_S1 + PICP_S2_DOUBLES) * sizeof(double),
                                      hipMemcpyDeviceToHost, ctx->stream));
        PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (n_inside) *n_inside = (int64_t)hs1[S1_N_INSIDE];
        if (hs1[S1_COUNT] == 0.0) { // :300
            *num_inliers = 0;
            *error = __builtin_inf();
            return PICP_OK;
        }
        if (mode == 1 || hs2[S2_N] >= (double)min_inliers) break;
    }
    int k = 0;
    for (int i = 0; i < 6; ++i)
        for (int j = i; j < 6; ++j, ++k) {
            H[i * 6 + j] = hs2[S2_H + k];
            H[j * 6 + i] = hs2[S2_H + k];
        }
    for (int i = 0; i < 6; ++i) b[i] = hs2[S2_B + i];
    *num_inliers = (int64_t)hs2[S2_N];
    *error = hs2[S2_N] > 0.0 ? hs2[S2_CHI] / hs2[S2_N] : __builtin_inf();
    return PICP_OK;
}

int picp_linearize_noalloc(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                           const double* z, int64_t M, const double* w_T_c0_host, const picp_camera* cam,
                           double kernel_threshold, int32_t min_inliers, double* out) {
    PICP_REQUIRE(ctx && cam && out, "picp_linearize_noalloc: null argument");
    if (int rc = check_corr(u, v, x, y, z, M)) return rc;
    if (M > PICP_NOALLOC_MAX_CORRESPONDENCES) {
        set_error("picp_linearize_noalloc: %lld correspondences exceed the limit of %lld of the entry point that "
                  "allocates nothing; use picp_linearize", (long long)M, (long long)PICP_NOALLOC_MAX_CORRESPONDENCES);
        return PICP_ERR_CAPACITY;
    }
    PICP_REQUIRE(ctx->noalloc != nullptr, "picp_linearize_noalloc: the ctx has no address for the static workspace");
    picp_schedule sc;
    memset(&sc, 0, sizeof(sc));
    sc.num_iterations = 1;
    sc.min_inliers = min_inliers;
    sc.kernel_threshold = kernel_threshold;
    sc.damping = 1.0;
    double* ws = ctx->noalloc; // this device's copy of the static array
    double* state = ws + NA_STATE;
    double* s1 = ws + NA_S1;
    double* s2 = ws + NA_S2;
    double* part1 = ws + NA_PART1;
    double* part2 = ws + NA_PART2;
    if (int rc = picp_state_init(ctx, state, w_T_c0_host, &sc)) return rc; // pose travels as a kernel argument
    DeviceGuard g(ctx->device);
    const int cap1 = ctx->max_blocks1 < NOALLOC_MAX_WG1 ? ctx->max_blocks1 : NOALLOC_MAX_WG1;
    const int cap2 = ctx->max_blocks2 < NOALLOC_MAX_WG2 ? ctx->max_blocks2 : NOALLOC_MAX_WG2;
    int rows1 = 0, rows2 = 0;
    if (int rc = launch_stage1(ctx, u, v, x, y, z, M, state, cam, part1, cap1, &rows1)) return rc;
    hipLaunchKernelGGL(fold_kernel<PICP_S1_DOUBLES>, dim3(1), dim3(NT), 0, ctx->stream, (const double*)part1, rows1,
                       (const double*)nullptr, s1);
    // two rounds of {stage 2, fold, decide}: the second one only does something when the first found
    // the mean-filtered set too small (src/ProjectiveICP.py:307-310); every kernel of it returns at
    // once when the first round has set the stop flag
    for (int round = 0; round < 2; ++round) {
        if (int rc = launch_stage2(ctx, u, v, x, y, z, M, state, cam, s1, 1, part2, cap2, &rows2)) return rc;
        hipLaunchKernelGGL(fold_kernel<PICP_S2_DOUBLES>, dim3(1), dim3(NT), 0, ctx->stream, (const double*)part2, rows2,
                           (const double*)state, s2);
        hipLaunchKernelGGL(linearize_decide_kernel, dim3(1), dim3(64), 0, ctx->stream, state, (const double*)s1,
                           (const double*)s2, (int)min_inliers, out);
    }
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_invert_pose(picp_ctx* ctx, const double* T_host, double* Tinv_host) {
    PICP_REQUIRE(ctx && T_host && Tinv_host, "picp_invert_pose: null argument");
    DeviceGuard g(ctx->device);
    double* d = ctx->scratch + SCR_MISC;
    PICP_HIP_CHECK(hipMemcpyAsync(d, T_host, 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(invert_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)d, d + 16);
    PICP_HIP_CHECK(hipGetLastError());
    PICP_HIP_CHECK(hipMemcpyAsync(Tinv_host, d + 16, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PICP_OK;
}

int picp_v2T(picp_ctx* ctx, const double* v6_host, double* T_host) {
    PICP_REQUIRE(ctx && v6_host && T_host, "picp_v2T: null argument");
    DeviceGuard g(ctx->device);
    double* d = ctx->scratch + SCR_MISC;
    PICP_HIP_CHECK(hipMemcpyAsync(d, v6_host, 6 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(v2T_kernel, dim3(1), dim3(64), 0, ctx->stream, (const double*)d, d + 16);
    PICP_HIP_CHECK(hipGetLastError());
    PICP_HIP_CHECK(hipMemcpyAsync(T_host, d + 16, 16 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    PICP_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return PICP_OK;
}

} // extern "C"
