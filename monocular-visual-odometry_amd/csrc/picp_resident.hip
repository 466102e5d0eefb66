// The whole pose loop in ONE launch with the correspondences held in registers.
//
// When M correspondences fit the register files of the chip (one 512-thread workgroup per CU, up to
// R_MAX correspondences per thread: 2^21 on 256 CUs), the five input arrays are read from HBM once;
// every iteration then works on registers and LDS.  The two device-wide sums an iteration needs
// (src/ProjectiveICP.py:301 needs the mean of chi over all stage-1 survivors before the inlier set
// of :302-310 can be chosen, and :313-317 sum H and b over that set) are exchanged between the
// workgroups through small records in global memory:
//
//   stage 1 on registers -> workgroup sum -> publish record -> every workgroup collects and folds all records
//   stage 2 on registers -> workgroup sum -> publish record -> every workgroup collects and folds all records
//   every workgroup runs the identical 6x6 solve + schedules on its own copy of the loop state
//
// All workgroups fold the same records in the same order, so their copies of the state stay
// bit-identical and control flow (stop test, redo mode) is uniform over the grid without a
// broadcast.  Larger inputs use the streaming kernels of picp_icp.hip.
//
// Exchange without read-modify-write atomics and without cache write-backs (a shared arrival
// counter serialises 256 arrivals and 256 pollers on one address, and a release fence makes each
// workgroup write back its whole L2: both cost tens of microseconds per exchange on 256 CUs).
// Records are written and read with device-scope atomic stores and loads, which go to memory.  A
// record entry that has not been published yet holds a sentinel -- a NaN bit pattern no sum can
// produce -- and a reader re-reads an entry until it is not the sentinel, so "has everyone
// arrived" and "fetch the data" are the same loads.  Three slots rotate: exchange k publishes into
// slot k%3; when a workgroup has collected all records of exchange k, every workgroup has
// published there and so has finished reading slot (k-1)%3, and the workgroup resets its record in
// that slot to the sentinel.  The slot is used again in exchange k+2; a reader gets there only
// after it has seen this workgroup's exchange-(k+1) record, which is stored after the reset has been
// acknowledged.
//
// Every wait is a bounded spin: the grid is never larger than the number of CUs and a workgroup
// needs less than one CU, so on an otherwise idle device all workgroups are resident and a wait
// ends within microseconds.  If the bound is ever hit (the device is shared and part of the grid
// was not scheduled), the workgroup sets a failure flag and leaves; the others run into their own
// bounds, the launch ends, and the host repeats the loop with the streaming kernels.  Every wave
// therefore reaches an exit whatever the scheduling.
#include <hip/hip_ext.h>

#include "picp_common.h"
#include "picp_device.h"

namespace picp {

constexpr int NTR = 512;                 // eight waves: two per SIMD
constexpr int RW = PICP_S2_DOUBLES;      // doubles per exchange record (both stages use one slot layout)
constexpr int R_MAX = 16;                // correspondences per thread
constexpr int R_KEEP_MAX = 8;            // up to here LDS also keeps stage 1's chi and 1/z for stage 2
constexpr int R_LEAN_MAX = 6;            // up to here the lean form of stage 2 pays with a general camera matrix (at 8 it spills)
constexpr int R_LEAN_MAX_PINHOLE = 8;    // ... and with a pinhole matrix (256 registers at 8, none spilled: 222.9 against 226.3 us per launch at 2^20 x 20)
constexpr int SLOTS = 3;
constexpr int FIRST_LOOK_DELAY = 16;      // s_sleep units (64 clocks) between publishing and the first poll
constexpr int FIRST_LOOK_DELAY_WIDE = 12; // the same for the 32-double records (swept 0 .. 20 for both: flat within 0.1 us from 12 to 16, 0.3 - 0.7 us worse at 0 .. 8)
constexpr int MAX_DEVICES = 64;           // per-device bookkeeping of kernel attributes
constexpr int SINGLE_WG_MAX = 4 * NTR;    // inputs up to here run as one workgroup (no exchange)
constexpr int MAX_GRID = 256;            // rows one fold covers with all of a thread's loads in flight
constexpr unsigned long long SENTINEL_BITS = 0xFFF8C0DEC0DE5EEDull; // a NaN payload arithmetic never produces

struct GridExchange {
    double* slots;       // [SLOTS][gridDim.x][RW]: the set of slots this launch exchanges through
    double* restore;     // the other set, which this launch fills with sentinels for the next launch (or null)
    int restore_doubles; // its length
    unsigned spin_limit; // re-reads before a waiting thread gives up
};

// A workgroup-uniform double as a scalar-register value: the loop's pose, threshold and mean are
// read from LDS (vector registers); broadcasting lane 0's copy lets the compiler keep them in scalar
// registers for the whole stage instead of 2 vector registers each.
__device__ __forceinline__ double uniform(double v) {
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ bool is_sentinel(double v) { return (unsigned long long)__double_as_longlong(v) == SENTINEL_BITS; }

struct StartPose {
    double m[16];
};

// Every record of a set of slots (or of a rank's exchange memory) holds the sentinel.  Run before a
// launch whose set is not known to be clean: the first launch of a context; after that the launches
// clean each other's sets (restore_other_set).
__global__ void resident_fill_kernel(double* slots, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) slots[i] = __longlong_as_double((long long)SENTINEL_BITS);
}

// ---- 16-byte accesses for the wide (stage-2) records ------------------------------------------------
// A record of the second exchange is 32 doubles, and every workgroup reads every record: with 8-byte
// device-scope accesses that is 32 fabric writes per workgroup and 8 192 separate 8-byte loads per
// workgroup and exchange, and the exchange took 0.8 us longer than the 4-double one at equal skew.
// 16-byte accesses halve both counts.  HIP's atomics stop at 8 bytes, so these are written as the
// instructions themselves (global_load/store_dwordx4 with the sc1 bit, i.e. device scope, as the
// 8-byte atomics compile to); the loads of one thread are issued together and waited for inside the
// same statement, so the compiler never sees a register whose load is still in flight.  Each 8-byte
// half is checked for the sentinel on its own (a 16-byte store may become visible in two halves).
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void store_pair_sc1(double* p, f64x2 v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// Eight consecutive records' pieces: one address, the record stride (RW doubles = 256 bytes) as the
// instructions' immediate offsets.
__device__ __forceinline__ void load8_pairs_sc1(const double* p, f64x2 (&v)[8]) {
    static_assert(RW * sizeof(double) == 256, "the offsets below are the record stride");
    asm volatile(
        "global_load_dwordx4 %0, %8, off sc1\n\t"
        "global_load_dwordx4 %1, %8, off offset:256 sc1\n\t"
        "global_load_dwordx4 %2, %8, off offset:512 sc1\n\t"
        "global_load_dwordx4 %3, %8, off offset:768 sc1\n\t"
        "global_load_dwordx4 %4, %8, off offset:1024 sc1\n\t"
        "global_load_dwordx4 %5, %8, off offset:1280 sc1\n\t"
        "global_load_dwordx4 %6, %8, off offset:1536 sc1\n\t"
        "global_load_dwordx4 %7, %8, off offset:1792 sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
        : "v"(p)
        : "memory");
}

// Two sets of slots alternate between the launches of a context: while a launch exchanges through one
// set, it leaves the other one full of sentinels for the launch after it -- in its prologue, behind its
// first loads, where the stores cost nothing -- instead of a fill kernel between every two loops
// (4.5 us of kernel plus a launch gap per loop).  The previous launch, which used that set, has ended
// (stream order), and the stores have landed when this launch ends.  Spread over all workgroups.
__device__ __forceinline__ void restore_other_set(const GridExchange& gx) {
    if (gx.restore == nullptr) return;
    const f64x2 sentinel = {__longlong_as_double((long long)SENTINEL_BITS), __longlong_as_double((long long)SENTINEL_BITS)};
    const int pairs = gx.restore_doubles / 2;
    for (int i = (int)blockIdx.x + (int)gridDim.x * (int)threadIdx.x; i < pairs; i += (int)gridDim.x * (int)blockDim.x)
        store_pair_sc1(gx.restore + 2 * (size_t)i, sentinel);
}

// grid_total for the 32-double records, in 16-byte pieces.  Thread (grp, pair) owns the column pair
// (2 pair, 2 pair + 1) of the eight rows 8 grp .. 8 grp + 7: 32 groups cover 256 rows.  Rows past the
// grid are read all the same (the slots are allocated for 256 rows and eight more) and dropped.
// Fixed order of additions: a thread's rows pairwise, then the four groups of a wave (lane swaps),
// then the waves pairwise.
template <int NV>
__device__ __forceinline__ bool grid_total_pairs(double mine, double& out, const GridExchange& gx, unsigned& epoch,
                                                 double* lds, int* lds_flag) {
    constexpr int PW = RW / 2;      // 16 pairs per record
    constexpr int GR = NTR / PW;    // 32 row groups
    constexpr int B = MAX_GRID / GR; // 8 rows per thread
    static_assert(B == 8 && NV <= RW, "load8_pairs_sc1 issues eight loads");
    const int G = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double* cur = gx.slots + (size_t)(epoch % SLOTS) * G * RW;
    double* old = gx.slots + (size_t)((epoch + SLOTS - 1u) % SLOTS) * G * RW; // the previous exchange's slot
    ++epoch;
    const f64x2 sentinel = {__longlong_as_double((long long)SENTINEL_BITS), __longlong_as_double((long long)SENTINEL_BITS)};
    if (threadIdx.x < 64) {
        // lane t < 16 publishes columns 2t and 2t + 1, which lanes 2t and 2t + 1 hold
        const double v = (threadIdx.x < NV) ? mine : 0.0;
        f64x2 rec;
        rec.x = __shfl(v, (2 * lane) & 63, 64);
        rec.y = __shfl(v, (2 * lane + 1) & 63, 64);
        if (lane < PW) {
            // The reset this wave stored at the end of the previous exchange has been acknowledged
            // (the stores are write-through at device scope), so it is visible before this record is.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            store_pair_sc1(cur + (size_t)blockIdx.x * RW + 2 * lane, rec);
        }
    }

    const int pair = threadIdx.x % PW, grp = threadIdx.x / PW;
    __builtin_amdgcn_s_sleep(FIRST_LOOK_DELAY_WIDE); // see grid_total

    const double* base = cur + (size_t)(grp * B) * RW + 2 * pair;
    f64x2 v[B];
#pragma unroll
    for (int k = 0; k < B; ++k) v[k] = f64x2{0.0, 0.0};
    if (grp * B < G) { // whole waves skip this on small grids
        load8_pairs_sc1(base, v);
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (grp * B + k >= G) v[k] = f64x2{0.0, 0.0};
    }

    unsigned spins = 0;
    for (;;) {
        bool missing = false;
#pragma unroll
        for (int k = 0; k < B; ++k) missing |= is_sentinel(v[k].x) || is_sentinel(v[k].y);
        if (!missing) break;
        if (++spins > gx.spin_limit) {
            *lds_flag = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < B; ++k) {
            if (is_sentinel(v[k].x)) v[k].x = __hip_atomic_load(base + k * RW, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (is_sentinel(v[k].y)) v[k].y = __hip_atomic_load(base + k * RW + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    double vx[B], vy[B];
#pragma unroll
    for (int k = 0; k < B; ++k) {
        vx[k] = v[k].x;
        vy[k] = v[k].y;
    }
    double ax = pairwise_sum(vx), ay = pairwise_sum(vy);
    // the four row groups of a wave sit 16 lanes apart
    ax = halve_bit4(ax, ax);
    ay = halve_bit4(ay, ay);
    ax = halve_bit5(ax, ax);
    ay = halve_bit5(ay, ay);
    if (lane < PW) {
        lds[wave * RW + 2 * lane] = ax;
        lds[wave * RW + 2 * lane + 1] = ay;
    }

    __syncthreads();
    double tot = 0.0;
    if (lane < NV) { // every wave: lane t ends with the total of column t
        double part[NTR / 64];
#pragma unroll
        for (int w = 0; w < NTR / 64; ++w) part[w] = lds[w * RW + lane];
        tot = pairwise_sum(part);
    }
    out = tot;
    // Every workgroup has published into `cur`, hence has finished reading `old`: reset this
    // workgroup's record there; that slot is next used two exchanges from now.
    if (threadIdx.x < PW) store_pair_sc1(old + (size_t)blockIdx.x * RW + 2 * threadIdx.x, sentinel);
    return *lds_flag == 0;
}

// Grid-wide sum of NV values.  `mine`: this workgroup's total of value tid (threads tid < NV).
// On success lane t < NV of every wave of EVERY workgroup holds the same grid total of value t in `out`
// (with a single workgroup: the lanes that hold `mine`, which block_sum leaves in every wave).
// Returns false (uniformly over the workgroup) when a wait ran into its bound.
template <int NV, int B>
__device__ __forceinline__ bool grid_total(double mine, double& out, const GridExchange& gx, unsigned& epoch,
                                           double* lds, int* lds_flag) {
    const int G = gridDim.x;
    if (G == 1) {
        out = mine;
        return true;
    }
    using Fold = PartialFold<NV, NTR, B>;
    constexpr int CW = Fold::CW, GR = Fold::G;
    static_assert(CW <= RW && RW <= 64, "the record is written by the first wave");
    if constexpr (CW == RW) return grid_total_pairs<NV>(mine, out, gx, epoch, lds, lds_flag); // the 32-double records
    static_assert(B * GR >= MAX_GRID, "one load per row in flight");
    double* cur = gx.slots + (size_t)(epoch % SLOTS) * G * RW;
    double* old = gx.slots + (size_t)((epoch + SLOTS - 1u) % SLOTS) * G * RW; // the previous exchange's slot
    ++epoch;
    if (threadIdx.x < CW) {
        // The reset this wave stored at the end of the previous exchange has been acknowledged
        // (the stores are write-through at device scope), so it is visible before this record is.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(cur + (size_t)blockIdx.x * RW + threadIdx.x, (threadIdx.x < NV) ? mine : 0.0,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // collect: thread (grp, col) owns rows grp, grp + GR, ... of column col
    const int col = threadIdx.x % CW, grp = threadIdx.x / CW;
    // A record cannot be visible sooner than a store-to-memory plus a load-from-memory latency after
    // it was published, and loads issued before that only compete with the stores for the memory
    // channels: a short pause before the first look (about 0.4 us; swept 0 .. 1.7 us on 8 .. 256
    // workgroups) shortens the whole exchange by 0.4 - 0.5 us.
    __builtin_amdgcn_s_sleep(FIRST_LOOK_DELAY);
    Fold f;
#pragma unroll
    for (int k = 0; k < B; ++k) {
        const int r = grp + k * GR;
        f.v[k] = (col < NV && r < G)
                     ? __hip_atomic_load(cur + (size_t)r * RW + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                     : 0.0;
    }
    unsigned spins = 0;
    for (;;) {
        bool missing = false;
#pragma unroll
        for (int k = 0; k < B; ++k) missing |= is_sentinel(f.v[k]);
        if (!missing) break;
        if (++spins > gx.spin_limit) {
            *lds_flag = 1;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < B; ++k)
            if (is_sentinel(f.v[k]))
                f.v[k] = __hip_atomic_load(cur + (size_t)(grp + k * GR) * RW + col, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
    }
    out = f.finish_in_wave(lds); // synchronises the workgroup once
    // Every workgroup has published into `cur`, hence has finished reading `old`: reset this
    // workgroup's record there; that slot is next used two exchanges from now.
    if (threadIdx.x < RW)
        __hip_atomic_store(old + (size_t)blockIdx.x * RW + threadIdx.x, __longlong_as_double((long long)SENTINEL_BITS),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *lds_flag == 0;
}

// ---- exchange between the GPUs of a sharded run ---------------------------------------------------
// Every rank (one process per GPU) runs the single-launch loop on its shard.  After the workgroups of
// a GPU have exchanged their sums (grid_total), workgroup 0 stores that GPU's total into a slot of
// EVERY rank's exchange memory -- device memory of the peers, mapped into this process -- and every
// workgroup of every GPU then collects the `world` records from its own GPU's memory and adds them in
// rank order.  All ranks add the same records in the same order, so the replicated loop states stay
// bit-identical, as they do between the workgroups of one GPU.
//
// Same rules as inside a GPU: an entry that has not arrived holds the sentinel, stores and loads are
// write-through / uncached (system scope here), every wait is bounded.  Two slots alternate.  Before
// workgroup 0 publishes exchange e it resets its own GPU's slot of exchange e-1: it has collected the
// local exchange that precedes e, so every local workgroup has finished reading e-1, and a peer
// writes into that slot again (exchange e+1) only after it has seen this rank's record of exchange
// e, which is stored after the reset has been acknowledged.  The exchange count runs on from launch
// to launch (epoch0), so nothing has to be cleaned between launches.
constexpr int XS = 2;                         // alternating slots
constexpr int XR = PICP_PEER_MAX_RANKS;       // records per slot
struct PeerExchange {
    double* slots[XR];   // exchange memory of every rank: [XS][XR][RW]; slots[rank] is this GPU's own
    int rank, world;     // world == 1: no exchange
    unsigned epoch0;     // exchanges completed by earlier launches
    unsigned spin_limit; // re-reads before a waiting thread gives up
    unsigned first_spin_limit; // the same for a launch's first exchange: the ranks' hosts reach their launches at
                               // different times (input set-up, a stalled process), so the first record of a peer
                               // may be seconds away; later ones follow within an iteration's time
};

template <int NV>
__device__ __forceinline__ bool cross_total(double mine, double& out, const PeerExchange& px, unsigned& xepoch,
                                            double* lds_a, double* lds_b, int* lds_flag) {
    constexpr int CW = NV <= 4 ? 4 : RW; // doubles of a record that travel
    static_assert(XR * RW <= NTR, "one lane per entry of a slot");
    const int W = px.world;
    const unsigned e = xepoch++;
    const int tid = threadIdx.x;
    __syncthreads(); // lds_a / lds_b are free: every thread is past the reads of the sums before
    if (tid < CW) lds_a[tid] = (tid < NV) ? mine : 0.0;
    double* own = px.slots[px.rank];
    double* cur = own + (size_t)(e % XS) * XR * RW;
    if (blockIdx.x == 0) {
        double* old = own + (size_t)((e + XS - 1u) % XS) * XR * RW;
        if (tid < W * RW)
            __hip_atomic_store(old + tid, __longlong_as_double((long long)SENTINEL_BITS), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the resets are in memory ...
        __syncthreads();                                  // ... those of every wave, before any record leaves
        if (tid < W * CW) {
            const int peer = tid / CW, col = tid % CW;
            __hip_atomic_store(px.slots[peer] + (size_t)(e % XS) * XR * RW + (size_t)px.rank * RW + col, lds_a[col],
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    const bool reader = tid < W * CW;
    const int r = tid / CW, col = tid % CW;
    double v = 0.0;
    if (reader) {
        __builtin_amdgcn_s_sleep(FIRST_LOOK_DELAY);
        unsigned spins = 0;
        for (;;) {
            v = __hip_atomic_load(cur + (size_t)r * RW + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (!is_sentinel(v)) break;
            if (++spins > (e == px.epoch0 ? px.first_spin_limit : px.spin_limit)) {
                *lds_flag = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        lds_b[tid] = v;
    }
    __syncthreads();
    double tot = 0.0;
    const int lane = tid & 63; // lane t < NV of EVERY wave ends with the total of value t, as after grid_total
    if (lane < NV)
        for (int k = 0; k < W; ++k) tot += lds_b[k * CW + lane]; // rank order on every GPU
    out = tot;
    return *lds_flag == 0;
}

// Finite and far from overflow in any product the stages form with it (see stage2_kept, MODERATE):
// |v| < 2^199 (about 8e59), read off the exponent field, which is all ones for infinities and NaN.
// Integer arithmetic on purpose: for the pose, which is the same in every lane, it runs on the scalar unit.
__device__ __forceinline__ unsigned exponent_field(double v) { return ((unsigned)__double2hiint(v) >> 20) & 0x7ffu; }
constexpr unsigned MODERATE_EXPONENT = 1023u + 199u;
__device__ __forceinline__ bool moderate(double v) { return exponent_field(v) < MODERATE_EXPONENT; }

// Stage 1 of one correspondence (as stage1_point) that also leaves what stage 2 needs of it in
// `keep`: chi (NaN when the pair is outside the image or above the threshold, so that every later
// comparison with it is false) and 1 / (K p)_z, the one division of the projection.
// `ok`: the lane holds a correspondence in this place (a lane past the end of the input runs the same
// arithmetic on a copy of another correspondence and counts nothing: no lane-dependent branch).
template <bool PINHOLE>
__device__ __forceinline__ void stage1_keep(const double* __restrict__ T, const CameraK& cam, double thr, double mu,
                                            double mv, double x, double y, double z, bool ok, Stage1Sums& s,
                                            double (&keep)[2]) {
    const Projected p = project_point<PINHOLE>(T, cam, x, y, z);
    const double e0 = mu - p.u, e1 = mv - p.v;         // src/ProjectiveICP.py:333
    const double chi = chi_of(e0, e1);                 // :293
    const bool inside = p.inside && ok;
    const bool in1 = inside && (chi <= thr);           // :290, :294
    s.n_inside += inside ? 1u : 0u;
    s.n1 += in1 ? 1u : 0u;
    s.sum_chi += in1 ? chi : 0.0;
    keep[0] = in1 ? chi : __builtin_nan("");
    keep[1] = p.zi;
}

// Stage 2 of a correspondence from the values stage 1 kept: the camera-frame point, K p and the pixel
// are formed again with the expressions of project_point and the kept reciprocal (an accepted pair is
// finite and in the image, where the pinhole and the general product with K agree and the reciprocal
// form of the division is the one project_point used).
// Branch-free: every lane runs the arithmetic, and a lane whose pair is not in the inlier set
// (`keep` false) has the point, the reciprocal and the residual replaced by zeros before they are
// used, so that its Jacobian and all its products are exact zeros and its sums do not change.  With
// the lane-dependent branch gone, the correspondences of a thread form one straight-line block and
// the compiler interleaves their dependent chains; behind a branch per correspondence each chain ran
// on its own, exposed to its full latency (measured: a wave alone on its SIMD needed 570 cycles per
// correspondence for about 100 instructions).
//
// MODERATE: the caller has checked that the pose, the camera matrix and everything this lane loaded
// (point and measurement, kept or not) are finite and of moderate magnitude (moderate() below).  Then
// zeroing the reciprocal alone is enough: the camera point, K p and the residual of a lane outside
// the set stay finite, every entry of its Jacobian carries the factor 1/z = 0 and is an exact zero, and
// so is every product.  That spares ten of the twelve selects per correspondence.
template <bool PINHOLE, bool HAVE_ZI = true, bool MODERATE = false>
__device__ __forceinline__ void stage2_kept(const double* __restrict__ T, const CameraK& cam, double mu, double mv,
                                            double x, double y, double z, double chi, double zi, bool keep,
                                            Stage2Sums& s) {
#pragma clang fp contract(off)
    Projected p;
    p.pcx = __builtin_fma(T[2], z, __builtin_fma(T[1], y, T[0] * x)) + T[3];
    p.pcy = __builtin_fma(T[6], z, __builtin_fma(T[5], y, T[4] * x)) + T[7];
    p.pcz = __builtin_fma(T[10], z, __builtin_fma(T[9], y, T[8] * x)) + T[11];
    if (!MODERATE) {
        p.pcx = keep ? p.pcx : 0.0;
        p.pcy = keep ? p.pcy : 0.0;
        p.pcz = keep ? p.pcz : 0.0;
    }
    if (PINHOLE) {
        p.h0 = __builtin_fma(cam.k[2], p.pcz, cam.k[0] * p.pcx);
        p.h1 = __builtin_fma(cam.k[5], p.pcz, cam.k[4] * p.pcy);
        p.h2 = cam.k[8] * p.pcz;
    } else {
        p.h0 = __builtin_fma(cam.k[2], p.pcz, __builtin_fma(cam.k[1], p.pcy, cam.k[0] * p.pcx));
        p.h1 = __builtin_fma(cam.k[5], p.pcz, __builtin_fma(cam.k[4], p.pcy, cam.k[3] * p.pcx));
        p.h2 = __builtin_fma(cam.k[8], p.pcz, __builtin_fma(cam.k[7], p.pcy, cam.k[6] * p.pcx));
    }
    if (!HAVE_ZI) zi = 1.0 / p.h2; // the caller kept only chi
    zi = keep ? zi : 0.0;
    double e0 = mu - div_with_recip(p.h0, p.h2, zi);
    double e1 = mv - div_with_recip(p.h1, p.h2, zi);
    if (!MODERATE) {
        e0 = keep ? e0 : 0.0;
        e1 = keep ? e1 : 0.0;
    }
    p.zi = zi;
    double J0[6], J1[6];
    jacobian_rows<PINHOLE>(p, cam, J0, J1);
    s.S.add<PINHOLE>(J0, J1, e0, e1);
    s.chi += keep ? chi : 0.0;
    s.n += keep ? 1u : 0u;
}

#ifndef PICP_STAGE_PROBE // tools/stage_probe.hip includes this file up to here: the stages' per-correspondence code alone

// MODERATE: the lean form of stage 2 (stage2_kept).  It needs every loaded value, the camera matrix
// (checked by the host) and every pose of the loop to be moderate(): the lanes count what is not, the
// count travels with the first exchange of every iteration, and when it is not zero every workgroup
// leaves at the same point, workgroup 0 having set the host-visible stop flag to STOP_IMMODERATE; the
// host then runs the kernel's other form, which makes no assumption.
// SHARDED: this rank's slice of a sharded run that fits on chip; after every exchange between the
// workgroups of this GPU one between the GPUs (cross_total), exactly as in icp_persistent_kernel, so a
// rank may run either kernel on its slice, whatever its peers run.  General form only (not MODERATE).
template <int R, bool PINHOLE, bool MODERATE, bool SHARDED = false>
__global__ __launch_bounds__(NTR) void icp_resident_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                          const double* __restrict__ x, const double* __restrict__ y,
                                                          const double* __restrict__ z, int64_t M,
                                                          double* __restrict__ state, StartPose pose0, CameraK cam,
                                                          picp_schedule sc, GridExchange gx, double* __restrict__ hist,
                                                          int hist_rows, double* __restrict__ host_state,
                                                          PeerExchange px) {
    static_assert(!(SHARDED && MODERATE), "the sharded form has no lean variant");
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double lds[NTR];      // workgroup sums
    __shared__ double lds_fold[NTR]; // folds of the exchanged records (a second buffer spares barriers)
    __shared__ double row[PICP_HIST_STRIDE];
    __shared__ int flag;
    __shared__ StepShared sh;
    // Dynamic LDS, thread-private columns (conflict-free): the measurements u, v as [R][2][NTR]
    // (they are only read, once or twice per iteration, so they need not occupy registers) and the
    // stage-1 results kept for stage 2 (chi, 1/z) as [R][2][NTR].
    extern __shared__ double dyn[];
    constexpr bool KEEP = (R <= R_KEEP_MAX); // above that LDS only has room for u, v: stage 2 projects again
    double* muv = dyn;
    double* kept = dyn + R * 2 * NTR;

    // correspondence (r, workgroup, thread) -> index (r * grid + workgroup) * NTR + thread: every load
    // instruction of a wave reads 64 consecutive doubles
    restore_other_set(gx);
    double rx[R], ry[R], rz[R];
    const int64_t first = (int64_t)blockIdx.x * NTR + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * NTR;
    int mine = 0; // this thread holds correspondences r = 0 .. mine-1
    bool tame = true; // everything this lane loaded is moderate()
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = first + r * step;
        const bool ok = i < M;
        const int64_t j = ok ? i : 0;
        const double mu = u[j], mv = v[j];
        muv[(r * 2 + 0) * NTR + threadIdx.x] = mu;
        muv[(r * 2 + 1) * NTR + threadIdx.x] = mv;
        rx[r] = x[j];
        ry[r] = y[j];
        rz[r] = z[j];
        homogeneous_nan(rx[r], ry[r], rz[r]);
        if (MODERATE) tame = tame && moderate(mu) && moderate(mv) && moderate(rx[r]) && moderate(ry[r]) && moderate(rz[r]);
        mine += ok ? 1 : 0;
    }

    // loop entry (src/ProjectiveICP.py:157-171): every workgroup builds its own copy of the state
    if (threadIdx.x == 0) {
        init_state(st, pose0.m, sc.kernel_threshold, sc.damping, sc.num_iterations);
        init_next(st, &sh);
        flag = 0;
    }
    __syncthreads();

    unsigned epoch = 0;
    [[maybe_unused]] unsigned xepoch = SHARDED ? px.epoch0 : 0u;
#ifdef PICP_STAMPS
    const unsigned long long core0 = __builtin_amdgcn_s_memtime(), wall0 = __builtin_amdgcn_s_memrealtime();
    // Diagnostic build only: clock stamps of workgroup 0 at the segment boundaries of an iteration,
    // left in the unused tail of the history row (entries 74 .. 79, microseconds since the iteration
    // began: stage 1, its workgroup sum, its exchange, stage 2, its workgroup sum, its exchange).
    unsigned long long stamp[7] = {0, 0, 0, 0, 0, 0, 0};
    unsigned long long kept_stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // iteration 6's, printed by every workgroup at the end
    __shared__ unsigned long long wave_stamp[7][NTR / 64]; // the same boundaries as every wave of workgroup 0 sees them
#define PICP_STAMP(k)                                                             \
    {                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
        stamp[k] = __builtin_amdgcn_s_memrealtime();                              \
        if ((threadIdx.x & 63) == 0) wave_stamp[k][threadIdx.x >> 6] = stamp[k];  \
        __builtin_amdgcn_sched_barrier(0);                                        \
    }
#else
#define PICP_STAMP(k)
#endif
    // The iteration's inputs come from `sh` (solve_step_lanes): the state block is brought up to date
    // behind the stages' back.
    while (sh.next[NEXT_STOP] == 0.0) {
        PICP_STAMP(0);
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = uniform(sh.Tinv[i]);
        const double thr = uniform(sh.next[NEXT_THR]);
        const bool all = (sh.next[NEXT_MODE] != 0.0);
        const int it = (int)sh.next[NEXT_IT];

        Stage1Sums a{0.0, 0u, 0u};
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (KEEP || r < mine) { // KEEP: no branch, the lanes past the end are masked inside
                double keep[2];
                stage1_keep<PINHOLE>(T, cam, thr, muv[(r * 2 + 0) * NTR + threadIdx.x],
                                     muv[(r * 2 + 1) * NTR + threadIdx.x], rx[r], ry[r], rz[r], r < mine, a, keep);
                if (KEEP) {
                    kept[(r * 2 + 0) * NTR + threadIdx.x] = keep[0];
                    kept[(r * 2 + 1) * NTR + threadIdx.x] = keep[1];
                }
            }
        PICP_STAMP(1);
        double g1;
        {
            constexpr int NV1 = MODERATE ? 4 : 3; // the lean form also counts the lanes with immoderate values
            static_assert(NV1 <= PICP_S1_DOUBLES, "stage-1 record");
            double tot;
            if constexpr (MODERATE) {
                // the pose is the same in every workgroup; testing it here, without branches and beside
                // stage 1's arithmetic, keeps it off the path from the solve step to stage 1
                unsigned pose_exponent = 0;
#pragma unroll
                for (int i = 0; i < 12; ++i) pose_exponent = max(pose_exponent, exponent_field(T[i]));
                const bool pose_tame = pose_exponent < MODERATE_EXPONENT;
                const double vals[4] = {a.sum_chi, (double)a.n1, (double)a.n_inside, (tame && pose_tame) ? 0.0 : 1.0};
                tot = block_sum<4, NTR, false>(vals, lds);
            } else {
                const double vals[3] = {a.sum_chi, (double)a.n1, (double)a.n_inside};
                tot = block_sum<3, NTR, false>(vals, lds);
            }
            PICP_STAMP(2);
            g1 = tot;
            if (gridDim.x > 1) {
                if (!grid_total<NV1, 2>(tot, g1, gx, epoch, lds_fold, &flag)) return;
            } else {
                __syncthreads(); // every wave has read its totals before stage 2's sum reuses `lds`
            }
            if constexpr (SHARDED) {
                if (px.world > 1) {
                    const double local = g1;
                    if (!cross_total<NV1>(local, g1, px, xepoch, lds, lds_fold, &flag)) return;
                }
            }
        }
        PICP_STAMP(3);
        // The grid totals sit in lanes 0 .. NV1-1 of every wave: read as scalars, no trip through LDS.
        if (MODERATE && lane_bcast<S1_PAD>(g1) != 0.0) { // identical in every workgroup
            if (blockIdx.x == 0 && threadIdx.x == 0 && host_state != nullptr) host_state[ST_STOP] = STOP_IMMODERATE;
            return;
        }
        const double n1 = lane_bcast<S1_COUNT>(g1); // identical in every workgroup, so the branch below is grid-uniform
        double g2 = 0.0;                            // :300 when no pair passed stage 1
        if (n1 != 0.0) {
            const double mean = lane_bcast<S1_SUM_CHI>(g1) / n1; // np.mean, src/ProjectiveICP.py:301
            Stage2Sums s;
            s.clear();
            if (KEEP) {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double chi = kept[(r * 2 + 0) * NTR + threadIdx.x];
                    // NaN (rejected by stage 1) fails both tests; `all`: the stage-1 set, else chi < mean (:302)
                    const bool in_set = (r < mine) && (chi <= thr) && (all || chi < mean);
                    stage2_kept<PINHOLE, true, MODERATE>(T, cam, muv[(r * 2 + 0) * NTR + threadIdx.x],
                                                         muv[(r * 2 + 1) * NTR + threadIdx.x], rx[r], ry[r], rz[r], chi,
                                                         kept[(r * 2 + 1) * NTR + threadIdx.x], in_set, s);
                }
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
                    if (r < mine)
                        stage2_point<PINHOLE>(T, cam, thr, mean, all, muv[(r * 2 + 0) * NTR + threadIdx.x],
                                              muv[(r * 2 + 1) * NTR + threadIdx.x], rx[r], ry[r], rz[r], s);
            }
            PICP_STAMP(4);
            double vals[S2_USED];
            pack_sets(s, vals);
            const double tot = block_sum<S2_USED, NTR, false>(vals, lds);
            PICP_STAMP(5);
            g2 = tot;
            if (!grid_total<S2_USED, 16>(tot, g2, gx, epoch, lds_fold, &flag)) return;
            if constexpr (SHARDED) {
                if (px.world > 1) {
                    const double local = g2;
                    if (!cross_total<S2_USED>(local, g2, px, xepoch, lds, lds_fold, &flag)) return;
                }
            }
            if ((threadIdx.x & 63) >= S2_USED) g2 = 0.0;
        }
        PICP_STAMP(6);

        const bool record = (blockIdx.x == 0 && hist != nullptr && it < hist_rows);
        const bool advanced = solve_step_lanes(st, g1, g2, all, sc, record ? row : nullptr, &sh);
#ifdef PICP_STAMPS
        if (record && threadIdx.x < 6) row[74 + threadIdx.x] = (double)(stamp[threadIdx.x + 1] - stamp[0]) * 0.01;
        if (it == 6)
            for (int k = 0; k < 7; ++k) kept_stamp[k] = stamp[k];
        if (it == 7) kept_stamp[7] = stamp[0]; // = end of iteration 6's solve step
#endif
        if (record) { // workgroup 0 only: the row is complete once every lane has passed this barrier
            __syncthreads();
            if (advanced && threadIdx.x < PICP_HIST_STRIDE)
                hist[(size_t)it * PICP_HIST_STRIDE + threadIdx.x] = row[threadIdx.x];
        }
    }
    __syncthreads(); // the last tail's writes to the state block
    if constexpr (SHARDED) {
        if (threadIdx.x == 0) st[ST_XEPOCH] = (double)xepoch; // exchanges between the GPUs so far (next launch's epoch0)
        __syncthreads();
    }
#ifdef PICP_STAMPS
    if (threadIdx.x == 0 && blockIdx.x == 0 && host_state != nullptr && hist != nullptr) {
        // the shader clock the loop actually ran at: core-clock counter against the constant 100 MHz one
        const unsigned long long core1 = __builtin_amdgcn_s_memtime(), wall1 = __builtin_amdgcn_s_memrealtime();
        printf("shader clock during the loop: %.0f MHz (%llu core cycles in %.1f us)\n",
               (double)(core1 - core0) / ((double)(wall1 - wall0) * 0.01), core1 - core0, (double)(wall1 - wall0) * 0.01);
    }
    if (threadIdx.x == 0 && host_state != nullptr && hist != nullptr) // (one line per workgroup, on the recorded runs only)
        printf("wg %d: %llu %llu %llu %llu %llu %llu %llu %llu\n", (int)blockIdx.x, kept_stamp[0], kept_stamp[1],
               kept_stamp[2], kept_stamp[3], kept_stamp[4], kept_stamp[5], kept_stamp[6], kept_stamp[7]);
#endif
    if (blockIdx.x == 0 && threadIdx.x < PICP_STATE_DOUBLES) {
        state[threadIdx.x] = st[threadIdx.x];
        // a second copy straight into host-visible memory spares the caller a device-to-host copy
        if (host_state != nullptr) host_state[threadIdx.x] = st[threadIdx.x];
    }
}

// ---- the whole loop in one launch for inputs beyond the register files ----------------------------
// The same loop with the correspondences streamed from memory twice per iteration (as the stage
// kernels of picp_icp.hip do) instead of held on chip: one 512-thread workgroup per CU walks its
// grid-stride share forward in stage 1 and, for inputs the memory-side cache can help with, backward
// in stage 2 (so each scan starts with what the previous one read last), and the two sums are exchanged between the workgroups inside the kernel.
// Against three launches per iteration this saves the launch ramps and gaps, the record folds in the
// prologues and the separate solve launch: about 10 us per iteration, which matters while a scan
// takes tens of microseconds.  The first loads of the next scan are issued before the exchange
// that precedes it, so their latency is hidden behind it.
//
// HOLD + HOLDR > 0: the first (HOLD + HOLDR) * gridDim.x * NTR correspondences do not take part in the
// streaming: they are read once, at the start of the launch, and stay on chip -- HOLD per thread with all five
// values in LDS (thread-private columns [HOLD][5][NTR]), HOLDR per thread with the point in registers and the
// measurement in LDS ([HOLDR][2][NTR] behind the former) -- and every scan visits them from there, one of them
// after each streaming step while that step's successor is in flight, so their arithmetic hides behind the
// memory traffic of the rest.  What is held on chip is traffic the scans do not cause: an eighth of a
// 2^23-correspondence slice.
#ifndef PICP_HOLD
#define PICP_HOLD 5
#define PICP_HOLDR 5
#endif
constexpr int PERSISTENT_HOLD = PICP_HOLD;   // 100 KB of LDS ...
constexpr int PERSISTENT_HOLDR = PICP_HOLDR; // ... + 40 KB and 30 registers per lane, beside 10 KB of static LDS
static_assert((PERSISTENT_HOLD * 5 + PERSISTENT_HOLDR * 2) * NTR * sizeof(double) <= 146 * 1024, "LDS of the held correspondences");
// ORDER: how the scans walk the arrays -- both forward; stage 2 backward (SCAN_REVERSE, for inputs the memory-side
// cache can help with); both forward with non-temporal loads (SCAN_FORWARD_NONTEMPORAL, for inputs far beyond it).
template <bool PINHOLE, int ORDER, bool SHARDED = false, int HOLD = 0, int HOLDR = 0>
__global__ __launch_bounds__(NTR) void icp_persistent_kernel(const double* __restrict__ u, const double* __restrict__ v,
                                                            const double* __restrict__ x, const double* __restrict__ y,
                                                            const double* __restrict__ z, int64_t M,
                                                            double* __restrict__ state, StartPose pose0, CameraK cam,
                                                            picp_schedule sc, GridExchange gx, double* __restrict__ hist,
                                                            int hist_rows, double* __restrict__ host_state,
                                                            PeerExchange px) {
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double lds[NTR];
    __shared__ double lds_fold[NTR];
    __shared__ double s1[PICP_S1_DOUBLES];
    __shared__ double s2[PICP_S2_DOUBLES];
    __shared__ double row[PICP_HIST_STRIDE];
    __shared__ int flag;
    __shared__ StepShared sh;
    extern __shared__ double held[]; // [HOLD][5][NTR] u, v, x, y, z, then [HOLDR][2][NTR] u, v: the correspondences kept on chip
    constexpr int HELD = HOLD + HOLDR;
    // (vector values, not arrays: an array that is indexed with the loop-carried `h` ends up in scratch memory)
    typedef double HeldVec __attribute__((ext_vector_type(HOLDR > 0 ? HOLDR : 1)));
    [[maybe_unused]] HeldVec rx, ry, rz;
    const int64_t M_all = M;
    if (HELD > 0) { // the streams cover what is not held (an even number is held: the rest stays 16-byte aligned)
        const int64_t M_held = (int64_t)HELD * gridDim.x * NTR;
        u += M_held, v += M_held, x += M_held, y += M_held, z += M_held;
        M -= M_held;
    }
    CorrStream<2, NTR, ORDER == SCAN_REVERSE ? SCAN_FORWARD : ORDER> fwd;
    CorrStream<2, NTR, ORDER> rev; // stage 2's direction: see stage2_reverse()
    fwd.start(u, v, x, y, z, M);
    if (HELD > 0) {
        const int64_t M_held = M_all - M;
        const double *u0 = u - M_held, *v0 = v - M_held, *x0 = x - M_held, *y0 = y - M_held, *z0 = z - M_held;
#pragma unroll
        for (int h = 0; h < HELD; ++h) {
            const int64_t i = ((int64_t)h * gridDim.x + blockIdx.x) * NTR + threadIdx.x; // < M_held <= M_all (host)
            double px = x0[i], py = y0[i], pz = z0[i];
            homogeneous_nan(px, py, pz);
            if (h < HOLD) {
                held[(h * 5 + 0) * NTR + threadIdx.x] = u0[i];
                held[(h * 5 + 1) * NTR + threadIdx.x] = v0[i];
                held[(h * 5 + 2) * NTR + threadIdx.x] = px;
                held[(h * 5 + 3) * NTR + threadIdx.x] = py;
                held[(h * 5 + 4) * NTR + threadIdx.x] = pz;
            } else {
                held[(HOLD * 5 + (h - HOLD) * 2 + 0) * NTR + threadIdx.x] = u0[i];
                held[(HOLD * 5 + (h - HOLD) * 2 + 1) * NTR + threadIdx.x] = v0[i];
                rx[h - HOLD] = px, ry[h - HOLD] = py, rz[h - HOLD] = pz;
            }
        }
    }
    // Visits held correspondence h (the same h in every thread).
    auto visit_held = [&](int h, auto&& visit) __attribute__((always_inline)) {
        if (h < HOLD) {
            const double* c = held + (size_t)(h * 5) * NTR + threadIdx.x;
            visit(c[0], c[NTR], c[2 * NTR], c[3 * NTR], c[4 * NTR]);
        } else if (HOLDR > 0) {
            const int r = h - HOLD;
            const double* c = held + (size_t)(HOLD * 5 + r * 2) * NTR + threadIdx.x;
            visit(c[0], c[NTR], rx[r], ry[r], rz[r]);
        }
    };
    restore_other_set(gx);
    if (threadIdx.x == 0) {
        init_state(st, pose0.m, sc.kernel_threshold, sc.damping, sc.num_iterations);
        flag = 0;
    }
    __syncthreads();

    unsigned epoch = 0;
    unsigned xepoch = px.epoch0;
    while (st[ST_STOP] == 0.0) {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = uniform(st[ST_C_T_W + i]);
        const double thr = uniform(st[ST_THR]);
        const bool all = (st[ST_MODE] != 0.0);
        const int it = (int)st[ST_IT];

        Stage1Sums a{0.0, 0u, 0u};
        auto visit1 = [&](double mu, double mv, double qx, double qy, double qz) __attribute__((always_inline)) {
            stage1_point<PINHOLE>(T, cam, thr, mu, mv, qx, qy, qz, a);
        };
        if (HELD > 0) {
            int h = 0; // the same in every thread: the held correspondences are visited one per streaming step
            fwd.run(visit1, [&]() __attribute__((always_inline)) {
                if (h < HELD) visit_held(h++, visit1);
            });
            while (h < HELD) visit_held(h++, visit1);
        } else {
            fwd.run(visit1);
        }
        rev.start(u, v, x, y, z, M); // in flight during the exchange
        {
            const double vals[3] = {a.sum_chi, (double)a.n1, (double)a.n_inside};
            const double tot = block_sum<3, NTR, false>(vals, lds);
            double g = tot;
            if (gridDim.x > 1 && !grid_total<3, 2>(tot, g, gx, epoch, lds_fold, &flag)) return;
            if (SHARDED && px.world > 1) {
                const double local = g;
                if (!cross_total<3>(local, g, px, xepoch, lds, lds_fold, &flag)) return;
            }
            if (threadIdx.x < PICP_S1_DOUBLES) s1[threadIdx.x] = (threadIdx.x < 3) ? g : 0.0;
            __syncthreads();
        }
        const double n1 = s1[S1_COUNT]; // identical in every workgroup, so the branch below is grid-uniform
        if (n1 != 0.0) {
            const double mean = uniform(s1[S1_SUM_CHI] / n1); // np.mean, src/ProjectiveICP.py:301
            Stage2Sums s;
            s.clear();
            auto visit2 = [&](double mu, double mv, double qx, double qy, double qz) __attribute__((always_inline)) {
                stage2_point<PINHOLE>(T, cam, thr, mean, all, mu, mv, qx, qy, qz, s);
            };
            if (HELD > 0) {
                int h = 0;
                rev.run(visit2, [&]() __attribute__((always_inline)) {
                    if (h < HELD) visit_held(h++, visit2);
                });
                while (h < HELD) visit_held(h++, visit2);
            } else {
                rev.run(visit2);
            }
            fwd.start(u, v, x, y, z, M); // the next iteration's first loads
            double vals[S2_USED];
            pack_sets(s, vals);
            const double tot = block_sum<S2_USED, NTR, false>(vals, lds);
            double g = tot;
            if (gridDim.x > 1 && !grid_total<S2_USED, 16>(tot, g, gx, epoch, lds_fold, &flag)) return;
            if (SHARDED && px.world > 1) {
                const double local = g;
                if (!cross_total<S2_USED>(local, g, px, xepoch, lds, lds_fold, &flag)) return;
            }
            if (threadIdx.x < PICP_S2_DOUBLES) s2[threadIdx.x] = (threadIdx.x < S2_USED) ? g : 0.0;
            __syncthreads();
        } else {
            fwd.start(u, v, x, y, z, M);
            if (threadIdx.x < PICP_S2_DOUBLES) s2[threadIdx.x] = 0.0; // :300
            __syncthreads();
        }

        const bool record = (blockIdx.x == 0 && hist != nullptr && it < hist_rows);
        const bool advanced = solve_step(st, s1, s2, sc, record ? row : nullptr, &sh);
        if (record && advanced && threadIdx.x < PICP_HIST_STRIDE)
            hist[(size_t)it * PICP_HIST_STRIDE + threadIdx.x] = row[threadIdx.x];
    }
    if (SHARDED && threadIdx.x == 0) st[ST_XEPOCH] = (double)xepoch; // exchanges between the GPUs so far (next launch's epoch0)
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < PICP_STATE_DOUBLES) {
        state[threadIdx.x] = st[threadIdx.x];
        if (host_state != nullptr) host_state[threadIdx.x] = st[threadIdx.x];
    }
}

// ---- one problem per workgroup, correspondences held on chip ------------------------------------
// The same loop body for a batch of independent problems (problem p = correspondences
// [offsets[p], offsets[p+1])): a 256-thread workgroup per problem, R correspondences per thread,
// no exchange between workgroups.  Two or three workgroups share a CU (registers and LDS are sized
// for that), so one problem's single-lane solve step overlaps the others' stages.  A problem larger than
// R * 256 correspondences is not run (its result says so: num_iterations = -1); the host picks R
// from the caller's bound on the problem size.
constexpr int NTB = 256;
constexpr int BATCHED_THREE_PER_CU_MAX_R = 8;

template <int R, bool PINHOLE>
__global__ __launch_bounds__(NTB, (R <= BATCHED_THREE_PER_CU_MAX_R) ? 3 : 2) void icp_batched_resident_kernel(
    const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ x, const double* __restrict__ y,
    const double* __restrict__ z, const int64_t* __restrict__ offsets, const double* __restrict__ poses0, CameraK cam,
    picp_schedule sc, picp_result* __restrict__ results, double* __restrict__ hist, int hist_rows) {
    __shared__ double st[PICP_STATE_DOUBLES];
    __shared__ double lds[NTB];
    __shared__ StepShared sh;
    extern __shared__ double dyn[]; // u, v as [R][2][NTB], then chi, 1/z of stage 1 as [R][2][NTB]
    double* muv = dyn;
    double* kept = dyn + R * 2 * NTB;
    // Up to 8 per thread only chi is kept (1/z is formed again in stage 2): 48 KB of LDS and 168
    // registers per lane, so three workgroups share a CU; measured 10 % faster on 4 096 x 2 048 than
    // two workgroups that also keep 1/z.
    constexpr int KS = (R <= BATCHED_THREE_PER_CU_MAX_R) ? 1 : 2;

    const int prob = blockIdx.x;
    const int64_t beg = offsets[prob];
    const int64_t cnt = offsets[prob + 1] - beg;
    if (cnt > (int64_t)R * NTB) {
        if (threadIdx.x == 0) {
            picp_result r;
            for (int i = 0; i < 16; ++i) r.T[i] = poses0[(size_t)prob * 16 + i];
            r.valid = 0;
            r.num_iterations = -1;
            r.best_index = 0;
            r.reserved = 0;
            r.best_error = d_inf();
            results[prob] = r;
        }
        return;
    }
    if (threadIdx.x == 0) {
        init_state(st, poses0 + (size_t)prob * 16, sc.kernel_threshold, sc.damping, sc.num_iterations);
        init_next(st, &sh);
    }
    double rx[R], ry[R], rz[R];
    int mine = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = (int64_t)r * NTB + threadIdx.x;
        const bool ok = i < cnt;
        const int64_t j = beg + (ok ? i : 0);
        const bool any = cnt > 0;
        muv[(r * 2 + 0) * NTB + threadIdx.x] = any ? u[j] : 0.0;
        muv[(r * 2 + 1) * NTB + threadIdx.x] = any ? v[j] : 0.0;
        rx[r] = any ? x[j] : 0.0;
        ry[r] = any ? y[j] : 0.0;
        rz[r] = any ? z[j] : 0.0;
        homogeneous_nan(rx[r], ry[r], rz[r]);
        mine += ok ? 1 : 0;
    }
    __syncthreads();
    double* myhist = (hist != nullptr) ? hist + (size_t)prob * hist_rows * PICP_HIST_STRIDE : nullptr;

    // As in the grid-wide loop: the workgroup sums stay in registers (lane t of every wave holds entry t)
    // and the iteration's inputs come from `sh` (solve_step_lanes).
    while (sh.next[NEXT_STOP] == 0.0) {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = uniform(sh.Tinv[i]);
        const double thr = uniform(sh.next[NEXT_THR]);
        const bool all = (sh.next[NEXT_MODE] != 0.0);
        const int it = (int)sh.next[NEXT_IT];

        Stage1Sums a{0.0, 0u, 0u};
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r < mine) { // (branches kept here: the other workgroups of the CU cover a chain's latency)
                double keep[2];
                stage1_keep<PINHOLE>(T, cam, thr, muv[(r * 2 + 0) * NTB + threadIdx.x],
                                     muv[(r * 2 + 1) * NTB + threadIdx.x], rx[r], ry[r], rz[r], true, a, keep);
                kept[(r * KS + 0) * NTB + threadIdx.x] = keep[0];
                if (KS == 2) kept[(r * KS + 1) * NTB + threadIdx.x] = keep[1];
            }
        double g1;
        {
            const double vals[3] = {a.sum_chi, (double)a.n1, (double)a.n_inside};
            g1 = block_sum<3, NTB>(vals, lds); // (trailing barrier: stage 2's sum reuses `lds`)
        }
        const double n1 = lane_bcast<S1_COUNT>(g1);
        double g2 = 0.0; // :300 when no pair passed stage 1
        if (n1 != 0.0) {
            const double mean = lane_bcast<S1_SUM_CHI>(g1) / n1; // np.mean, src/ProjectiveICP.py:301
            Stage2Sums s;
            s.clear();
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r < mine) {
                    const double chi = kept[(r * KS + 0) * NTB + threadIdx.x];
                    if ((chi <= thr) && (all || chi < mean))
                        stage2_kept<PINHOLE, KS == 2>(T, cam, muv[(r * 2 + 0) * NTB + threadIdx.x],
                                                      muv[(r * 2 + 1) * NTB + threadIdx.x], rx[r], ry[r], rz[r], chi,
                                                      KS == 2 ? kept[(r * KS + 1) * NTB + threadIdx.x] : 0.0, true, s);
                }
            double vals[S2_USED];
            pack_sets(s, vals);
            g2 = block_sum<S2_USED, NTB, false>(vals, lds); // the tail's barrier comes before `lds` is written again
            if ((threadIdx.x & 63) >= S2_USED) g2 = 0.0;
        }
        double* row = (myhist != nullptr && it < hist_rows) ? myhist + (size_t)it * PICP_HIST_STRIDE : nullptr;
        // The tail is a latency chain on single lanes while the other workgroups of the CU stream
        // through their stages: its waves go first whenever they can issue (2 % on 4 096 x 2 048).
        __builtin_amdgcn_s_setprio(3);
        (void)solve_step_lanes(st, g1, g2, all, sc, row, &sh);
        __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads(); // the last tail's writes to the state block
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

static CameraK camera_of(const picp_camera* cam) {
    CameraK c;
    for (int i = 0; i < 9; ++i) c.k[i] = cam->K[i];
    c.width = cam->width;
    c.height = cam->height;
    return c;
}

static int resident_grid(const picp_ctx* ctx) {
    const int g = ctx->num_cus < MAX_GRID ? ctx->num_cus : MAX_GRID;
    return (ctx->grid_limit > 0 && ctx->grid_limit < g) ? ctx->grid_limit : g;
}

// One set: three slots of up to MAX_GRID records, read in blocks of eight records that may reach past a
// slot's last record (grid_total_pairs).  A context holds two sets (restore_other_set).
int resident_slot_rows(const picp_ctx*) { return SLOTS * MAX_GRID + 8; }

// The set of slots for the next launch of `grid` workgroups, clean, and the other set for that launch to
// restore; only the first launch of a context (and the one after a failed HIP call) needs a fill kernel.
static GridExchange take_slots(picp_ctx* ctx, int grid) {
    GridExchange gx{ctx->rslots, nullptr, 0, ctx->resident_spin_limit};
    if (grid <= 1) return gx; // a single workgroup exchanges nothing
    const int n = resident_slot_rows(ctx) * RW;
    const int set = ctx->slot_set;
    gx.slots = ctx->rslots + (size_t)set * n;
    gx.restore = ctx->rslots + (size_t)(1 - set) * n;
    gx.restore_doubles = n;
    if (!ctx->slots_clean[set])
        hipLaunchKernelGGL(resident_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, gx.slots, n);
    ctx->slots_clean[set] = 0;     // the launch dirties its own set ...
    ctx->slots_clean[1 - set] = 1; // ... and leaves the other one clean, whether it completes or gives up
    ctx->slot_set = 1 - set;
    return gx;
}

// The kernels' view of the exchange between the GPUs: none (world 1), or the ctx's connected one.
static PeerExchange peer_exchange_of(const picp_ctx* ctx, bool sharded) {
    PeerExchange px;
    memset(&px, 0, sizeof(px));
    px.world = 1;
    if (sharded) {
        for (int k = 0; k < ctx->peer_world; ++k) px.slots[k] = ctx->peer_slots[k];
        px.rank = ctx->peer_rank;
        px.world = ctx->peer_world;
        px.epoch0 = ctx->peer_epoch;
        px.spin_limit = ctx->peer_spin_limit;
        px.first_spin_limit = ctx->peer_spin_limit > (0xffffffffu >> 4) ? 0xffffffffu : ctx->peer_spin_limit << 4;
    }
    return px;
}

int64_t resident_capacity(const picp_ctx* ctx) {
    const int64_t cap = (int64_t)resident_grid(ctx) * NTR * R_MAX;
    return ctx->resident_limit < cap ? ctx->resident_limit : cap;
}

template <int R>
static void launch_resident(picp_ctx* ctx, int grid, const double* u, const double* v, const double* x, const double* y,
                            const double* z, int64_t M, double* state, const StartPose& pose, const picp_camera* cam,
                            const picp_schedule* sched, double* hist, int hist_rows, double* host_state, hipEvent_t e0,
                            hipEvent_t e1, bool sharded = false) {
    const CameraK c = camera_of(cam);
    const GridExchange gx = take_slots(ctx, grid);
    const PeerExchange px = peer_exchange_of(ctx, sharded);
    // The lean form where it pays (up to R_LEAN_MAX correspondences per thread), unless a recent loop met
    // immoderate values or the camera matrix itself is not moderate.  Sharded: the general form only.
    const bool pinhole = is_pinhole(cam->K);
    constexpr bool LEAN_P = (R <= R_LEAN_MAX_PINHOLE), LEAN_G = (R <= R_LEAN_MAX);
    const bool can_lean = pinhole ? LEAN_P : LEAN_G;
    bool lean = can_lean && ctx->full_select_calls == 0 && !sharded;
    for (int i = 0; i < 9; ++i) lean = lean && fabs(cam->K[i]) < 0x1p199; // moderate(), as the kernel tests its values
    if (can_lean && ctx->full_select_calls > 0 && !sharded) ctx->full_select_calls -= 1;
    auto kern = pinhole ? (lean ? icp_resident_kernel<R, true, LEAN_P> : icp_resident_kernel<R, true, false>)
                        : (lean ? icp_resident_kernel<R, false, LEAN_G> : icp_resident_kernel<R, false, false>);
    if constexpr (R == 2 || R == 8 || R == 16) { // the sizes a sharded run is rounded up to (enqueue_resident)
        if (sharded) kern = is_pinhole(cam->K) ? icp_resident_kernel<R, true, false, true> : icp_resident_kernel<R, false, false, true>;
    }
    constexpr unsigned kept_bytes = R * (R <= R_KEEP_MAX ? 4 : 2) * NTR * sizeof(double); // u, v (, chi, 1/z)
    if (kept_bytes > 48 * 1024) { // above the default dynamic-LDS allowance: raise it once per kernel and device
        static bool raised[8][MAX_DEVICES] = {};
        const int which = (is_pinhole(cam->K) ? 1 : 0) + (lean ? 2 : 0) + (sharded ? 4 : 0);
        if (ctx->device >= MAX_DEVICES || !raised[which][ctx->device]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)kept_bytes);
            if (ctx->device < MAX_DEVICES) raised[which][ctx->device] = true;
        }
    }
    if (e0 != nullptr || e1 != nullptr) // events bound to this dispatch: no marker packet of their own on the stream
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NTR), kept_bytes, ctx->stream, e0, e1, 0, u, v, x, y, z, M, state, pose, c,
                              *sched, gx, hist, hist_rows, host_state, px);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTR), kept_bytes, ctx->stream, u, v, x, y, z, M, state, pose, c, *sched,
                           gx, hist, hist_rows, host_state, px);
}

// Enqueues the loop, starting from w_T_c0_host.  Precondition:
// 0 < M <= resident_capacity(ctx).  A launch that gave up at a grid-wide wait leaves the state's
// stop flag clear (the final state is only written by a loop that ran to its end).  host_state:
// optional host-visible (pinned) block that also receives the final state, cleared here first.
int enqueue_resident(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                     const double* z, int64_t M, const double* w_T_c0_host, double* state, const picp_camera* cam,
                     const picp_schedule* sched, double* hist, int hist_rows, double* host_state, hipEvent_t e0,
                     hipEvent_t e1, bool sharded) {
    PICP_REQUIRE(M > 0 && M <= resident_capacity(ctx), "resident loop: %lld correspondences exceed the capacity",
                 (long long)M);
    PICP_REQUIRE(!sharded || (ctx->peer_connected && !ctx->peer_broken),
                 "sharded loop: the exchange between the GPUs is not usable");
    // fewest correspondences per thread that still fit one workgroup per CU, then no more
    // workgroups than that R needs
    // ... except that up to SINGLE_WG_MAX correspondences one workgroup does everything: a few more
    // correspondences per thread cost less than the two exchanges a second workgroup brings
    // (per iteration, measured: 1 024 correspondences 5.7 us against 8.0 us, 2 048 6.9 against 8.0, 3 072 no gain).
    const int64_t lanes = (M <= SINGLE_WG_MAX) ? NTR : (int64_t)resident_grid(ctx) * NTR;
    const int64_t per_thread = (M + lanes - 1) / lanes;
    static const int steps[] = {1, 2, 3, 4, 6, 8, 12, 16};
    int R = 16;
    for (int k = 7; k >= 0; --k)
        if (steps[k] >= per_thread) R = steps[k];
    if (sharded) R = per_thread <= 2 ? 2 : (per_thread <= 8 ? 8 : 16); // three instantiations of the sharded form
    const int grid = (int)((M + (int64_t)NTR * R - 1) / ((int64_t)NTR * R));
    PICP_REQUIRE(grid >= 1 && grid <= resident_grid(ctx), "resident loop: grid %d does not fit %d CUs", grid,
                 resident_grid(ctx));
    if (host_state != nullptr) host_state[ST_STOP] = 0.0;
    StartPose pose;
    memcpy(pose.m, w_T_c0_host, sizeof(pose.m));
#define PICP_RESIDENT_CASE(RR) \
    case RR: launch_resident<RR>(ctx, grid, u, v, x, y, z, M, state, pose, cam, sched, hist, hist_rows, host_state, e0, e1, sharded); break;
    switch (R) {
        PICP_RESIDENT_CASE(1)
        PICP_RESIDENT_CASE(2)
        PICP_RESIDENT_CASE(3)
        PICP_RESIDENT_CASE(4)
        PICP_RESIDENT_CASE(6)
        PICP_RESIDENT_CASE(8)
        PICP_RESIDENT_CASE(12)
    default:
        launch_resident<16>(ctx, grid, u, v, x, y, z, M, state, pose, cam, sched, hist, hist_rows, host_state, e0, e1, sharded);
        break;
    }
#undef PICP_RESIDENT_CASE
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

// The streaming single-launch loop (icp_persistent_kernel) for 16-byte aligned arrays of any length.
// Same completion protocol as enqueue_resident.  sharded: this rank's slice of a sharded run, with the
// exchange between the GPUs inside the kernel (the ctx's peer exchange must be connected); M may then
// be zero (a rank without correspondences still takes part in every exchange).
int enqueue_persistent(picp_ctx* ctx, const double* u, const double* v, const double* x, const double* y,
                       const double* z, int64_t M, const double* w_T_c0_host, double* state, const picp_camera* cam,
                       const picp_schedule* sched, double* hist, int hist_rows, double* host_state, hipEvent_t e0,
                       hipEvent_t e1, bool sharded) {
    PICP_REQUIRE(M > 0 || sharded, "persistent loop: no correspondences");
    const int64_t steps = ((M >> 1) + NTR - 1) / NTR; // workgroups needed for one grid-stride step
    const int grid = (int)(steps < 1 ? 1 : (steps > resident_grid(ctx) ? resident_grid(ctx) : steps));
    if (host_state != nullptr) host_state[ST_STOP] = 0.0;
    StartPose pose;
    memcpy(pose.m, w_T_c0_host, sizeof(pose.m));
    const CameraK c = camera_of(cam);
    PICP_REQUIRE(!sharded || (ctx->peer_connected && !ctx->peer_broken),
                 "sharded loop: the exchange between the GPUs is not usable");
    const GridExchange gx = take_slots(ctx, grid);
    const PeerExchange px = peer_exchange_of(ctx, sharded);
    const bool pin = is_pinhole(cam->K);
    const bool rev = stage2_reverse(ctx, M);
    // Part of the input on chip for the whole launch when at least one full streaming step is left beside it.
    const bool hold = ctx->persistent_hold && M >= (int64_t)(PERSISTENT_HOLD + PERSISTENT_HOLDR + 2) * grid * NTR;
    using Kern = void (*)(const double*, const double*, const double*, const double*, const double*, int64_t, double*,
                          StartPose, CameraK, picp_schedule, GridExchange, double*, int, double*, PeerExchange);
    const int order = rev ? SCAN_REVERSE : (scan_nontemporal(ctx, M) ? SCAN_FORWARD_NONTEMPORAL : SCAN_FORWARD);
#define PICP_PK(PIN, ORD, SH, H, HR) icp_persistent_kernel<PIN, ORD, SH, H, HR>
#define PICP_PK_ORDERS(SH, H, HR)                                                                       \
    {                                                                                                   \
        {PICP_PK(false, SCAN_FORWARD, SH, H, HR), PICP_PK(true, SCAN_FORWARD, SH, H, HR)},              \
            {PICP_PK(false, SCAN_REVERSE, SH, H, HR), PICP_PK(true, SCAN_REVERSE, SH, H, HR)},          \
        {                                                                                               \
            PICP_PK(false, SCAN_FORWARD_NONTEMPORAL, SH, H, HR), PICP_PK(true, SCAN_FORWARD_NONTEMPORAL, SH, H, HR) \
        }                                                                                               \
    }
    static const Kern table[2][2][3][2] = {
        {PICP_PK_ORDERS(false, 0, 0), PICP_PK_ORDERS(true, 0, 0)},
        {PICP_PK_ORDERS(false, PERSISTENT_HOLD, PERSISTENT_HOLDR), PICP_PK_ORDERS(true, PERSISTENT_HOLD, PERSISTENT_HOLDR)}};
#undef PICP_PK_ORDERS
#undef PICP_PK
    const Kern kern = table[hold ? 1 : 0][sharded ? 1 : 0][order][pin ? 1 : 0];
    const unsigned held_bytes = hold ? (PERSISTENT_HOLD * 5 + PERSISTENT_HOLDR * 2) * NTR * sizeof(double) : 0;
    if (hold) { // above the default dynamic-LDS allowance: raise it once per kernel and device
        static bool raised[12][MAX_DEVICES] = {};
        const int which = (sharded ? 6 : 0) + order * 2 + (pin ? 1 : 0);
        if (ctx->device >= MAX_DEVICES || !raised[which][ctx->device]) {
            PICP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)held_bytes));
            if (ctx->device < MAX_DEVICES) raised[which][ctx->device] = true;
        }
    }
    if (e0 != nullptr || e1 != nullptr) // events bound to this dispatch: no marker packet of their own on the stream
        hipExtLaunchKernelGGL(kern, dim3(grid), dim3(NTR), held_bytes, ctx->stream, e0, e1, 0, u, v, x, y, z, M, state, pose, c,
                              *sched, gx, hist, hist_rows, host_state, px);
    else
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NTR), held_bytes, ctx->stream, u, v, x, y, z, M, state, pose, c, *sched, gx,
                           hist, hist_rows, host_state, px);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

// Correspondences of an M-correspondence input (or slice) that the streaming single launch keeps on chip.
int64_t persistent_held(const picp_ctx* ctx, int64_t M) {
    const int64_t steps = ((M >> 1) + NTR - 1) / NTR;
    const int64_t grid = steps < 1 ? 1 : (steps > resident_grid(ctx) ? resident_grid(ctx) : steps);
    const bool hold = ctx->persistent_hold && M >= (int64_t)(PERSISTENT_HOLD + PERSISTENT_HOLDR + 2) * grid * NTR;
    return hold ? (int64_t)(PERSISTENT_HOLD + PERSISTENT_HOLDR) * grid * NTR : 0;
}

int peer_slot_doubles() { return XS * XR * RW; }

// Sentinels into a rank's exchange memory for the GPUs (set-up only).
void peer_fill(picp_ctx* ctx, double* slots) {
    const int n = XS * XR * RW;
    hipLaunchKernelGGL(resident_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, slots, n);
}

template <int R>
static void launch_batched(picp_ctx* ctx, int B, const double* u, const double* v, const double* x, const double* y,
                           const double* z, const int64_t* offsets, const double* poses0, const picp_camera* cam,
                           const picp_schedule* sched, picp_result* results, double* hist, int hist_rows) {
    const CameraK c = camera_of(cam);
    auto kern = is_pinhole(cam->K) ? icp_batched_resident_kernel<R, true> : icp_batched_resident_kernel<R, false>;
    constexpr unsigned dyn_bytes = R * (R <= BATCHED_THREE_PER_CU_MAX_R ? 3 : 4) * NTB * sizeof(double);
    if (dyn_bytes > 48 * 1024) {
        static bool raised[2][MAX_DEVICES] = {};
        const int which = is_pinhole(cam->K) ? 1 : 0;
        if (ctx->device >= MAX_DEVICES || !raised[which][ctx->device]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)dyn_bytes);
            if (ctx->device < MAX_DEVICES) raised[which][ctx->device] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3(B), dim3(NTB), dyn_bytes, ctx->stream, u, v, x, y, z, offsets, poses0, c, *sched, results,
                       hist, hist_rows);
}

// Largest problem the on-chip batched kernel takes.
int64_t batched_resident_capacity() { return (int64_t)NTB * 16; }

// Batch of B problems, none larger than max_count <= batched_resident_capacity().
int enqueue_batched_resident(picp_ctx* ctx, int B, int64_t max_count, const double* u, const double* v, const double* x,
                             const double* y, const double* z, const int64_t* offsets, const double* poses0,
                             const picp_camera* cam, const picp_schedule* sched, picp_result* results, double* hist,
                             int hist_rows) {
    const int64_t per = (max_count + NTB - 1) / NTB;
    if (per <= 1) launch_batched<1>(ctx, B, u, v, x, y, z, offsets, poses0, cam, sched, results, hist, hist_rows);
    else if (per <= 2) launch_batched<2>(ctx, B, u, v, x, y, z, offsets, poses0, cam, sched, results, hist, hist_rows);
    else if (per <= 4) launch_batched<4>(ctx, B, u, v, x, y, z, offsets, poses0, cam, sched, results, hist, hist_rows);
    else if (per <= 8) launch_batched<8>(ctx, B, u, v, x, y, z, offsets, poses0, cam, sched, results, hist, hist_rows);
    else launch_batched<16>(ctx, B, u, v, x, y, z, offsets, poses0, cam, sched, results, hist, hist_rows);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

#endif // PICP_STAGE_PROBE

} // namespace picp
