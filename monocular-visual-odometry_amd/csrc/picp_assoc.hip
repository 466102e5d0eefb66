// Appearance association (src/data_association.py:3-57), camera projection of point sets
// (src/Camera.py:82-137) and the pair gather, on gfx950.
//
// The reference matches descriptors by Python list equality: every component equal under float
// comparison, all equal pairs reported, query-major order.  That is an all-pairs test, and it is
// kept all-pairs here.  What is tuned is the per-pair cost: equal doubles have equal low 32-bit
// words (+0.0 and -0.0 differ only in the high word, and NaN is excluded by the full test), so
// each pair is first screened with one 32-bit integer compare of component 0's low word; only
// the pairs that pass (true matches plus a 2^-32 fraction of the rest) get the full float
// comparison of all D components.  Set-2 screening keys reach the compares as scalar operands;
// each lane screens R query rows against every key.
#include <stdlib.h>

#include "picp_common.h"
#include "picp_device.h"

namespace picp {

constexpr int ANT = 256;        // threads per workgroup
constexpr int SCAN_ITEMS = 4;   // prefix-sum elements per thread
constexpr int SCAN_BLOCK = ANT * SCAN_ITEMS;

__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t m = a < b ? a : b;
    return m < c ? m : c; // v_min3_u32
}

__device__ __forceinline__ uint32_t low_word(double d) { return (uint32_t)__double_as_longlong(d); }

// n_init > 0: also clears the per-query results the split scan accumulates into (count 0, first "none").
__global__ __launch_bounds__(ANT) void assoc_keys_kernel(const double* __restrict__ app, int64_t N, int D,
                                                         uint32_t* __restrict__ key, int64_t* __restrict__ count,
                                                         int64_t* __restrict__ first, int64_t n_init) {
    const int64_t i = (int64_t)blockIdx.x * ANT + threadIdx.x;
    if (i < N) key[i] = low_word(app[i * D]);
    if (i < n_init) {
        count[i] = 0;
        first[i] = INT64_MAX;
    }
}

// Full comparison of two descriptor rows (src/data_association.py:31): list equality = every
// component equal under float ==.
__device__ __forceinline__ bool rows_equal(const double* __restrict__ a, const double* __restrict__ b, int D) {
    bool eq = true;
    for (int d = 0; d < D; ++d) eq = eq && (a[d] == b[d]);
    return eq;
}

// For each query row i: number of equal rows of set 2 and the first of them.
//
// The screening keys of set 2 are the same for every lane, so they are fetched with scalar loads
// (eight keys per step, the next step's keys requested before the current ones are compared) and
// enter the compares as scalar operands: per pair one v_cmp_eq_u32 into a lane mask and one scalar
// OR, nothing else -- no LDS staging, no barriers.  A lane screens R query rows per key.  Only when
// some lane of the wave hits within a step does the wave look at that step's keys one by one.
// key2 must be readable up to index round_up(N2, 8) + 7 (the workspace is over-allocated).
// SPLIT: the rows of set 2 are cut into gridDim.y segments of seg_len rows (a multiple of 8) and workgroup
// (x, y) screens its query rows against segment y only -- for inputs with too few query rows to fill the
// chip with one workgroup per 256 R of them (a rank's share of a sharded association).  The segments'
// results meet in count / first (cleared beforehand) through integer atomics, issued only by the lanes
// that found a match: the number of matches and the smallest matching row do not depend on the order.
template <int R, bool SPLIT = false>
__global__ __launch_bounds__(ANT) void assoc_scan_kernel(const double* __restrict__ app1, int64_t N1,
                                                         const double* __restrict__ app2, int64_t N2, int D,
                                                         const uint32_t* __restrict__ key2,
                                                         int64_t* __restrict__ count, int64_t* __restrict__ first,
                                                         int64_t seg_len) {
    uint32_t qk[R];
    int64_t qi[R];
    int64_t cnt[R];
    int64_t fst[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        qi[r] = ((int64_t)blockIdx.x * R + r) * ANT + threadIdx.x;
        qk[r] = (qi[r] < N1) ? low_word(app1[qi[r] * D]) : 0u;
        cnt[r] = 0;
        fst[r] = -1;
    }
    const uint4* __restrict__ keys4 = reinterpret_cast<const uint4*>(key2);
    const int64_t k_begin = SPLIT ? (int64_t)blockIdx.y * seg_len : 0;
    const int64_t k_end = SPLIT ? ((k_begin + seg_len < N2) ? k_begin + seg_len : N2) : N2;
    uint4 na = keys4[k_begin >> 2], nb = keys4[(k_begin >> 2) + 1];
    for (int64_t k = k_begin; k < k_end; k += 8) {
        const uint4 a = na, b = nb;
        // Scalar loads return out of order, so a wave can only wait for all of them: make the wait
        // for this step's keys happen here, before the next step's keys are requested, so that
        // request stays in flight during the compares.
        asm volatile("" ::"s"(a.x), "s"(a.w), "s"(b.x), "s"(b.w));
        na = keys4[(k >> 2) + 2]; // uniform address: s_load
        nb = keys4[(k >> 2) + 3];
        // Screening in vector registers only: acc = min over the step's keys of (query key XOR key),
        // which is zero exactly when one of them is equal.  Twelve vector instructions per eight
        // pairs (8 v_xor + 4 v_min3) and no lane masks: a compare per pair would need a scalar OR
        // per pair as well, and the CU's single scalar unit cannot keep up with four SIMDs of them.
        uint32_t any = 0xffffffffu;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t acc = umin3(qk[r] ^ a.x, qk[r] ^ a.y, qk[r] ^ a.z);
            acc = umin3(acc, qk[r] ^ a.w, qk[r] ^ b.x);
            acc = umin3(acc, qk[r] ^ b.y, qk[r] ^ b.z);
            any = umin3(any, acc, qk[r] ^ b.w);
        }
        const bool hit = (any == 0u);
        if (hit) {
            const uint32_t kk[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (qi[r] >= N1) continue;
                // an opaque copy of the query key: otherwise the compiler keeps all 8 R lane masks of
                // the screening step alive for reuse here and spills scalar registers in the hot loop
                uint32_t q = qk[r];
                asm volatile("" : "+v"(q));
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int64_t j = k + c;
                    if (j < N2 && q == kk[c] && rows_equal(app1 + qi[r] * D, app2 + j * D, D)) {
                        if (cnt[r] == 0) fst[r] = j;
                        cnt[r] += 1;
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (qi[r] < N1) {
            if (SPLIT) {
                if (cnt[r] > 0) {
                    atomicAdd(reinterpret_cast<unsigned long long*>(count + qi[r]), (unsigned long long)cnt[r]);
                    atomicMin(reinterpret_cast<unsigned long long*>(first + qi[r]), (unsigned long long)fst[r]);
                }
            } else {
                count[qi[r]] = cnt[r];
                first[qi[r]] = fst[r];
            }
        }
    }
}

// ---- hash join: the same pairs without the all-pairs scan --------------------------------------
// Equal rows have equal hashes, so each query row only needs the rows of set 2 in its hash bucket.
// Set 2 goes into an open-addressing table (linear probing, at most a quarter full); an entry is
// (32-bit tag of the hash) << 32 | (row + 1), 0 = empty.  A query walks its probe sequence to the
// first empty entry and runs the full comparison (rows_equal) on every entry whose tag matches, so
// the set of pairs is exactly that of the all-pairs scan; within a query row the matches are
// emitted in increasing set-2 row, which restores the reference's order (src/data_association.py:29-30)
// whatever order the rows were inserted in.
// The hash is taken over the descriptor's values, not its bytes: -0.0 and +0.0 compare equal and
// are hashed alike; a row with a NaN component equals nothing (not even itself) and is left out.
// Many equal rows make long probe sequences; a walk that exceeds HASH_MAX_PROBE entries, or a
// query row with more than HASH_MAX_DUP matches, raises a flag and the host falls back to the
// all-pairs scan, which has no such limit.
constexpr int HASH_MAX_PROBE = 2048;
constexpr int HASH_LOOK = 4; // table entries a counting walk fetches per step
constexpr int HASH_MAX_DUP = 32;

// Returns false for a row that contains NaN.
__device__ __forceinline__ bool row_hash(const double* __restrict__ row, int D, uint64_t* h_out) {
    uint64_t h = 0x243f6a8885a308d3ull;
    bool ok = true;
    for (int d = 0; d < D; ++d) {
        const double x = row[d];
        ok = ok && (x == x);
        const uint64_t bits = (x == 0.0) ? 0ull : (uint64_t)__double_as_longlong(x);
        h = (h ^ bits) * 0x9e3779b97f4a7c15ull;
        h ^= h >> 29;
    }
    h *= 0xbf58476d1ce4e5b9ull;
    h ^= h >> 32;
    *h_out = h ? h : 1ull; // 0 is kept free: it marks a row without a hash in hash_place_kernel's output
    return ok;
}

// Brings the workgroup's rows [row0, row0 + rows) of a row-major descriptor array into LDS with
// coalesced loads (consecutive lanes read consecutive 8-byte words; a lane reading its own row
// straight from memory would touch a different cache line in every lane, D times over).  Rows are
// DS = D | 1 doubles apart in LDS: an odd stride in doubles puts the 32 lanes of an LDS access
// group on different banks when each lane then walks its own row.  Ends with a barrier.
__device__ __forceinline__ void stage_rows(const double* __restrict__ app, int64_t row0, int rows, int D, int DS,
                                           double* lds) {
    const double* src = app + row0 * D;
    const int n = rows * D;
    if ((reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        // 16-byte loads, two consecutive words each (the second may belong to the next row)
        const double2* src2 = reinterpret_cast<const double2*>(src);
        const int step = 2 * ANT;
        int r = (2 * (int)threadIdx.x) / D, c = (2 * (int)threadIdx.x) % D;
        const int dr = step / D, dc = step % D;
        for (int e = 2 * threadIdx.x; e + 1 < n; e += step) {
            const double2 v = src2[e >> 1];
            lds[r * DS + c] = v.x;
            lds[(c + 1 < D) ? r * DS + c + 1 : (r + 1) * DS] = v.y;
            r += dr;
            c += dc;
            if (c >= D) {
                c -= D;
                r += 1;
            }
        }
        if ((n & 1) && threadIdx.x == 0) lds[(rows - 1) * DS + D - 1] = src[n - 1];
    } else {
        int r = threadIdx.x / D, c = threadIdx.x % D;
        const int dr = ANT / D, dc = ANT % D;
        for (int e = threadIdx.x; e < n; e += ANT) {
            lds[r * DS + c] = src[e];
            r += dr;
            c += dc;
            if (c >= D) {
                c -= D;
                r += 1;
            }
        }
    }
    __syncthreads();
}

// Clears the table (and the flag word behind it).  A kernel of the library's own rather than
// hipMemsetAsync: that call keeps the host busy for longer than the clearing takes on the device, and
// the device then waits for the launches behind it.  words is a multiple of 2; table is 16-byte aligned.
__global__ __launch_bounds__(ANT) void hash_clear_kernel(unsigned long long* __restrict__ table, int64_t words) {
    ulonglong2* t2 = reinterpret_cast<ulonglong2*>(table);
    const int64_t n2 = words / 2;
    for (int64_t k = (int64_t)blockIdx.x * ANT + threadIdx.x; k < n2; k += (int64_t)gridDim.x * ANT)
        t2[k] = ulonglong2{0ull, 0ull};
}

// The table is built in two launches, because atomic operations are what a one-launch build spends
// its time in (three quarters of it at 10^6 rows): compare-and-swap on a random entry costs several
// times a store to it.
// First launch: every row of set 2 writes its entry into its home slot with a plain 8-byte store --
// of the rows that share a home slot one survives, whichever store lands last -- and saves its hash.
// Second launch: a row finds its own entry in its home slot (nine rows in ten at a quarter load) or goes
// down the probe sequence with compare-and-swap from the next slot on.  The home slot is occupied
// either way, so a query's walk from the home slot to the first empty entry still meets every row
// with that hash.
__global__ __launch_bounds__(ANT) void hash_place_kernel(const double* __restrict__ app2, int64_t N2, int D,
                                                         unsigned long long* __restrict__ table, uint64_t mask,
                                                         uint64_t* __restrict__ hashes) {
    extern __shared__ double rows_lds[];
    const int DS = D | 1;
    const int64_t row0 = (int64_t)blockIdx.x * ANT;
    const int64_t left = N2 - row0;
    stage_rows(app2, row0, left < ANT ? (int)left : ANT, D, DS, rows_lds);
    const int64_t j = row0 + threadIdx.x;
    if (j >= N2) return;
    uint64_t h;
    if (!row_hash(rows_lds + threadIdx.x * DS, D, &h)) {
        hashes[j] = 0; // a row with a NaN equals nothing: it stays out of the table
        return;
    }
    hashes[j] = h;
    table[h & mask] = ((h >> 32) << 32) | (unsigned long long)(j + 1);
}

__global__ __launch_bounds__(ANT) void hash_settle_kernel(const uint64_t* __restrict__ hashes, int64_t N2,
                                                          unsigned long long* __restrict__ table, uint64_t mask,
                                                          int* __restrict__ flag, int* __restrict__ host_flag) {
    const int64_t j = (int64_t)blockIdx.x * ANT + threadIdx.x;
    if (j >= N2) return;
    const uint64_t h = hashes[j];
    if (h == 0) return;
    const unsigned long long entry = ((h >> 32) << 32) | (unsigned long long)(j + 1);
    uint64_t slot = h & mask;
    if (table[slot] == entry) return;
    // Look before swapping: the next HASH_LOOK entries are read together, and the swap is tried on the first
    // empty one only (an entry seen occupied stays occupied; one seen empty may have been taken since, then
    // the swap fails and the walk goes on behind it).
    slot = (slot + 1) & mask;
    for (int probe = 1; probe < HASH_MAX_PROBE;) {
        unsigned long long e[HASH_LOOK];
#pragma unroll
        for (int k = 0; k < HASH_LOOK; ++k)
            e[k] = __hip_atomic_load(&table[(slot + k) & mask], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int k = 0;
        while (k < HASH_LOOK && e[k] != 0ull) ++k;
        if (k < HASH_LOOK && atomicCAS(&table[(slot + k) & mask], 0ull, entry) == 0ull) return;
        const int passed = k < HASH_LOOK ? k + 1 : HASH_LOOK;
        probe += passed;
        slot = (slot + passed) & mask;
    }
    atomicExch(flag, 1);
    *host_flag = 1;
}

// rows_equal for a query row held in LDS against a row of set 2 in memory.  WIDE: the rows of set 2
// are read with 16-byte loads (D even and the array 16-byte aligned, so every row is), half as many
// requests to what is a random cache line per query.
template <bool WIDE>
__device__ __forceinline__ bool row_equals_staged(const double* q, const double* __restrict__ b, int D) {
    bool eq = true;
    if (WIDE) {
        const double2* b2 = reinterpret_cast<const double2*>(b);
        for (int d = 0; d < D / 2; ++d) {
            const double2 v = b2[d];
            eq = eq && (q[2 * d] == v.x) && (q[2 * d + 1] == v.y);
        }
    } else {
        for (int d = 0; d < D; ++d) eq = eq && (q[d] == b[d]);
    }
    return eq;
}

// Walks query row i's probe sequence; calls match(j) for every equal row j of set 2.
// Returns false when the walk was cut short.
template <typename F>
__device__ __forceinline__ bool hash_walk(const double* __restrict__ q, const double* __restrict__ app2, int D,
                                          const unsigned long long* __restrict__ table, uint64_t mask, uint64_t h,
                                          F&& match) {
    const unsigned long long tag = h >> 32;
    uint64_t slot = h & mask;
    for (int probe = 0; probe < HASH_MAX_PROBE; ++probe) {
        const unsigned long long e = table[slot];
        if (e == 0ull) return true;
        if ((e >> 32) == tag) {
            const int64_t j = (int64_t)(e & 0xffffffffull) - 1;
            if (rows_equal(q, app2 + j * D, D)) match(j);
        }
        slot = (slot + 1) & mask;
    }
    return false;
}

// The queries of a wave are verified against their candidate rows of set 2 together: a candidate row
// is a random 80-byte piece of memory, and a lane that reads its own candidate with D (or D/2) loads
// makes the wave request every one of 64 scattered cache lines D times over.  Here a row is read by a
// group of P consecutive lanes, one 8- or 16-byte part each (P = D, or D/2 when WIDE), so one load
// instruction fetches 64/P whole rows and every cache line is requested once; the parts' comparisons
// are combined with a ballot.  cand: this lane's candidate row or -1; q_wave: the wave's 64 query
// rows in LDS (row stride DS).  Returns whether this lane's candidate equals its query row.
template <bool WIDE>
__device__ __forceinline__ bool verify_candidates(int64_t cand, const double* q_wave, int DS,
                                                  const double* __restrict__ app2, int D) {
    const int lane = threadIdx.x & 63;
    const int P = WIDE ? D / 2 : D;  // lanes per row
    const int G = 64 / P;            // rows per load instruction
    const int g = lane / P, part = lane - g * P;
    const unsigned long long want = __ballot(cand >= 0);
    bool equal = false;
    // The loads of up to VERIFY_BATCH steps are issued before the first comparison: the steps are
    // independent random reads, and one memory round trip per step would make a wave with candidates
    // in all its lanes wait six round trips (D = 10) where one does.
    constexpr int VERIFY_BATCH = 8;
    for (int base0 = 0; base0 < 64; base0 += G * VERIFY_BATCH) {
        double2 v[VERIFY_BATCH];
        bool active[VERIFY_BATCH];
#pragma unroll
        for (int c = 0; c < VERIFY_BATCH; ++c) {
            const int base = base0 + c * G;
            active[c] = false;
            v[c] = double2{0.0, 0.0};
            if (base >= 64) continue;
            const unsigned long long chunk = (G >= 64) ? ~0ull : (((1ull << G) - 1ull) << base);
            if ((want & chunk) == 0ull) continue; // uniform: no candidate among these queries
            const int ql = base + g;              // the query this lane helps with
            const int src = ql < 64 ? ql : 63;
            const int64_t j = __shfl(cand, src, 64);
            active[c] = (g < G) && (ql < 64) && (j >= 0);
            if (active[c]) {
                if (WIDE)
                    v[c] = reinterpret_cast<const double2*>(app2 + j * D)[part];
                else
                    v[c].x = app2[j * D + part];
            }
        }
#pragma unroll
        for (int c = 0; c < VERIFY_BATCH; ++c) {
            const int base = base0 + c * G;
            if (base >= 64) continue;
            const unsigned long long chunk = (G >= 64) ? ~0ull : (((1ull << G) - 1ull) << base);
            if ((want & chunk) == 0ull) continue;
            const int ql = base + g;
            const int src = ql < 64 ? ql : 63;
            bool part_equal = true;
            if (active[c]) {
                const double* qrow = q_wave + src * DS;
                part_equal = WIDE ? ((qrow[2 * part] == v[c].x) && (qrow[2 * part + 1] == v[c].y)) : (qrow[part] == v[c].x);
            }
            // a query's row is equal when none of its P lanes saw a difference
            const unsigned long long differs = __ballot(active[c] && !part_equal);
            const int mine = lane - base; // this lane's place among the chunk's queries
            if (mine >= 0 && mine < G && cand >= 0) {
                const unsigned long long group = ((P >= 64) ? ~0ull : ((1ull << P) - 1ull)) << (mine * P);
                equal = (differs & group) == 0ull;
            }
        }
    }
    return equal;
}

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* lds /*ANT/64 + 1*/, int64_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    int64_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < ANT / 64; ++w) {
        if (w < wave) base += lds[w];
        tot += lds[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

template <bool WIDE>
__global__ __launch_bounds__(ANT) void hash_count_kernel(const double* __restrict__ app1, int64_t N1,
                                                         const double* __restrict__ app2, int D,
                                                         const unsigned long long* __restrict__ table, uint64_t mask,
                                                         uint64_t* __restrict__ found,
                                                         int64_t* __restrict__ blocksum, int* __restrict__ flag,
                                                         int* __restrict__ host_flag) {
    extern __shared__ double rows_lds[];
    __shared__ int64_t scan_lds[ANT / 64 + 1];
    const int DS = D | 1;
    const int64_t row0 = (int64_t)blockIdx.x * ANT;
    const int64_t left = N1 - row0;
    stage_rows(app1, row0, left < ANT ? (int)left : ANT, D, DS, rows_lds);
    const int64_t i = row0 + threadIdx.x;
    const double* q = rows_lds + threadIdx.x * DS;
    const double* q_wave = rows_lds + (threadIdx.x & ~63) * DS;
    int64_t cnt = 0, fst = -1;
    uint64_t h = 0;
    // every lane of the wave stays in the walk (the verification is a wave-wide step); lanes without
    // a query row, or whose row holds a NaN (equal to nothing), are finished from the start
    bool walking = (i < N1) && row_hash(q, D, &h);
    const unsigned long long tag = h >> 32;
    uint64_t slot = h & mask;
    int probe = 0;
    bool complete = true;
    // A walk is a chain of dependent random reads, and the wave stays in it until its longest walk has
    // ended (a handful of entries at a quarter load); HASH_LOOK entries are therefore fetched per step, all in
    // flight together (they mostly share a cache line), which cuts the chain to two or three steps.
    while (__ballot(walking) != 0ull) {
        unsigned long long e[HASH_LOOK];
        if (walking) {
#pragma unroll
            for (int k = 0; k < HASH_LOOK; ++k) e[k] = table[(slot + k) & mask];
        }
        int64_t cand[HASH_LOOK];
#pragma unroll
        for (int k = 0; k < HASH_LOOK; ++k) {
            cand[k] = -1;
            if (walking) {
                if (e[k] == 0ull) {
                    walking = false;
                } else {
                    if ((e[k] >> 32) == tag) cand[k] = (int64_t)(e[k] & 0xffffffffull) - 1;
                    if (++probe >= HASH_MAX_PROBE) {
                        walking = false;
                        complete = false;
                    }
                }
            }
        }
        slot = (slot + HASH_LOOK) & mask;
#pragma unroll
        for (int k = 0; k < HASH_LOOK; ++k) {
            if (__ballot(cand[k] >= 0) != 0ull) {
                if (verify_candidates<WIDE>(cand[k], q_wave, DS, app2, D)) {
                    fst = (cnt == 0 || cand[k] < fst) ? cand[k] : fst;
                    cnt += 1;
                }
            }
        }
    }
    if (i < N1) {
        if (!complete || cnt > HASH_MAX_DUP) { // the host repeats the association with the all-pairs scan
            atomicExch(flag, 1);
            *host_flag = 1;
        }
        // one word per query row: number of equal rows (at most HASH_MAX_PROBE) and the first of them
        found[i] = ((uint64_t)cnt << 32) | (uint64_t)(uint32_t)fst;
    } else {
        cnt = 0;
    }
    // the workgroup's number of pairs, for the prefix sum over workgroups (scan_block_offsets_kernel);
    // the emit kernel redoes the sum inside the workgroup from found[]
    int64_t total;
    (void)block_exclusive_scan(cnt, scan_lds, &total);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = total;
}

__global__ __launch_bounds__(ANT) void hash_emit_kernel(const double* __restrict__ app1, int64_t N1,
                                                        const double* __restrict__ app2, int D,
                                                        const unsigned long long* __restrict__ table, uint64_t mask,
                                                        const uint64_t* __restrict__ found,
                                                        const int64_t* __restrict__ blockstart, int64_t* __restrict__ idx1,
                                                        int64_t* __restrict__ idx2, int64_t capacity,
                                                        const int* __restrict__ flag) {
    __shared__ int64_t scan_lds[ANT / 64 + 1];
    // The count kernel gave up (long walks or very many equal rows): the host repeats the association with
    // the all-pairs scan, and the counts here may be huge or incomplete -- emit nothing.
    if (*flag != 0) return;
    const int64_t i = (int64_t)blockIdx.x * ANT + threadIdx.x;
    const uint64_t f = (i < N1) ? found[i] : 0ull;
    const int64_t c = (int64_t)(f >> 32);
    // this row's place in the output: the pairs of the workgroups before this one (blockstart), then those
    // of the rows before it in the workgroup
    int64_t total;
    const int64_t o = blockstart[blockIdx.x] + block_exclusive_scan(c, scan_lds, &total);
    if (c == 0) return;
    if (c == 1) {
        if (o < capacity) {
            idx1[o] = i;
            idx2[o] = (int64_t)(f & 0xffffffffull);
        }
        return;
    }
    // several equal rows (at most HASH_MAX_DUP, or the host would not have come here): emit them in
    // increasing row order by repeated selection of the next larger one
    uint64_t h;
    (void)row_hash(app1 + i * D, D, &h);
    int64_t last = -1;
    for (int64_t k = 0; k < c; ++k) {
        int64_t best = INT64_MAX;
        (void)hash_walk(app1 + i * D, app2, D, table, mask, h, [&](int64_t j) {
            if (j > last && j < best) best = j;
        });
        if (o + k < capacity) {
            idx1[o + k] = i;
            idx2[o + k] = best;
        }
        last = best;
    }
}

// ---- exclusive prefix sum of int64 counts ------------------------------------------------------
__global__ __launch_bounds__(ANT) void scan_block_sums_kernel(const int64_t* __restrict__ count, int64_t N,
                                                              int64_t* __restrict__ blocksum) {
    __shared__ int64_t lds[ANT / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < N) s += count[base + k];
    int64_t tot;
    (void)block_exclusive_scan(s, lds, &tot);
    if (threadIdx.x == 0) blocksum[blockIdx.x] = tot;
}

// One workgroup of 1024 threads turns blocksum[0..nb) into its exclusive scan and stores the grand
// total at [nb]: thread k takes a run of consecutive entries, the runs' sums are scanned across the
// workgroup, and every thread then writes its run.  host_total: optional host-visible word that
// receives the grand total as well.
constexpr int OFFT = 1024;
__global__ __launch_bounds__(OFFT) void scan_block_offsets_kernel(int64_t* __restrict__ blocksum, int64_t nb,
                                                                  int64_t* __restrict__ host_total) {
    __shared__ int64_t wave_sum[OFFT / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t per = (nb + OFFT - 1) / OFFT;
    const int64_t lo = (int64_t)threadIdx.x * per;
    const int64_t hi = lo + per < nb ? lo + per : nb;
    int64_t run = 0;
    for (int64_t k = lo; k < hi; ++k) run += blocksum[k];
    int64_t inc = run;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int64_t t = __shfl_up(inc, off, 64);
        if (lane >= off) inc += t;
    }
    if (lane == 63) wave_sum[wave] = inc;
    __syncthreads();
    int64_t base = 0, total = 0;
#pragma unroll
    for (int w = 0; w < OFFT / 64; ++w) {
        if (w < wave) base += wave_sum[w];
        total += wave_sum[w];
    }
    int64_t ex = base + inc - run;
    for (int64_t k = lo; k < hi; ++k) {
        const int64_t v = blocksum[k];
        blocksum[k] = ex;
        ex += v;
    }
    if (threadIdx.x == 0) {
        blocksum[nb] = total;
        if (host_total != nullptr) *host_total = total;
    }
}

__global__ __launch_bounds__(ANT) void scan_write_kernel(const int64_t* __restrict__ count, int64_t N,
                                                         const int64_t* __restrict__ blocksum, int64_t nb,
                                                         int64_t* __restrict__ offset) {
    __shared__ int64_t lds[ANT / 64 + 1];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    int64_t c[SCAN_ITEMS];
    int64_t s = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        c[k] = (base + k < N) ? count[base + k] : 0;
        s += c[k];
    }
    int64_t tot;
    int64_t ex = block_exclusive_scan(s, lds, &tot) + blocksum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < N) offset[base + k] = ex;
        ex += c[k];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) offset[N] = blocksum[nb];
}

// Emits the pairs in query-major order: row i writes its matches, in increasing j, starting at
// offset[i].  One match (the usual case) needs no second look at set 2.
__global__ __launch_bounds__(ANT) void assoc_emit_kernel(const double* __restrict__ app1, int64_t N1,
                                                         const double* __restrict__ app2, int64_t N2, int D,
                                                         const int64_t* __restrict__ count,
                                                         const int64_t* __restrict__ first,
                                                         const int64_t* __restrict__ offset,
                                                         int64_t* __restrict__ idx1, int64_t* __restrict__ idx2,
                                                         int64_t capacity) {
    const int64_t i = (int64_t)blockIdx.x * ANT + threadIdx.x;
    if (i >= N1) return;
    const int64_t c = count[i];
    if (c == 0) return;
    int64_t o = offset[i];
    int64_t j = first[i];
    if (o < capacity) {
        idx1[o] = i;
        idx2[o] = j;
    }
    int64_t left = c - 1;
    for (++j, ++o; left > 0 && j < N2; ++j) {
        if (rows_equal(app1 + i * D, app2 + j * D, D)) {
            if (o < capacity) {
                idx1[o] = i;
                idx2[o] = j;
            }
            ++o;
            --left;
        }
    }
}

// ---- projection of a point set and pair gather -------------------------------------------------
__global__ __launch_bounds__(ANT) void project_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                      const double* __restrict__ z, int64_t N, Pose12 T, CameraK cam,
                                                      uint8_t* __restrict__ inside, double* __restrict__ uu,
                                                      double* __restrict__ vv) {
    const int64_t i = (int64_t)blockIdx.x * ANT + threadIdx.x;
    if (i >= N) return;
    double px = x[i], py = y[i], pz = z[i];
    homogeneous_nan(px, py, pz);
    const Projected p = project_point<false>(T.m, cam, px, py, pz);
    inside[i] = p.inside ? 1 : 0;
    uu[i] = p.u;
    vv[i] = p.v;
}

__global__ __launch_bounds__(ANT) void gather_pairs_kernel(const int64_t* __restrict__ idx1,
                                                           const int64_t* __restrict__ idx2, int64_t n,
                                                           const double* __restrict__ pos1,
                                                           const double* __restrict__ pos2, double* __restrict__ u,
                                                           double* __restrict__ v, double* __restrict__ x,
                                                           double* __restrict__ y, double* __restrict__ z) {
    const int64_t k = (int64_t)blockIdx.x * ANT + threadIdx.x;
    if (k >= n) return;
    const int64_t i = idx1[k], j = idx2[k];
    u[k] = pos1[i * 2 + 0];
    v[k] = pos1[i * 2 + 1];
    x[k] = pos2[j * 3 + 0];
    y[k] = pos2[j * 3 + 1];
    z[k] = pos2[j * 3 + 2];
}

static int ensure_assoc_workspace(picp_ctx* ctx, int64_t N1, int64_t N2) {
    if (N1 > ctx->assoc_cap) {
        if (ctx->assoc_count) (void)hipFree(ctx->assoc_count);
        if (ctx->assoc_first) (void)hipFree(ctx->assoc_first);
        if (ctx->assoc_offset) (void)hipFree(ctx->assoc_offset);
        if (ctx->assoc_blocksum) (void)hipFree(ctx->assoc_blocksum);
        ctx->assoc_count = ctx->assoc_first = ctx->assoc_offset = ctx->assoc_blocksum = nullptr;
        ctx->assoc_cap = 0;
        const int64_t cap = N1 + N1 / 4 + 1024;
        const int64_t nb = (cap + ANT - 1) / ANT; // one number per workgroup of the hash join's count kernel
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_count, sizeof(int64_t) * cap));
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_first, sizeof(int64_t) * cap));
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_offset, sizeof(int64_t) * (cap + 1)));
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_blocksum, sizeof(int64_t) * (nb + 1)));
        ctx->assoc_cap = cap;
    }
    if (N2 + 16 > ctx->assoc_key_cap) { // the scan prefetches up to 15 keys past the last one
        if (ctx->assoc_key) (void)hipFree(ctx->assoc_key);
        ctx->assoc_key = nullptr;
        ctx->assoc_key_cap = 0;
        const int64_t cap = N2 + N2 / 4 + 1024;
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_key, sizeof(uint32_t) * cap));
        ctx->assoc_key_cap = cap;
    }
    return PICP_OK;
}

// words: table entries plus the control words behind them; rows2: rows of set 2 (one saved hash each).
static int ensure_hash_table(picp_ctx* ctx, int64_t words, int64_t rows2) {
    if (words > ctx->assoc_table_cap) {
        if (ctx->assoc_table) (void)hipFree(ctx->assoc_table);
        ctx->assoc_table = nullptr;
        ctx->assoc_table_cap = 0;
        const int64_t cap = words + words / 8;
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_table, sizeof(unsigned long long) * cap));
        ctx->assoc_table_cap = cap;
        ctx->assoc_table_clean = 0;
    }
    if (rows2 > ctx->assoc_hash_cap) {
        if (ctx->assoc_hash) (void)hipFree(ctx->assoc_hash);
        ctx->assoc_hash = nullptr;
        ctx->assoc_hash_cap = 0;
        const int64_t cap = rows2 + rows2 / 4 + 1024;
        PICP_HIP_CHECK(hipMalloc(&ctx->assoc_hash, sizeof(uint64_t) * cap));
        ctx->assoc_hash_cap = cap;
    }
    return PICP_OK;
}

} // namespace picp

using namespace picp;

extern "C" {

int picp_set_association_mode(picp_ctx* ctx, int32_t mode) {
    PICP_REQUIRE(ctx != nullptr, "picp_set_association_mode: null ctx");
    PICP_REQUIRE(mode >= 0 && mode <= 2, "picp_set_association_mode: mode %d is not 0 (auto), 1 (all pairs) or 2 (hash join)",
                 (int)mode);
    ctx->assoc_mode = mode;
    return PICP_OK;
}

// The hash join: table build (two launches), count, prefix sum over the workgroups' numbers of pairs,
// emit; then one wait for the number of pairs and the fallback flag, which the kernels leave in
// host-visible memory.  *retry is set when the join met more equal rows than it handles and the
// caller should run the all-pairs scan.
static int hash_join_pass(picp_ctx* ctx, const double* app1, int64_t N1, const double* app2, int64_t N2, int D,
                          int64_t* idx1, int64_t* idx2, int64_t capacity, int64_t* count, bool* retry) {
    hipStream_t s = ctx->stream;
    *retry = false;
    const unsigned grid1 = (unsigned)((N1 + ANT - 1) / ANT), grid2 = (unsigned)((N2 + ANT - 1) / ANT);
    uint64_t entries = 1024;
    while (entries < 4 * (uint64_t)N2) entries <<= 1; // at most a quarter full

    if (int rc = ensure_hash_table(ctx, (int64_t)entries + 2, N2)) return rc; // + the flag word, and one to keep the count even
    const uint64_t mask = entries - 1;
    int* flag = reinterpret_cast<int*>(ctx->assoc_table + entries);
    int64_t* h = reinterpret_cast<int64_t*>(ctx->pinned);
    h[0] = 0;
    h[1] = 0;
    int* host_flag = reinterpret_cast<int*>(h + 1);
    // The table is cleared behind the join that used it, after the point the host waits for, so the next join
    // finds it clean and the clearing runs while the host turns around; only a first or larger table is
    // cleared here.
    if (ctx->assoc_table_clean < (int64_t)entries + 2)
        hipLaunchKernelGGL(hash_clear_kernel, dim3((unsigned)(8 * ctx->num_cus)), dim3(ANT), 0, s, ctx->assoc_table,
                           (int64_t)entries + 2);
    ctx->assoc_table_clean = 0;
    const unsigned stage_bytes = (unsigned)(sizeof(double) * ANT * (size_t)(D | 1)); // the workgroup's rows in LDS
    hipLaunchKernelGGL(hash_place_kernel, dim3(grid2), dim3(ANT), stage_bytes, s, app2, N2, D, ctx->assoc_table, mask,
                       ctx->assoc_hash);
    hipLaunchKernelGGL(hash_settle_kernel, dim3(grid2), dim3(ANT), 0, s, (const uint64_t*)ctx->assoc_hash, N2,
                       ctx->assoc_table, mask, flag, host_flag);
    const bool wide = (D % 2 == 0) && ((reinterpret_cast<uintptr_t>(app2) & 15u) == 0);
    auto count_kernel = wide ? hash_count_kernel<true> : hash_count_kernel<false>;
    hipLaunchKernelGGL(count_kernel, dim3(grid1), dim3(ANT), stage_bytes, s, app1, N1, app2, D,
                       (const unsigned long long*)ctx->assoc_table, mask, reinterpret_cast<uint64_t*>(ctx->assoc_count),
                       ctx->assoc_blocksum, flag, host_flag);
    hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(OFFT), 0, s, ctx->assoc_blocksum, (int64_t)grid1, h);
    PICP_HIP_CHECK(hipGetLastError());
    // The pairs are emitted before the host knows their number: the emit kernel writes nothing beyond
    // `capacity`, so one wait at the end serves the count, the fallback flag and the index lists (a second
    // wait between counting and emitting costs more than an emit whose output is then thrown away).
    if (capacity > 0) {
        hipLaunchKernelGGL(hash_emit_kernel, dim3(grid1), dim3(ANT), 0, s, app1, N1, app2, D,
                           (const unsigned long long*)ctx->assoc_table, mask,
                           reinterpret_cast<const uint64_t*>(ctx->assoc_count), (const int64_t*)ctx->assoc_blocksum, idx1,
                           idx2, capacity, (const int*)flag);
        PICP_HIP_CHECK(hipGetLastError());
    }
    PICP_HIP_CHECK(hipEventRecord(ctx->assoc_done, s));
    hipLaunchKernelGGL(hash_clear_kernel, dim3((unsigned)(8 * ctx->num_cus)), dim3(ANT), 0, s, ctx->assoc_table,
                       (int64_t)entries + 2);
    PICP_HIP_CHECK(hipGetLastError());
    ctx->assoc_table_clean = (int64_t)entries + 2;
    PICP_HIP_CHECK(hipEventSynchronize(ctx->assoc_done));
    if (h[1] != 0) {
        *retry = true;
        return PICP_OK;
    }
    *count = h[0];
    if (h[0] > capacity) {
        set_error("picp_associate: %lld pairs found, index arrays hold %lld", (long long)h[0], (long long)capacity);
        return PICP_ERR_CAPACITY;
    }
    return PICP_OK;
}

// The all-pairs scan: counts, prefix sum, emit.
static int scan_pass(picp_ctx* ctx, const double* app1, int64_t N1, const double* app2, int64_t N2, int D,
                     int64_t* idx1, int64_t* idx2, int64_t capacity, int64_t* count) {
    hipStream_t s = ctx->stream;
    const unsigned grid1 = (unsigned)((N1 + ANT - 1) / ANT);
    // Many query rows: eight per lane amortise the key fetches and the loop overhead, and a workgroup per
    // 2 048 rows fills the chip.  Fewer (a rank's share of a sharded association, a frame's few hundred
    // points): the rows of set 2 are cut into segments as well, so that rows x segments fills it.
    int rows_per_lane = (N1 >= (int64_t)ANT * 8 * 8) ? 8 : 1;
    if (const char* e = getenv("PICP_ASSOC_ROWS_PER_LANE")) rows_per_lane = atoi(e); // tuning override
    const int R = rows_per_lane == 8 ? 8 : (rows_per_lane == 4 ? 4 : 1);
    const unsigned nbs = (unsigned)((N1 + (int64_t)ANT * R - 1) / ((int64_t)ANT * R));
    constexpr int64_t MIN_SEGMENT = 8192; // rows of set 2 per segment, at least
    int64_t segments = 1;
    // (swept at 10^6 rows of set 2: 10^6 query rows 57 ms unsplit, 45 with 4 segments, 41 with 16, 40 with 64, which is
    // the vector units' issue rate for the screening's 1.5 instructions per pair; 125 000 query rows 40 / 9.8 / 6.2 / 5.4 ms)
    if (nbs < 48u * (unsigned)ctx->num_cus) {
        segments = (48 * (int64_t)ctx->num_cus + nbs - 1) / nbs;
        const int64_t most = (N2 + MIN_SEGMENT - 1) / MIN_SEGMENT;
        if (segments > most) segments = most;
        if (segments < 1) segments = 1;
    }
    if (const char* e = getenv("PICP_ASSOC_SEGMENTS")) segments = atoll(e) > 0 ? atoll(e) : 1; // tuning override
    const int64_t seg_len = ((N2 + segments - 1) / segments + 7) / 8 * 8;
    segments = seg_len > 0 ? (N2 + seg_len - 1) / seg_len : 1;
    const bool split = segments > 1;
    const int64_t n_keys_grid = (split && N1 > N2) ? N1 : N2;
    hipLaunchKernelGGL(assoc_keys_kernel, dim3((unsigned)((n_keys_grid + ANT - 1) / ANT)), dim3(ANT), 0, s, app2, N2, D,
                       ctx->assoc_key, ctx->assoc_count, ctx->assoc_first, split ? N1 : (int64_t)0);
    if (split) {
        auto kern = R == 8 ? assoc_scan_kernel<8, true> : (R == 4 ? assoc_scan_kernel<4, true> : assoc_scan_kernel<1, true>);
        hipLaunchKernelGGL(kern, dim3(nbs, (unsigned)segments), dim3(ANT), 0, s, app1, N1, app2, N2, D,
                           (const uint32_t*)ctx->assoc_key, ctx->assoc_count, ctx->assoc_first, seg_len);
    } else {
        auto kern = R == 8 ? assoc_scan_kernel<8> : (R == 4 ? assoc_scan_kernel<4> : assoc_scan_kernel<1>);
        hipLaunchKernelGGL(kern, dim3(nbs), dim3(ANT), 0, s, app1, N1, app2, N2, D, (const uint32_t*)ctx->assoc_key,
                           ctx->assoc_count, ctx->assoc_first, (int64_t)0);
    }
    PICP_HIP_CHECK(hipGetLastError());
    const int64_t nb = (N1 + SCAN_BLOCK - 1) / SCAN_BLOCK;
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(ANT), 0, s, (const int64_t*)ctx->assoc_count,
                       N1, ctx->assoc_blocksum);
    hipLaunchKernelGGL(scan_block_offsets_kernel, dim3(1), dim3(OFFT), 0, s, ctx->assoc_blocksum, nb, (int64_t*)nullptr);
    hipLaunchKernelGGL(scan_write_kernel, dim3((unsigned)nb), dim3(ANT), 0, s, (const int64_t*)ctx->assoc_count, N1,
                       (const int64_t*)ctx->assoc_blocksum, nb, ctx->assoc_offset);
    PICP_HIP_CHECK(hipGetLastError());
    // The pairs are emitted before the host knows their number: the emit kernel writes nothing beyond
    // `capacity`, so one wait at the end serves the count and the index lists (a second wait between
    // counting and emitting costs more than an emit whose output is then thrown away).
    if (capacity > 0) {
        hipLaunchKernelGGL(assoc_emit_kernel, dim3(grid1), dim3(ANT), 0, s, app1, N1, app2, N2, D,
                           (const int64_t*)ctx->assoc_count, (const int64_t*)ctx->assoc_first,
                           (const int64_t*)ctx->assoc_offset, idx1, idx2, capacity);
        PICP_HIP_CHECK(hipGetLastError());
    }
    int64_t* h = reinterpret_cast<int64_t*>(ctx->pinned);
    PICP_HIP_CHECK(hipMemcpyAsync(h, ctx->assoc_offset + N1, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    PICP_HIP_CHECK(hipStreamSynchronize(s));
    *count = h[0];
    if (h[0] > capacity) {
        set_error("picp_associate: %lld pairs found, index arrays hold %lld", (long long)h[0], (long long)capacity);
        return PICP_ERR_CAPACITY;
    }
    return PICP_OK;
}

int picp_associate(picp_ctx* ctx, const double* app1, int64_t N1, const double* app2, int64_t N2, int32_t D,
                   int64_t* idx1, int64_t* idx2, int64_t capacity, int64_t* count) {
    PICP_REQUIRE(ctx && count, "picp_associate: null argument");
    PICP_REQUIRE(N1 >= 0 && N2 >= 0 && D >= 1 && capacity >= 0, "picp_associate: bad sizes N1=%lld N2=%lld D=%d",
                 (long long)N1, (long long)N2, D);
    PICP_REQUIRE((N1 == 0 || app1) && (N2 == 0 || app2), "picp_associate: null descriptor array");
    PICP_REQUIRE(capacity == 0 || (idx1 && idx2), "picp_associate: null index array");
    *count = 0;
    if (N1 == 0 || N2 == 0) return PICP_OK;
    DeviceGuard g(ctx->device);
    if (int rc = ensure_assoc_workspace(ctx, N1, N2)) return rc;
    // Hash join when the all-pairs scan would be the slower of the two (it screens about 1.7e13
    // pairs per second; the join costs a few tens of microseconds of launches plus ~1 ns per row).
    const bool fits = N2 < (int64_t)0xfffffffe; // table entries hold 32-bit row numbers
    PICP_REQUIRE(!(ctx->assoc_mode == 2 && !fits), "picp_associate: the hash join holds 32-bit row numbers");
    const bool stageable = (size_t)ANT * (size_t)(D | 1) * sizeof(double) <= 48 * 1024; // a workgroup's rows in LDS (D <= 23)
    if (ctx->assoc_mode == 2 && !stageable) ctx->assoc_hash_fallbacks += 1; // wider descriptors: the scan has no limit
    const bool hashed = stageable && (ctx->assoc_mode == 2 ||
                                      (ctx->assoc_mode == 0 && fits && (double)N1 * (double)N2 >= 4.0e9));
    if (hashed) {
        bool retry = false;
        if (int rc = hash_join_pass(ctx, app1, N1, app2, N2, (int)D, idx1, idx2, capacity, count, &retry)) return rc;
        if (!retry) return PICP_OK;
        ctx->assoc_hash_fallbacks += 1; // many equal rows: the all-pairs scan has no limit on them
    }
    return scan_pass(ctx, app1, N1, app2, N2, (int)D, idx1, idx2, capacity, count);
}

int picp_association_fallbacks(picp_ctx* ctx) { return ctx ? ctx->assoc_hash_fallbacks : 0; }

int picp_project(picp_ctx* ctx, const double* x, const double* y, const double* z, int64_t N,
                 const double* c_T_w_host, const picp_camera* cam, uint8_t* inside, double* uv_u, double* uv_v) {
    PICP_REQUIRE(ctx && c_T_w_host && cam, "picp_project: null argument");
    PICP_REQUIRE(N >= 0, "picp_project: negative count");
    PICP_REQUIRE(c_T_w_host[12] == 0.0 && c_T_w_host[13] == 0.0 && c_T_w_host[14] == 0.0 && c_T_w_host[15] == 1.0,
                 "picp_project: c_T_w last row must be (0,0,0,1)");
    if (N == 0) return PICP_OK;
    PICP_REQUIRE(x && y && z && inside && uv_u && uv_v, "picp_project: null array");
    DeviceGuard g(ctx->device);
    Pose12 T;
    for (int i = 0; i < 12; ++i) T.m[i] = c_T_w_host[i];
    CameraK c;
    for (int i = 0; i < 9; ++i) c.k[i] = cam->K[i];
    c.width = cam->width;
    c.height = cam->height;
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)((N + ANT - 1) / ANT)), dim3(ANT), 0, ctx->stream, x, y, z, N, T,
                       c, inside, uv_u, uv_v);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

int picp_gather_pairs(picp_ctx* ctx, const int64_t* idx1, const int64_t* idx2, int64_t n, const double* pos1,
                      const double* pos2, double* u, double* v, double* x, double* y, double* z) {
    PICP_REQUIRE(ctx != nullptr, "picp_gather_pairs: null ctx");
    PICP_REQUIRE(n >= 0, "picp_gather_pairs: negative count");
    if (n == 0) return PICP_OK;
    PICP_REQUIRE(idx1 && idx2 && pos1 && pos2 && u && v && x && y && z, "picp_gather_pairs: null array");
    DeviceGuard g(ctx->device);
    hipLaunchKernelGGL(gather_pairs_kernel, dim3((unsigned)((n + ANT - 1) / ANT)), dim3(ANT), 0, ctx->stream, idx1,
                       idx2, n, pos1, pos2, u, v, x, y, z);
    PICP_HIP_CHECK(hipGetLastError());
    return PICP_OK;
}

} // extern "C"
