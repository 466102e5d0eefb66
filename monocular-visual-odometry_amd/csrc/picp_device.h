// Device-side math shared by every kernel of libpicp_hip.so (gfx950 only).
//
// All arithmetic is float64.  The functions restate, per correspondence and per iteration,
// what the reference computes in numpy; each one names the reference lines it follows
// (paths relative to the reference checkout).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/picp_hip.h"

namespace picp {

// ---- layout of the device state block (PICP_STATE_DOUBLES doubles) -------------------------
// The loop's scalar state (src/ProjectiveICP.py:157-171) lives in device memory so iterations
// chain without a host round trip.
enum : int {
    ST_W_T_C = 0,        // [16] current pose w_T_c0
    ST_C_T_W = 16,       // [16] its inverse = the camera state c_T_w (:232)
    ST_THR = 32,         // kernel_threshold
    ST_LAMBDA = 33,      // dumping_factor
    ST_ERR_PREV = 34,    // error_prev
    ST_MEAN_SLOPE = 35,  // error_mean_slope
    ST_SIGMA_SLOPE = 36, // error_sigma_slope
    ST_STUCK = 37,       // stuck_counter
    ST_FLICKER = 38,     // flickering_counter
    ST_IT = 39,          // iterations completed (icp_iteration)
    ST_STOP = 40,        // 1 once the stop test fired or num_iterations was reached
    ST_RING = 41,        // [10] error_slope_ring_buffer
    ST_BEST_ERR = 51,    // running minimum of the error list (first minimum wins, :249)
    ST_BEST_IT = 52,     // its index
    ST_MODE = 53,        // 0: stage 2 sums the mean-filtered set; 1: it re-sums the stage-1 set (see S2_*)
    ST_N_STAGE2 = 54,    // mean-filter survivors found before a switch to mode 1 (for the history row)
    ST_XEPOCH = 55,      // sharded single-launch loop: exchanges between the GPUs completed so far (all launches)
    ST_BEST_T = 64,      // [16] pose of that iteration (:254)
    ST_START_T = 80,     // [16] pose on entry, returned when the result is invalid (:251)
};

// stage-1 record
enum : int { S1_SUM_CHI = 0, S1_COUNT = 1, S1_N_INSIDE = 2, S1_PAD = 3 };
// stage-2 record: H, b, sum of chi and member count of ONE candidate inlier set.
//   mode 0: F = {chi <= threshold and chi < mean}   (the stage-2 set, src/ProjectiveICP.py:302)
//   mode 1: A = {chi <= threshold}                  (the stage-1 set, :294)
// The reference uses F when it has >= min_inliers members, else A (:307-310).  F almost always
// qualifies, so the scan sums only F; when the solve step finds F too small it sets the state's
// mode flag and does not advance, and the next round of stage launches re-sums A for the same pose.
// (Summing both sets in every scan costs 40 more fused multiply-adds per correspondence and twice
// the accumulator registers, for a result that is used only in degenerate iterations.)
enum : int {
    S2_H = 0,    // [21] upper triangle of sum J^T J, row-major (00,01,..,05,11,..,55)
    S2_B = 21,   // [6]  sum J^T e
    S2_CHI = 27, // sum of chi over the set
    S2_N = 28,   // members
    S2_USED = 29, // number of meaningful entries; the record is padded to PICP_S2_DOUBLES
};

struct CameraK {
    double k[9];
    double width, height;
};
struct Pose12 { // rows 0..2 of a 4x4 pose, passed to kernels by value
    double m[12];
};

// True when K = [[fx,0,cx],[0,fy,cy],[0,0,k8]]: the shape of every pinhole camera matrix and of
// the bundled one (data/camera.dat:2-4).  The kernels then skip the products with the zero
// entries; since 0*x + y == y for finite x, the results are the same numbers.
inline bool is_pinhole(const double* K) { return K[1] == 0.0 && K[3] == 0.0 && K[6] == 0.0 && K[7] == 0.0; }

__device__ __forceinline__ double d_inf() { return __longlong_as_double(0x7ff0000000000000LL); }

// a / b given r = 1.0 / b (correctly rounded): one multiply and a fused residual correction
// (q + (a - q b) r).  With a correctly rounded reciprocal this returns the correctly rounded
// quotient while everything stays far from overflow/underflow; the caller checks that and
// otherwise divides.
__device__ __forceinline__ double div_with_recip(double a, double b, double r) {
    const double q = a * r;
    const double rem = __builtin_fma(-q, b, a);
    return __builtin_fma(rem, r, q);
}

// 1.0 / b for a denominator of ordinary magnitude (about 1e-150 < |b| < 1e150, which the caller checks
// afterwards): the hardware seed, two Newton steps and the final residual step -- the sequence a
// float64 division compiles to, without its scaling of extreme operands and its fix-up of zeros,
// infinities and NaN, which change nothing in this range.  Same bits as 1.0 / b there, five
// instructions fewer.
__device__ __forceinline__ double recip_ordinary(double b) {
    double r = __builtin_amdgcn_rcp(b);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-b, r, 1.0), r);
    return __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
}

// ---- Camera.project_point (src/Camera.py:104-119) -------------------------------------------
// T = c_T_w rows 0..2 (12 doubles).  The reference also forms the homogeneous w and divides by
// it; every pose it builds has last row (0,0,0,1), so for finite points w == 1 exactly and the
// division is the identity (non-finite points: homogeneous_nan below, applied where the points
// are loaded).  Order follows the reference: both pixel divisions happen before any test, then
// "z <= 0 -> outside", then the four image-bound tests.  Tests are written as negations of the
// reference's reject conditions so that NaN behaves as in Python (a NaN coordinate rejects
// nothing).
// The reference's homogeneous coordinate w = 0 x + 0 y + 0 z + 1 (src/Camera.py:104-106, pose last
// row (0, 0, 0, 1)) is NaN as soon as one coordinate is infinite or NaN (0 * inf), and the division
// by it then turns the whole camera point into NaN, whatever the pose.  A point with all three
// coordinates NaN goes through the transform the same way, so such points are replaced by
// (NaN, NaN, NaN) when they are loaded; finite points are left as they are.
__device__ __forceinline__ void homogeneous_nan(double& x, double& y, double& z) {
#pragma clang fp contract(off)
    const double t = __builtin_fma(0.0, z, __builtin_fma(0.0, y, 0.0 * x)); // 0 or NaN
    const bool bad = (t != t);
    x = bad ? t : x;
    y = bad ? t : y;
    z = bad ? t : z;
}

struct Projected {
    double pcx, pcy, pcz; // point in camera coordinates
    double h0, h1, h2;    // K @ pc
    double zi;            // 1 / h2
    double u, v;
    bool inside;
};

// The arithmetic is spelled out (fused multiply-adds in a fixed order, no compiler contraction):
// every kernel must get the same bits for the same point, because one kernel's chi is compared
// with a mean formed from another kernel's chi of the same points (src/ProjectiveICP.py:302).
template <bool PINHOLE>
__device__ __forceinline__ Projected project_point(const double* __restrict__ T, const CameraK& cam, double x,
                                                   double y, double z) {
#pragma clang fp contract(off)
    Projected p;
    p.pcx = __builtin_fma(T[2], z, __builtin_fma(T[1], y, T[0] * x)) + T[3];
    p.pcy = __builtin_fma(T[6], z, __builtin_fma(T[5], y, T[4] * x)) + T[7];
    p.pcz = __builtin_fma(T[10], z, __builtin_fma(T[9], y, T[8] * x)) + T[11];
    if (PINHOLE) {
        p.h0 = __builtin_fma(cam.k[2], p.pcz, cam.k[0] * p.pcx);
        p.h1 = __builtin_fma(cam.k[5], p.pcz, cam.k[4] * p.pcy);
        p.h2 = cam.k[8] * p.pcz;
    } else {
        p.h0 = __builtin_fma(cam.k[2], p.pcz, __builtin_fma(cam.k[1], p.pcy, cam.k[0] * p.pcx));
        p.h1 = __builtin_fma(cam.k[5], p.pcz, __builtin_fma(cam.k[4], p.pcy, cam.k[3] * p.pcx));
        p.h2 = __builtin_fma(cam.k[8], p.pcz, __builtin_fma(cam.k[7], p.pcy, cam.k[6] * p.pcx));
    }
    p.zi = recip_ordinary(p.h2); // also z_inv of the Jacobian (src/ProjectiveICP.py:345); range checked below
    p.u = div_with_recip(p.h0, p.h2, p.zi);
    p.v = div_with_recip(p.h1, p.h2, p.zi);
    const double big = 1e150, small = 1e-150;
    const bool safe = (fabs(p.h2) > small) && (fabs(p.h2) < big) && (fabs(p.u) + fabs(p.v) < big);
    if (!safe) {
        // Extreme magnitudes, infinities or NaN (never the case for real map points): the full
        // products with K (0 * inf is NaN, so K's zero entries matter here) and plain IEEE divisions.
        p.h0 = __builtin_fma(cam.k[2], p.pcz, __builtin_fma(cam.k[1], p.pcy, cam.k[0] * p.pcx));
        p.h1 = __builtin_fma(cam.k[5], p.pcz, __builtin_fma(cam.k[4], p.pcy, cam.k[3] * p.pcx));
        p.h2 = __builtin_fma(cam.k[8], p.pcz, __builtin_fma(cam.k[7], p.pcy, cam.k[6] * p.pcx));
        p.zi = 1.0 / p.h2;
        p.u = p.h0 / p.h2;
        p.v = p.h1 / p.h2;
    }
    const bool behind = (p.pcz <= 0.0);
    const bool outside = (p.u < 0.0) | (p.u >= cam.width) | (p.v < 0.0) | (p.v >= cam.height); // no short circuit: no branches
    p.inside = !behind && !outside;
    return p;
}

// chi = e^T e (src/ProjectiveICP.py:293).  numpy evaluates the 2-vector product as
// fma(e1, e1, e0 * e0) (its BLAS dot kernel accumulates with fused multiply-adds; checked against
// `e.T @ e` on the build host); the same expression here, so that a chi that sits exactly on the
// threshold or on the mean falls on the same side as in the reference.
__device__ __forceinline__ double chi_of(double e0, double e1) {
#pragma clang fp contract(off)
    return __builtin_fma(e1, e1, e0 * e0);
}

// ---- __error_and_jacobian (src/ProjectiveICP.py:326-353) ------------------------------------
// J = J_proj @ K @ J_icp with J_icp = [I | -skew(p_hat)] (:341-343) and
// J_proj = [[1/z, 0, -x/z^2], [0, 1/z, -y/z^2]] on p_hat_cam = K p_hat (:345-349).
// p_hat is the point in camera coordinates (:336-337) -- the same transform as the projection,
// because the camera state is inv(w_T_c0) (:232).
// With a pinhole K the entries J0[1] and J1[0] are exactly zero and are not formed.
template <bool PINHOLE>
__device__ __forceinline__ void jacobian_rows(const Projected& p, const CameraK& cam, double (&J0)[6],
                                              double (&J1)[6]) {
    const double zi = p.zi;        // :345
    const double zi2 = zi * zi;    // :346
    const double m0 = -p.h0 * zi2; // :348
    const double m1 = -p.h1 * zi2; // :349
    // A = J_proj @ K  (2x3);  J = A @ [I | -skew(p)],  -skew(p) = [[0, pz, -py], [-pz, 0, px], [py, -px, 0]]
    if (PINHOLE) {
        const double a00 = zi * cam.k[0];
        const double a02 = zi * cam.k[2] + m0 * cam.k[8];
        const double a11 = zi * cam.k[4];
        const double a12 = zi * cam.k[5] + m1 * cam.k[8];
        J0[0] = a00;
        J0[1] = 0.0;
        J0[2] = a02;
        J0[3] = a02 * p.pcy;
        J0[4] = a00 * p.pcz - a02 * p.pcx;
        J0[5] = -(a00 * p.pcy);
        J1[0] = 0.0;
        J1[1] = a11;
        J1[2] = a12;
        J1[3] = a12 * p.pcy - a11 * p.pcz;
        J1[4] = -(a12 * p.pcx);
        J1[5] = a11 * p.pcx;
    } else {
        const double a00 = zi * cam.k[0] + m0 * cam.k[6];
        const double a01 = zi * cam.k[1] + m0 * cam.k[7];
        const double a02 = zi * cam.k[2] + m0 * cam.k[8];
        const double a10 = zi * cam.k[3] + m1 * cam.k[6];
        const double a11 = zi * cam.k[4] + m1 * cam.k[7];
        const double a12 = zi * cam.k[5] + m1 * cam.k[8];
        J0[0] = a00;
        J0[1] = a01;
        J0[2] = a02;
        J0[3] = a02 * p.pcy - a01 * p.pcz;
        J0[4] = a00 * p.pcz - a02 * p.pcx;
        J0[5] = a01 * p.pcx - a00 * p.pcy;
        J1[0] = a10;
        J1[1] = a11;
        J1[2] = a12;
        J1[3] = a12 * p.pcy - a11 * p.pcz;
        J1[4] = a10 * p.pcz - a12 * p.pcx;
        J1[5] = a11 * p.pcx - a10 * p.pcy;
    }
}

// Sums of one candidate inlier set: upper triangle of H and b.
struct SetSums {
    double H[21];
    double b[6];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int i = 0; i < 21; ++i) H[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) b[i] = 0.0;
    }
    // H += J^T J, b += J^T e  (src/ProjectiveICP.py:316-317).  With a pinhole K, J0[1] and J1[0]
    // are zero: their products are skipped (40 fused multiply-adds instead of 54).
    template <bool PINHOLE>
    __device__ __forceinline__ void add(const double (&J0)[6], const double (&J1)[6], double e0, double e1) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
#pragma unroll
            for (int j = i; j < 6; ++j) {
                const bool row0 = !(PINHOLE && (i == 1 || j == 1));
                const bool row1 = !(PINHOLE && (i == 0 || j == 0));
                if (row0) H[k] = __builtin_fma(J0[i], J0[j], H[k]);
                if (row1) H[k] = __builtin_fma(J1[i], J1[j], H[k]);
                ++k;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (!(PINHOLE && i == 1)) b[i] = __builtin_fma(J0[i], e0, b[i]);
            if (!(PINHOLE && i == 0)) b[i] = __builtin_fma(J1[i], e1, b[i]);
        }
    }
};

// ---- wavefront (64 lanes) and workgroup sums -------------------------------------------------
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// Cross-lane exchange primitives for one lane-id bit.  Each returns, for a pair of values (a, b),
// "my half plus the partner lane's same half": lanes whose bit is 0 get a_self + a_partner, lanes
// whose bit is 1 get b_self + b_partner, where the partner differs in that bit and agrees in all
// higher bits.  Bits 5 and 4 are the gfx950 lane swaps, bits 3..1 select + DPP move.
__device__ __forceinline__ double halve_bit5(double a, double b) {
    const u32x2_t lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
__device__ __forceinline__ double halve_bit4(double a, double b) {
    const u32x2_t lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const u32x2_t hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi.x, lo.x) + __hiloint2double(hi.y, lo.y);
}
template <int CTRL>
__device__ __forceinline__ double halve_dpp(double a, double b, bool up) {
    const double keep = up ? b : a;
    const double send = up ? a : b;
    return keep + dpp_move<CTRL>(send);
}

// Sum of one value over the 64 lanes of a wave, returned in every lane: a butterfly over the lane
// bits with the data-parallel-primitive moves (row_mirror, row_half_mirror, two quad permutes: the
// partner sets of successive steps are disjoint, so four steps give the sum of each row of 16)
// and the two lane swaps for bits 4 and 5.  No LDS traffic, unlike a ds_bpermute shuffle: about
// 20 instructions and 6 short dependent steps.  Fixed order, reproducible.
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_move<0x140>(v); // row_mirror:      l <-> 15 - l
    v += dpp_move<0x141>(v); // row_half_mirror: l <-> 7 - l within each 8
    v += dpp_move<0x4E>(v);  // quad_perm [2,3,0,1]
    v += dpp_move<0xB1>(v);  // quad_perm [1,0,3,2]
    v = halve_bit4(v, v);    // rows 0<->1, 2<->3
    v = halve_bit5(v, v);    // lanes 0..31 <-> 32..63
    return v;
}

// Column sums of a 64 x N matrix held one row per lane (N = 32 or 64): v[j] of lane l is element
// (l, j).  Halving steps: at the step for lane-id bit b a lane keeps the half of its values that its
// own bit selects and receives the partner lane's partial sums of that same half, so the work
// halves every step -- about 2N exchanges in all instead of 6N for N independent butterflies.
//   N = 64: six halving steps (bits 5..0); lane l ends with the total of value l.
//   N = 32: five halving steps (bits 5..1) and one plain pairwise add (bit 0); lanes 2k and 2k+1
//           both end with the total of value k.
// Partners: bit 5/4 by v_permlane32_swap / v_permlane16_swap, bit 3 row_mirror (15 - l within a row
// of 16), bit 2 row_half_mirror (7 - l within 8), bit 1 and bit 0 quad permutes (l ^ 2, l ^ 1).
// The order of additions is fixed, so the result is reproducible.
template <int N>
__device__ __forceinline__ double wave_column_sums(double (&v)[N]) {
    static_assert(N == 32 || N == 64, "32 or 64 values per lane");
    const int lane = threadIdx.x & 63;
    constexpr int H5 = N / 2, H4 = N / 4, H3 = N / 8, H2 = N / 16, H1 = N / 32;
#pragma unroll
    for (int i = 0; i < H5; ++i) v[i] = halve_bit5(v[i], v[i + H5]);
#pragma unroll
    for (int i = 0; i < H4; ++i) v[i] = halve_bit4(v[i], v[i + H4]);
    {
        const bool up = (lane & 8) != 0;
#pragma unroll
        for (int i = 0; i < H3; ++i) v[i] = halve_dpp<0x140>(v[i], v[i + H3], up); // row_mirror
    }
    {
        const bool up = (lane & 4) != 0;
#pragma unroll
        for (int i = 0; i < H2; ++i) v[i] = halve_dpp<0x141>(v[i], v[i + H2], up); // row_half_mirror
    }
    {
        const bool up = (lane & 2) != 0;
#pragma unroll
        for (int i = 0; i < H1; ++i) v[i] = halve_dpp<0x4E>(v[i], v[i + H1], up); // quad_perm [2,3,0,1]
    }
    if (N == 64) {
        v[0] = halve_dpp<0xB1>(v[0], v[1], (lane & 1) != 0); // quad_perm [1,0,3,2]
    } else {
        v[0] = v[0] + dpp_move<0xB1>(v[0]);
    }
    return v[0];
}

// Wave-wide sums of four values at once: two halving steps (lane bits 5 and 4) leave value k's partial
// sums in the 16 lanes 16 k .. 16 k + 15, four butterfly steps inside the rows of 16 finish them.
// Every lane of row k returns the wave's total of value k.  21 instructions instead of the 72 of four
// separate butterflies.  Fixed order of additions.
__device__ __forceinline__ double wave_sums4(double v0, double v1, double v2, double v3) {
    const double x = halve_bit5(v0, v2); // lanes < 32: value 0, lanes >= 32: value 2
    const double y = halve_bit5(v1, v3); //             value 1,              value 3
    double z = halve_bit4(x, y);         // rows of 16 lanes: values 0, 1, 2, 3
    z += dpp_move<0x140>(z); // row_mirror
    z += dpp_move<0x141>(z); // row_half_mirror
    z += dpp_move<0x4E>(z);  // quad_perm [2,3,0,1]
    z += dpp_move<0xB1>(z);  // quad_perm [1,0,3,2]
    return z;
}

// v[0] + .. + v[N-1] added pairwise ((v0 + v1) + (v2 + v3)) + ..: log2(N) dependent additions instead
// of N - 1, in an order that depends on N alone.
template <int N>
__device__ __forceinline__ double pairwise_sum(double (&v)[N]) {
#pragma unroll
    for (int stride = 1; stride < N; stride *= 2) {
#pragma unroll
        for (int i = 0; i + stride < N; i += 2 * stride) v[i] += v[i + stride];
    }
    return v[0];
}

// Sums NV per-thread values over a workgroup of NT threads (NT a multiple of 64, <= 1024).
// lds: at least (NT/64)*64 doubles.  On return lane t < NV of EVERY wave holds the workgroup total
// of value t (other lanes: unspecified).  Fixed order of additions: wave_column_sums
// inside each wave, then the waves pairwise.
// TRAILING_SYNC = false leaves out the final barrier: the caller must not write `lds` again before
// its next workgroup barrier.
template <int NV, int NT, bool TRAILING_SYNC = true>
__device__ __forceinline__ double block_sum(const double (&vals)[NV], double* lds) {
    static_assert(NV <= 64, "one lane per value");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    double out = 0.0;
    if (NV > 8) {
        constexpr int N = (NV <= 32) ? 32 : 64;
        double v[N];
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = (i < NV) ? vals[i] : 0.0;
        const double tot = wave_column_sums<N>(v);
        if (N == 64)
            lds[wave * 64 + lane] = tot;
        else if ((lane & 1) == 0)
            lds[wave * 64 + (lane >> 1)] = tot; // with 32 values, lanes 2k and 2k+1 hold the total of value k
        __syncthreads();
        if (lane < NV) {
            double part[NT / 64];
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) part[w] = lds[w * 64 + lane];
            out = pairwise_sum(part);
        }
    } else if (NV <= 4) {
        const double s = wave_sums4(vals[0], NV > 1 ? vals[NV > 1 ? 1 : 0] : 0.0, NV > 2 ? vals[NV > 2 ? 2 : 0] : 0.0,
                                    NV > 3 ? vals[NV > 3 ? 3 : 0] : 0.0);
        if ((lane & 15) == 0 && (lane >> 4) < NV) lds[wave * 64 + (lane >> 4)] = s; // row k holds value k
        __syncthreads();
        if (lane < NV) {
            double part[NT / 64];
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) part[w] = lds[w * 64 + lane];
            out = pairwise_sum(part);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const double s = wave_sum(vals[i]);
            if (lane == 0) lds[wave * 64 + i] = s;
        }
        __syncthreads();
        if (lane < NV) {
            double part[NT / 64];
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) part[w] = lds[w * 64 + lane];
            out = pairwise_sum(part);
        }
    }
    if (TRAILING_SYNC) __syncthreads();
    return out;
}

// Folds `rows` partial records of NV doubles (row stride `stride`) into totals, in two phases so
// that several folds and other loads can be in flight together: issue() starts the loads of this
// thread's first B rows, finish() adds them (and any further rows) and combines the threads.
// The workgroup of NT threads is split into NT/CW row groups of CW lanes (CW = NV rounded up to a
// power of two); a group adds its rows in index order, then the groups are combined in a fixed
// two-level order, so the result does not depend on scheduling.  After finish() every thread with
// tid < NV holds the total of column tid.  A one-load-per-iteration loop would pay a full memory
// round trip per row; here a thread has B independent loads in flight.
// lds: at least NT doubles.
template <int NV, int NT, int B>
struct PartialFold {
    static constexpr int CW = NV <= 4 ? 4 : (NV <= 8 ? 8 : (NV <= 16 ? 16 : (NV <= 32 ? 32 : 64)));
    static constexpr int G = NT / CW; // row groups
    static_assert(NV <= 64 && NT % 64 == 0 && NT % CW == 0, "column per lane");
    double v[B];
    __device__ __forceinline__ void issue(const double* __restrict__ part, int rows, int stride) {
        const int col = threadIdx.x % CW, grp = threadIdx.x / CW;
#pragma unroll
        for (int k = 0; k < B; ++k) {
            const int r = grp + k * G;
            v[k] = (col < NV && r < rows) ? part[(size_t)r * stride + col] : 0.0;
        }
    }
    __device__ __forceinline__ double finish(const double* __restrict__ part, int rows, int stride, double* lds) {
        const int col = threadIdx.x % CW, grp = threadIdx.x / CW;
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < B; ++k) acc += v[k];
        if (col < NV) {
            for (int r = grp + B * G; r < rows; r += G) acc += part[(size_t)r * stride + col]; // more rows than B * G: rare
        }
        lds[threadIdx.x] = acc;
        __syncthreads();
        double a = 0.0;
        if (threadIdx.x < 64) {
#pragma unroll
            for (int k = 0; k < NT / 64; ++k) a += lds[threadIdx.x + 64 * k];
        }
        __syncthreads();
        if (threadIdx.x < 64) lds[threadIdx.x] = a;
        __syncthreads();
        double out = 0.0;
        if (threadIdx.x < NV) {
#pragma unroll
            for (int g = 0; g < 64 / CW; ++g) out += lds[g * CW + threadIdx.x];
        }
        __syncthreads();
        return out;
    }
    // Same result layout as finish() for values already in v[] (no further rows), with one barrier:
    // the row groups of a wave are first added inside the wave (lane swaps for bits 5 and 4, row
    // shifts for the two bits below when a record is only four wide), then the waves through LDS.
    // lds: NT/64 * CW doubles, not written again by the caller before its next barrier.
    __device__ __forceinline__ double finish_in_wave(double* lds) {
        static_assert(CW == 4 || CW == 32, "record widths of the two exchanges");
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        double acc = pairwise_sum(v);
        if (CW == 4) {
            acc += dpp_move<0x114>(acc); // row_shr:4
            acc += dpp_move<0x118>(acc); // row_shr:8: lanes 12..15 of every row of 16 hold the row's four groups
            acc = halve_bit4(acc, acc);
        }
        acc = halve_bit5(acc, acc);
        if (CW == 4) {
            if (lane >= 12 && lane < 16) lds[wave * CW + (lane - 12)] = acc;
        } else {
            if (lane < CW) lds[wave * CW + lane] = acc;
        }
        __syncthreads();
        double out = 0.0;
        if (lane < NV) { // every wave: lane t ends with the total of column t
            double part[NT / 64];
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) part[w] = lds[w * CW + lane];
            out = pairwise_sum(part);
        }
        return out;
    }
};

// finish() with one barrier when every row was covered by issue() (rows <= B * G), else the general form.
template <int NV, int NT, int B>
__device__ __forceinline__ double finish_fold(PartialFold<NV, NT, B>& f, const double* __restrict__ part, int rows,
                                              int stride, double* lds) {
    if (rows <= B * PartialFold<NV, NT, B>::G) { // uniform over the workgroup
        const double out = f.finish_in_wave(lds);
        __syncthreads(); // lds may be written again right away by the caller
        return out;
    }
    return f.finish(part, rows, stride, lds);
}

template <int NV, int NT, int B = 16>
__device__ __forceinline__ double fold_partials(const double* __restrict__ part, int rows, int stride, double* lds) {
    PartialFold<NV, NT, B> f;
    f.issue(part, rows, stride);
    return f.finish(part, rows, stride, lds);
}

// ---- small dense routines used by the solve ---------------------------------------

// 1 / p to within an ulp or two: the hardware's reciprocal estimate and two Newton steps, i.e. the
// front half of an IEEE division without its final rounding correction (half the dependent
// instructions).  For pivots and determinants inside the solve, which sit on the loop's critical
// path and whose last bit does not matter; p must be a normal number well inside the exponent range.
__device__ __forceinline__ double recip_nr(double p) {
    double r = __builtin_amdgcn_rcp(p);
    r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
    return r;
}

// Index of entry (i, j) of a symmetric 6 x 6 matrix in its row-major upper triangle (00,01,..,05,11,..,55).
__device__ __forceinline__ constexpr int tri6(int i, int j) {
    return i <= j ? i * 6 - (i * (i - 1)) / 2 + (j - i) : j * 6 - (j * (j - 1)) / 2 + (i - j);
}

// Solves A x = rhs for a symmetric positive definite 6 x 6 matrix A (upper triangle, tri6 order) entirely in
// the calling lane's registers: an LDL^T factorisation with the right-hand side carried along as a seventh
// row of L, then the back substitution.  No pivoting (not needed for a positive definite matrix), pivots'
// reciprocals by recip_nr.  About sixty dependent float64 operations, 700 cycles for a wave that runs
// nothing else -- and nothing in it crosses lanes: the same elimination spread over the lanes of a wave
// (one matrix entry per lane, pivot by v_readlane, row and column entries by ds_bpermute) took 950 cycles in
// isolation and 1 600 inside the loop kernels, where the LDS crossbar is shared with the other waves
// (tools/gj_probe.hip).
__device__ __forceinline__ void solve_spd6(const double (&A)[21], const double (&rhs)[6], double (&x)[6]) {
    double L[7][6]; // L[i][j], i > j: the unit lower factor; row 6: the right-hand side's forward substitution
    double d[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v[6];
        double dj = A[tri6(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = L[j][k] * d[k];
            dj = __builtin_fma(-L[j][k], v[k], dj);
        }
        d[j] = dj;
        const double inv = recip_nr(dj);
#pragma unroll
        for (int i = j + 1; i < 7; ++i) {
            double t = (i < 6) ? A[tri6(i, j)] : rhs[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = __builtin_fma(-L[i][k], v[k], t);
            L[i][j] = t * inv;
        }
    }
#pragma unroll
    for (int j = 5; j >= 0; --j) { // L^T x = z with z_j = y_j / d_j = L[6][j]
        double t = L[6][j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) t = __builtin_fma(-L[i][j], x[i], t);
        x[j] = t;
    }
}

// Inverse of a pose matrix [A t; 0 0 0 1] as [A^-1, -A^-1 t; 0 0 0 1] with A^-1 = adj(A) / det(A).
// The reference calls np.linalg.inv (src/ProjectiveICP.py:232), a general LU inverse; a
// rigid-transform shortcut [A^T | -A^T t] is not usable, because A is built from float32 sines and
// cosines (src/geometry_utils.py:41) and is orthonormal only to ~1e-8.  A is always within ~1e-7
// of a rotation, so the cofactor form is as accurate as LU here (a few ulp) and is a short,
// mostly independent instruction sequence for the single lane that runs it.
__device__ inline void invert_pose(const double* T, double* Tinv) {
    const double a = T[0], b = T[1], c = T[2];
    const double d = T[4], e = T[5], f = T[6];
    const double g = T[8], h = T[9], i = T[10];
    const double c00 = e * i - f * h, c01 = c * h - b * i, c02 = b * f - c * e;
    const double c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
    const double c20 = d * h - e * g, c21 = b * g - a * h, c22 = a * e - b * d;
    const double det = a * c00 + b * c10 + c * c20;
    const double r = recip_nr(det);
    const double m00 = c00 * r, m01 = c01 * r, m02 = c02 * r;
    const double m10 = c10 * r, m11 = c11 * r, m12 = c12 * r;
    const double m20 = c20 * r, m21 = c21 * r, m22 = c22 * r;
    const double tx = T[3], ty = T[7], tz = T[11];
    Tinv[0] = m00;
    Tinv[1] = m01;
    Tinv[2] = m02;
    Tinv[3] = -(m00 * tx + m01 * ty + m02 * tz);
    Tinv[4] = m10;
    Tinv[5] = m11;
    Tinv[6] = m12;
    Tinv[7] = -(m10 * tx + m11 * ty + m12 * tz);
    Tinv[8] = m20;
    Tinv[9] = m21;
    Tinv[10] = m22;
    Tinv[11] = -(m20 * tx + m21 * ty + m22 * tz);
    Tinv[12] = 0.0;
    Tinv[13] = 0.0;
    Tinv[14] = 0.0;
    Tinv[15] = 1.0;
}

// float32 sine and cosine exactly as numpy computes them on x86 with FMA (its AVX2 / AVX-512
// float32 loops, which also serve scalars): Cody-Waite reduction by pi/2 in three float32 pieces,
// then degree-9 / degree-8 polynomials, all in float32 with fused multiply-adds where numpy uses
// them.  The reference's v2T takes np.cos / np.sin of float32 angles (src/geometry_utils.py:41),
// and these functions are not correctly rounded (for |x| ~ 1 about one result in ten differs from
// the rounded float64 value), so matching the reference's poses to 1e-9 needs the same
// arithmetic, not a more accurate one.  Beyond numpy's Cody-Waite range (|x| > 71476 for cos,
// 117435 for sin) numpy defers to libm; there the float64 function rounded to float32 is used.
__device__ inline void numpy_sincosf(float x, float* sin_out, float* cos_out) {
#pragma clang fp contract(off)
    const float two_over_pi = 0x1.45f306p-1f, magic = 0x1.8p+23f;
    const float pio2_hi = -0x1.921fb0p+00f, pio2_med = -0x1.5110b4p-22f, pio2_lo = -0x1.846988p-48f;
    float q = x * two_over_pi;
    q = q + magic; // round to nearest integer
    q = q - magic;
    float r = __builtin_fmaf(q, pio2_hi, x);
    r = __builtin_fmaf(q, pio2_med, r);
    r = __builtin_fmaf(q, pio2_lo, r);
    const float r2 = r * r;
    float c = __builtin_fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
    c = __builtin_fmaf(c, r2, 0x1.55553cp-05f);
    c = __builtin_fmaf(c, r2, -0x1.000000p-01f);
    c = __builtin_fmaf(c, r2, 0x1.000000p+00f);
    float sn = __builtin_fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
    sn = __builtin_fmaf(sn, r2, 0x1.11119ap-07f);
    sn = __builtin_fmaf(sn, r2, -0x1.555556p-03f);
    sn = __builtin_fmaf(sn, r2, 0.0f);
    sn = __builtin_fmaf(sn, r, r);
    const int iq = (int)q;
    float s_res = ((iq & 1) == 0) ? sn : c;
    s_res = ((iq & 2) == 2) ? (0.0f - s_res) : s_res;
    const int iqc = iq + 1;
    float c_res = ((iqc & 1) == 0) ? sn : c;
    c_res = ((iqc & 2) == 2) ? (0.0f - c_res) : c_res;
    const float ax = fabsf(x);
    if (!(ax <= 117435.992f)) s_res = (float)sin((double)x);
    if (!(ax <= 71476.0625f)) c_res = (float)cos((double)x);
    *sin_out = s_res;
    *cos_out = c_res;
}

// v2T (src/geometry_utils.py:38-45): translation copied; R = Rx Ry Rz (:19-23) where the angles
// are first cast to float32 (:41) so each sine/cosine is a float32 value, widened to float64
// before the two 3x3 products.
// R = (Rx Ry) Rz written out from the six sines and cosines (entries 0..2, 4..6, 8..10 of T).  The
// factors hold exact zeros and ones, so in the reference's two 3x3 products every entry of Rx Ry is a
// single product and every entry of R a sum of at most two products; the terms multiplied by zero or
// one are left out (they do not change a floating-point sum).
__device__ __forceinline__ void rotation_from_sincos(double sx, double cx, double sy, double cy, double sz, double cz,
                                                     double* T) {
    const double a00 = cy, a02 = sy;                       // Rx Ry, row 0 (a01 = 0)
    const double a10 = sx * sy, a11 = cx, a12 = -(sx * cy); // row 1
    const double a20 = -(cx * sy), a21 = sx, a22 = cx * cy; // row 2
    T[0] = a00 * cz;
    T[1] = -(a00 * sz);
    T[2] = a02;
    T[4] = a10 * cz + a11 * sz;
    T[5] = a11 * cz - a10 * sz;
    T[6] = a12;
    T[8] = a20 * cz + a21 * sz;
    T[9] = a21 * cz - a20 * sz;
    T[10] = a22;
}

__device__ inline void v2T_device(const double* v, double* T) {
    float fsx, fcx, fsy, fcy, fsz, fcz;
    numpy_sincosf((float)v[3], &fsx, &fcx);
    numpy_sincosf((float)v[4], &fsy, &fcy);
    numpy_sincosf((float)v[5], &fsz, &fcz);
    rotation_from_sincos(fsx, fcx, fsy, fcy, fsz, fcz, T);
    T[3] = v[0];
    T[7] = v[1];
    T[11] = v[2];
    T[12] = 0.0;
    T[13] = 0.0;
    T[14] = 0.0;
    T[15] = 1.0;
}

// C = A B for two pose matrices (last rows (0, 0, 0, 1), which every pose here has: the state is
// checked on entry and v2T builds the increment): the products with that row are exact, so only the
// twelve entries of the upper three rows are formed.
__device__ inline void matmul4(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            C[i * 4 + j] = A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j] + A[i * 4 + 2] * B[2 * 4 + j];
        C[i * 4 + 3] = A[i * 4 + 0] * B[0 * 4 + 3] + A[i * 4 + 1] * B[1 * 4 + 3] + A[i * 4 + 2] * B[2 * 4 + 3] + A[i * 4 + 3];
    }
    C[12] = 0.0;
    C[13] = 0.0;
    C[14] = 0.0;
    C[15] = 1.0;
}

// np.add.reduce over 10 contiguous doubles: numpy's pairwise summation keeps 8 running sums for
// the first 8 elements, combines them as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) and then adds the
// remaining elements one by one.  np.mean / np.std at src/ProjectiveICP.py:212-213 go through it.
__device__ inline double numpy_sum10(const double* a) {
    double s = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    s += a[8];
    s += a[9];
    return s;
}

// ---- one iteration's tail: set choice, solve, pose update, schedules, state, history ----------
// (src/ProjectiveICP.py:300-321, :269-271, :200-238).  The work has two independent serial chains --
// the pose chain (damped solve -> v2T -> pose product -> inverse) and the schedule chain (threshold,
// error-slope statistics, counters, damping, stop test), which needs only the error and the inlier
// count, i.e. the stage records -- plus copies.  A workgroup runs the two chains on two different
// waves at the same time (the pose chain spread over the lanes of its wave, the schedule chain on one
// lane), a third wave fills the history row's H and b, and after one barrier the lanes write the
// state block and the history row in parallel.

// A reduced stage record, read either from memory (LDS or global) or from the lanes of the calling
// wave (lane t holds entry t; every lane of the wave must be active in dyn()).
struct MemRecord {
    const double* p;
    template <int I>
    __device__ __forceinline__ double at() const { return p[I]; }
    __device__ __forceinline__ double dyn(int i) const { return p[i]; }
};

// The entries of the stage records that the tail's scalar decisions need.
struct StageScalars {
    double n_inside, n1, sum_chi; // stage 1
    double s2_n, s2_chi;          // stage 2: members and sum of chi of the summed set
};
template <class R1, class R2>
__device__ __forceinline__ StageScalars stage_scalars(const R1& r1, const R2& r2) {
    StageScalars q;
    q.n_inside = r1.template at<S1_N_INSIDE>();
    q.n1 = r1.template at<S1_COUNT>();
    q.sum_chi = r1.template at<S1_SUM_CHI>();
    q.s2_n = r2.template at<S2_N>();
    q.s2_chi = r2.template at<S2_CHI>();
    return q;
}

// Fills the H, b and padding entries of a history row from the stage records (a whole wave, lanes
// tid = 0..63).
template <class R2>
__device__ __forceinline__ void fill_history_linear(double* hist_row, const StageScalars& q, const R2& r2, int tid) {
    const bool have = (q.n_inside != 0.0) && (q.n1 != 0.0);
    const int i = tid / 6, j = tid % 6;
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const int tri = lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo); // index of (lo, hi) in the row-major upper triangle
    const int idx = tid < 36 ? S2_H + tri : (tid < 42 ? S2_B + (tid - 36) : 0);
    const double val = r2.dyn(idx); // every lane fetches: a lane-to-lane fetch needs the whole wave
    if (tid < 36) {
        hist_row[PICP_H_H + tid] = have ? val : 0.0;
    } else if (tid < 42) {
        hist_row[PICP_H_B + (tid - 36)] = have ? val : 0.0;
    } else if (tid < 42 + (PICP_HIST_STRIDE - (PICP_H_H + 36))) {
        hist_row[PICP_H_H + 36 + (tid - 42)] = 0.0;
    }
}

// Workgroup-shared results of the two chains (LDS).
struct StepShared {
    double T1[16];   // pose after the step (:271)
    double Tinv[16]; // its inverse, the next camera state (:232)
    double dx[6];
    // from the schedule chain
    double error, n_inliers, n_stage2, used_stage2, mean_stage1, done;
    double thr, lam, mean_slope, sigma_slope, stuck, flicker, slope, stop;
    int it;        // 1-based number of this iteration (:179)
    int ring_slot; // slot of the slope ring buffer that receives `slope`, or -1
    int improved;  // this iteration's error is the smallest so far (first minimum wins, :249)
    double next[4]; // solve_step_lanes: what the next iteration's stages read besides Tinv (NEXT_*)
    double sys[PICP_S2_DOUBLES]; // solve_step_lanes: the stage-2 record, for the pose wave's uniform reads
#ifdef PICP_STAMPS
    unsigned long long t[6]; // diagnostic build: clock stamps inside solve_step
#endif
};

// Cross-lane reads of a double inside a wave: from a lane every lane names for itself (ds_bpermute:
// register to register through the LDS crossbar, no memory), or from one fixed lane (v_readlane).
__device__ __forceinline__ double lane_gather(double v, int src_lane) {
    const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
template <int LANE>
__device__ __forceinline__ double lane_bcast(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), LANE);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), LANE);
    return __hiloint2double(hi, lo);
}
template <int LANE>
__device__ __forceinline__ float lane_bcastf(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), LANE));
}
struct LaneRecord {
    double v;
    template <int I>
    __device__ __forceinline__ double at() const { return lane_bcast<I>(v); }
    __device__ __forceinline__ double dyn(int i) const { return lane_gather(v, i); }
};

// Pose chain, one whole wave (all 64 lanes active): H and b of the chosen set, damped solve, v2T, pose
// product, inverse (src/ProjectiveICP.py:269-271, :232).  Every iteration of the loop waits for it.
// The 6 x 6 system [H + lambda I] dx = -b is solved by every lane for itself (solve_spd6: the operands are the
// same in all lanes, so nothing has to cross lanes; the matrix is symmetric positive definite -- lambda >= min
// damping > 0 --, for which elimination without pivoting is stable, and the reference's np.linalg.lstsq returns
// the same unique solution).  The three float32 sine/cosine pairs of v2T are computed by three lanes at once;
// the rest (pose product, cofactor inverse) has uniform operands and is computed by every lane alike, lane 0
// stores it.
template <class R2, int I = 0>
__device__ __forceinline__ void fetch_system(const R2& r2, bool have, double lambda, double (&A)[21], double (&rhs)[6]) {
    if constexpr (I < 21) {
        A[I] = have ? r2.template at<S2_H + I>() : 0.0;
        fetch_system<R2, I + 1>(r2, have, lambda, A, rhs);
    } else if constexpr (I < 27) {
        rhs[I - 21] = have ? -r2.template at<S2_B + (I - 21)>() : -0.0; // :270
        fetch_system<R2, I + 1>(r2, have, lambda, A, rhs);
    } else {
#pragma unroll
        for (int j = 0; j < 6; ++j) A[tri6(j, j)] += lambda; // :269
    }
}

template <class R2>
__device__ __forceinline__ void step_pose_wave(const double* st, const StageScalars& q, const R2& r2, StepShared* sh) {
    const int lane = threadIdx.x & 63;
    double T1[16], dx[6];
    if (q.n_inside == 0.0) { // uniform: __one_step returns before the solve when there are no points (:263)
#pragma unroll
        for (int i = 0; i < 6; ++i) dx[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 16; ++i) T1[i] = st[ST_W_T_C + i];
    } else {
        const bool have = (q.n1 != 0.0); // else H = 0, b = 0 (:300)
        // the system (H + lambda I) dx = -b (:269-270), the same in every lane: entry t of the record by
        // v_readlane (or a uniform load), then the per-lane solve
        const double lambda = st[ST_LAMBDA];
        double A[21], rhs[6];
        fetch_system(r2, have, lambda, A, rhs);
#ifdef PICP_STAMPS
        {
            double keep = A[0] + A[20] + rhs[5]; // the stamp below follows the fetches
            asm volatile("" : "+v"(keep));
            if (lane == 0) sh->t[3] = __builtin_amdgcn_s_memrealtime();
            A[0] += keep * 0.0;
        }
#endif
        solve_spd6(A, rhs, dx);
#ifdef PICP_STAMPS
        if (lane == 0) sh->t[4] = __builtin_amdgcn_s_memrealtime();
#endif
        // v2T (src/geometry_utils.py:38-45): lanes 0, 1, 2 take the three angles
        const double ang = (lane == 0) ? dx[3] : ((lane == 1) ? dx[4] : dx[5]);
        float fs, fc;
        numpy_sincosf((float)ang, &fs, &fc);
        const double sx = lane_bcastf<0>(fs), cx = lane_bcastf<0>(fc);
        const double sy = lane_bcastf<1>(fs), cy = lane_bcastf<1>(fc);
        const double sz = lane_bcastf<2>(fs), cz = lane_bcastf<2>(fc);
        double dT[16];
        rotation_from_sincos(sx, cx, sy, cy, sz, cz, dT);
        dT[3] = dx[0];
        dT[7] = dx[1];
        dT[11] = dx[2];
        dT[12] = 0.0;
        dT[13] = 0.0;
        dT[14] = 0.0;
        dT[15] = 1.0;
        matmul4(st + ST_W_T_C, dT, T1); // :271
    }
#ifdef PICP_STAMPS
    if (lane == 0) sh->t[5] = __builtin_amdgcn_s_memrealtime();
#endif
    double Tinv[16];
    invert_pose(T1, Tinv);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            sh->T1[i] = T1[i];
            sh->Tinv[i] = Tinv[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) sh->dx[i] = dx[i];
    }
}

// Schedule chain, one lane: error and inlier count of the chosen set, then :200-228.
__device__ inline void step_schedule_chain(const double* st, const StageScalars& q, const picp_schedule& sc,
                                           StepShared* sh, StepShared& out) {
    const double n_inside = q.n_inside;
    const double n1 = q.n1;
    const bool redo_mode = (st[ST_MODE] != 0.0);
    double error = d_inf(), n_inliers = 0.0, n_stage2 = 0.0, used_stage2 = 0.0, mean_stage1 = 0.0;
    const bool done = (n_inside != 0.0); // :263 vs :273
    if (done && n1 != 0.0) {
        mean_stage1 = q.sum_chi / n1;
        used_stage2 = redo_mode ? 0.0 : 1.0;
        // in redo mode the number of mean-filter survivors was kept in the state block
        n_stage2 = redo_mode ? st[ST_N_STAGE2] : q.s2_n;
        n_inliers = q.s2_n;
        error = (n_inliers > 0.0) ? q.s2_chi / n_inliers : d_inf(); // :319-321
    }
    // kernel threshold schedule (:200-205)
    double thr = st[ST_THR];
    if (n_inliers == n_inside) {
        thr = sc.min_kernel_threshold;
    } else if (n_inliers < (double)sc.min_inliers) {
        if (thr + 10.0 <= sc.max_kernel_threshold) thr += 10.0;
    } else {
        if (thr - 10.0 >= sc.min_kernel_threshold) thr -= 10.0;
    }
    // error slope statistics (:208-213); `it` is 1-based as in the reference (:179)
    const int it = (int)st[ST_IT] + 1;
    const double error_prev = st[ST_ERR_PREV];
    double mean_slope = st[ST_MEAN_SLOPE], sigma_slope = st[ST_SIGMA_SLOPE];
    int ring_slot = -1;
    double slope = 0.0;
    if (it > 1) {
        const double inf = d_inf();
        slope = (error == inf || error_prev == inf) ? 1e3 : fabs(error_prev - error);
        double ring[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) ring[i] = st[ST_RING + i];
        ring_slot = it % 10;
#pragma unroll
        for (int i = 0; i < 10; ++i)
            if (i == ring_slot) ring[i] = slope;
        mean_slope = numpy_sum10(ring) / 10.0; // np.mean
        double dev[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const double d = ring[i] - mean_slope;
            dev[i] = d * d;
        }
        sigma_slope = sqrt(numpy_sum10(dev) / 10.0); // np.std (population)
    }
    double stuck = st[ST_STUCK], flicker = st[ST_FLICKER];
    stuck = (done && mean_slope < 1e-2 && sigma_slope < 1e-2) ? stuck + 1.0 : 0.0; // :216-217
    flicker = (done && mean_slope > 1.0 && sigma_slope > 1.0) ? flicker + 1.0 : 0.0; // :220-221
    // damping schedule (:224-225), limit = 10 (:164)
    double lam = st[ST_LAMBDA];
    if ((lam / 2.0) >= sc.min_damping && (stuck > 10.0 || (stuck == 0.0 && flicker == 0.0))) lam /= 2.0;
    if ((lam * 2.0) <= sc.max_damping && flicker > 10.0) lam *= 2.0;
    bool stop = done && (error < 0.01 || stuck > 100.0) && (sc.disable_stop == 0); // :228
    if (it >= sc.num_iterations) stop = true;                                      // :178
    // The results stay in this lane's registers (`out`): the same lane writes them to the state block and
    // the history row after the workgroup's barrier.  Only `improved` is needed by other lanes.
    out.error = error;
    out.n_inliers = n_inliers;
    out.n_stage2 = n_stage2;
    out.used_stage2 = used_stage2;
    out.mean_stage1 = mean_stage1;
    out.done = done ? 1.0 : 0.0;
    out.thr = thr;
    out.lam = lam;
    out.mean_slope = mean_slope;
    out.sigma_slope = sigma_slope;
    out.stuck = stuck;
    out.flicker = flicker;
    out.slope = slope;
    out.stop = stop ? 1.0 : 0.0;
    out.it = it;
    out.ring_slot = ring_slot;
    // np.argmin returns the first minimum; NaN would win in numpy, which cannot arise here
    // because a NaN chi never passes chi <= threshold.
    out.improved = (it == 1 || error < st[ST_BEST_ERR]) ? 1 : 0;
    sh->improved = out.improved;
    sh->it = it;
}

// The whole tail, called by every thread of a workgroup of at least 192 threads.
// st: the state block (LDS or global; read, then written).  s1/s2: the reduced stage records.
// hist_row: this iteration's history row or nullptr.  sh: LDS scratch.
// Returns false (to every thread) when the iteration could not be completed because the stage-2
// record holds the mean-filtered set F and F has fewer than min_inliers members (:307): the state's
// mode flag is then set and nothing else changes; the caller runs the stages again (stage 2 then
// sums the stage-1 set).  Ends with a workgroup barrier.
__device__ __forceinline__ bool solve_step(double* st, const double* s1, const double* s2, const picp_schedule& sc,
                                           double* hist_row, StepShared* sh) {
    const int tid = threadIdx.x;
    const bool redo_mode = (st[ST_MODE] != 0.0);
    const MemRecord r1{s1}, r2{s2};
    const StageScalars q = stage_scalars(r1, r2);
    const double n_set = q.s2_n;
    if (q.n_inside != 0.0 && q.n1 != 0.0 && !redo_mode && n_set < (double)sc.min_inliers) { // :307
        __syncthreads(); // every thread has read the mode flag
        if (tid == 0) {
            st[ST_MODE] = 1.0;
            st[ST_N_STAGE2] = n_set;
        }
        __syncthreads();
        return false;
    }
#ifdef PICP_STAMPS
    if (tid == 0) sh->t[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < 64) step_pose_wave(st, q, r2, sh);
#ifdef PICP_STAMPS
    if (tid == 0) sh->t[1] = __builtin_amdgcn_s_memrealtime();
#endif
    StepShared mine; // the schedule chain's results, in the registers of the lane that computed them
    if (tid == 64) step_schedule_chain(st, q, sc, sh, mine);
#ifdef PICP_STAMPS
    if (tid == 64) sh->t[2] = __builtin_amdgcn_s_memrealtime();
#endif
    if (hist_row != nullptr && tid >= 128 && tid < 192) fill_history_linear(hist_row, q, r2, tid - 128);
    __syncthreads();
#ifdef PICP_STAMPS
    if (tid == 0) sh->t[3] = __builtin_amdgcn_s_memrealtime();
#endif
    // state update (:231-238) and history row, one entry per lane
    if (tid < 16) {
        const double t = sh->T1[tid];
        const double ti = sh->Tinv[tid];
        const int improved = sh->improved;
        st[ST_W_T_C + tid] = t;
        st[ST_C_T_W + tid] = ti;
        if (improved) st[ST_BEST_T + tid] = t;
        if (hist_row != nullptr) hist_row[PICP_H_T + tid] = t;
    } else if (tid < 22) {
        if (hist_row != nullptr) hist_row[PICP_H_DX + (tid - 16)] = sh->dx[tid - 16];
    } else if (tid == 64) {
        st[ST_THR] = mine.thr;
        st[ST_LAMBDA] = mine.lam;
        st[ST_ERR_PREV] = mine.error;
        st[ST_MEAN_SLOPE] = mine.mean_slope;
        st[ST_SIGMA_SLOPE] = mine.sigma_slope;
        st[ST_STUCK] = mine.stuck;
        st[ST_FLICKER] = mine.flicker;
        st[ST_IT] = (double)mine.it;
        st[ST_STOP] = mine.stop;
        st[ST_MODE] = 0.0;
        if (mine.ring_slot >= 0) st[ST_RING + mine.ring_slot] = mine.slope;
        if (mine.improved) {
            st[ST_BEST_ERR] = mine.error;
            st[ST_BEST_IT] = (double)(mine.it - 1);
        }
        if (hist_row != nullptr) {
            const double n_in = q.n_inside, n_s1 = q.n1;
            hist_row[PICP_H_ERROR] = mine.error;
            hist_row[PICP_H_NUM_INLIERS] = mine.n_inliers;
            hist_row[PICP_H_KERNEL_THR] = mine.thr;
            hist_row[PICP_H_DAMPING] = mine.lam;
            hist_row[PICP_H_N_INSIDE] = n_in;
            hist_row[PICP_H_DONE] = mine.done;
            hist_row[PICP_H_N_STAGE1] = n_s1;
            hist_row[PICP_H_N_STAGE2] = mine.n_stage2;
            hist_row[PICP_H_USED_STAGE2] = mine.used_stage2;
            hist_row[PICP_H_MEAN_STAGE1] = mine.mean_stage1;
        }
    }
    __syncthreads();
#ifdef PICP_STAMPS
    if (tid == 0 && blockIdx.x == 0 && sh->it == 6 && PICP_STAMPS > 1)
        printf("solve_step: elimination done at %.2f, pose product at %.2f, pose wave %.2f us, schedule chain done at %.2f, barrier passed at %.2f, end %.2f\n",
               (sh->t[4] - sh->t[0]) * 0.01, (sh->t[5] - sh->t[0]) * 0.01, (sh->t[1] - sh->t[0]) * 0.01, (sh->t[2] - sh->t[0]) * 0.01, (sh->t[3] - sh->t[0]) * 0.01,
               (__builtin_amdgcn_s_memrealtime() - sh->t[0]) * 0.01);
#endif
    return true;
}

// The tail for a loop that receives the reduced records in registers -- lane t of EVERY wave holds
// entry t of the stage-1 record in g1 and of the stage-2 record in g2 -- and whose next iteration
// reads its inputs from `sh` (camera pose sh->Tinv, and sh->next[NEXT_*]) instead of the state block.
// That takes three workgroup barriers and two LDS round trips out of the path between one iteration's
// last exchange and the next iteration's stage 1, compared with solve_step: no record is written to
// LDS and waited for, and after the single barrier behind the two chains the stages may start while
// the state block and the history row are still being brought up to date.  Those writes are ordered
// before their next readers (the next tail, the caller's final copy of the state) by the barriers the
// caller passes in between; the caller puts one barrier before it reads the state block or hist_row.
// Same return value as solve_step.  sh->Tinv and sh->next must hold the loop-entry values before the
// first iteration (init_next).  redo_mode: sh->next[NEXT_MODE] as the caller read it at the head of this
// iteration (the schedule chain overwrites it before the barrier, so it is not read again here).
enum : int { NEXT_THR = 0, NEXT_MODE = 1, NEXT_IT = 2, NEXT_STOP = 3 };

__device__ inline void init_next(const double* st, StepShared* sh) {
    for (int i = 0; i < 16; ++i) sh->Tinv[i] = st[ST_C_T_W + i];
    sh->next[NEXT_THR] = st[ST_THR];
    sh->next[NEXT_MODE] = 0.0;
    sh->next[NEXT_IT] = st[ST_IT];
    sh->next[NEXT_STOP] = st[ST_STOP];
}

__device__ __forceinline__ bool solve_step_lanes(double* st, double g1, double g2, bool redo_mode,
                                                 const picp_schedule& sc, double* hist_row, StepShared* sh) {
    const int tid = threadIdx.x;
    const LaneRecord r1{g1}, r2{g2};
    const StageScalars q = stage_scalars(r1, r2); // v_readlane: the same scalars in every wave
    if (q.n_inside != 0.0 && q.n1 != 0.0 && !redo_mode && q.s2_n < (double)sc.min_inliers) { // :307
        __syncthreads(); // every thread has read the mode flag
        if (tid == 0) {
            st[ST_MODE] = 1.0;
            st[ST_N_STAGE2] = q.s2_n;
            sh->next[NEXT_MODE] = 1.0;
        }
        __syncthreads();
        return false;
    }
#ifdef PICP_STAMPS
    if (tid == 0) sh->t[0] = __builtin_amdgcn_s_memrealtime();
#endif
    if (tid < 64) {
        // Every lane of the pose wave needs all 27 entries of the record.  Lane-to-scalar reads (v_readlane, two
        // per entry) took 0.4 us here; one store per lane and uniform 16-byte loads (LDS broadcasts them) take a
        // quarter of that.  Same wave, so the loads follow the store without a barrier.
        if (tid < PICP_S2_DOUBLES) sh->sys[tid] = g2;
        step_pose_wave(st, q, MemRecord{sh->sys}, sh);
    }
#ifdef PICP_STAMPS
    if (tid == 0) sh->t[1] = __builtin_amdgcn_s_memrealtime();
#endif
    StepShared mine; // the schedule chain's results, in the registers of the lane that computed them
    if (tid == 64) {
        step_schedule_chain(st, q, sc, sh, mine);
#ifdef PICP_STAMPS
        sh->t[2] = __builtin_amdgcn_s_memrealtime();
#endif
        sh->next[NEXT_THR] = mine.thr;
        sh->next[NEXT_MODE] = 0.0;
        sh->next[NEXT_IT] = (double)mine.it;
        sh->next[NEXT_STOP] = mine.stop;
    }
    if (hist_row != nullptr && tid >= 128 && tid < 192) fill_history_linear(hist_row, q, r2, tid - 128);
    __syncthreads();
#ifdef PICP_STAMPS
    if (tid == 0 && blockIdx.x == 0 && sh->it == 6 && PICP_STAMPS > 1)
        printf("solve_step_lanes: system fetched at %.2f, solved at %.2f, pose product at %.2f, pose wave %.2f us, schedule chain done at %.2f, barrier passed at %.2f\n",
               (sh->t[3] - sh->t[0]) * 0.01, (sh->t[4] - sh->t[0]) * 0.01, (sh->t[5] - sh->t[0]) * 0.01, (sh->t[1] - sh->t[0]) * 0.01, (sh->t[2] - sh->t[0]) * 0.01,
               (__builtin_amdgcn_s_memrealtime() - sh->t[0]) * 0.01);
#endif
    // state update (:231-238) and history row, one entry per lane; nobody waits for it here
    if (tid < 16) {
        const double t = sh->T1[tid];
        const double ti = sh->Tinv[tid];
        const int improved = sh->improved;
        st[ST_W_T_C + tid] = t;
        st[ST_C_T_W + tid] = ti;
        if (improved) st[ST_BEST_T + tid] = t;
        if (hist_row != nullptr) hist_row[PICP_H_T + tid] = t;
    } else if (tid < 22) {
        if (hist_row != nullptr) hist_row[PICP_H_DX + (tid - 16)] = sh->dx[tid - 16];
    } else if (tid == 64) {
        st[ST_THR] = mine.thr;
        st[ST_LAMBDA] = mine.lam;
        st[ST_ERR_PREV] = mine.error;
        st[ST_MEAN_SLOPE] = mine.mean_slope;
        st[ST_SIGMA_SLOPE] = mine.sigma_slope;
        st[ST_STUCK] = mine.stuck;
        st[ST_FLICKER] = mine.flicker;
        st[ST_IT] = (double)mine.it;
        st[ST_STOP] = mine.stop;
        st[ST_MODE] = 0.0;
        if (mine.ring_slot >= 0) st[ST_RING + mine.ring_slot] = mine.slope;
        if (mine.improved) {
            st[ST_BEST_ERR] = mine.error;
            st[ST_BEST_IT] = (double)(mine.it - 1);
        }
        if (hist_row != nullptr) {
            hist_row[PICP_H_ERROR] = mine.error;
            hist_row[PICP_H_NUM_INLIERS] = mine.n_inliers;
            hist_row[PICP_H_KERNEL_THR] = mine.thr;
            hist_row[PICP_H_DAMPING] = mine.lam;
            hist_row[PICP_H_N_INSIDE] = q.n_inside;
            hist_row[PICP_H_DONE] = mine.done;
            hist_row[PICP_H_N_STAGE1] = q.n1;
            hist_row[PICP_H_N_STAGE2] = mine.n_stage2;
            hist_row[PICP_H_USED_STAGE2] = mine.used_stage2;
            hist_row[PICP_H_MEAN_STAGE1] = mine.mean_stage1;
        }
    }
    return true;
}

// Loop entry (src/ProjectiveICP.py:157-171): pose, camera state, base threshold and damping,
// cleared slope statistics and counters.
__device__ inline void init_state(double* st, const double* pose, double thr, double lambda, int num_iterations) {
    double inv[16];
    invert_pose(pose, inv);
    for (int i = ST_BEST_IT + 1; i < PICP_STATE_DOUBLES; ++i) st[i] = 0.0;
    for (int i = 0; i < 16; ++i) {
        st[ST_W_T_C + i] = pose[i];
        st[ST_C_T_W + i] = inv[i];
        st[ST_BEST_T + i] = pose[i];
        st[ST_START_T + i] = pose[i];
    }
    st[ST_THR] = thr;
    st[ST_LAMBDA] = lambda;
    st[ST_ERR_PREV] = d_inf();
    st[ST_MEAN_SLOPE] = 0.0;
    st[ST_SIGMA_SLOPE] = 0.0;
    st[ST_STUCK] = 0.0;
    st[ST_FLICKER] = 0.0;
    st[ST_IT] = 0.0;
    st[ST_STOP] = (num_iterations <= 0) ? 1.0 : 0.0;
    for (int i = 0; i < 10; ++i) st[ST_RING + i] = 0.0;
    st[ST_BEST_ERR] = d_inf();
    st[ST_BEST_IT] = 0.0;
}

// ---- per-correspondence work of the two stages (shared by the streaming scans, the batched kernel
// and the register-resident loop kernel) ----------------------------------------------------------
struct Stage1Sums {
    double sum_chi;
    unsigned n1, n_inside;
};

template <bool PINHOLE>
__device__ __forceinline__ void stage1_point(const double* __restrict__ T, const CameraK& cam, double thr, double mu,
                                             double mv, double x, double y, double z, Stage1Sums& s) {
    const Projected p = project_point<PINHOLE>(T, cam, x, y, z);
    const double e0 = mu - p.u, e1 = mv - p.v;         // src/ProjectiveICP.py:333
    const double chi = chi_of(e0, e1);                 // :293
    const bool keep = p.inside && (chi <= thr);        // :290, :294
    s.n_inside += p.inside ? 1u : 0u;
    s.n1 += keep ? 1u : 0u;
    s.sum_chi += keep ? chi : 0.0;
}

struct Stage2Sums {
    SetSums S;
    double chi;
    unsigned n;
    __device__ __forceinline__ void clear() {
        S.clear();
        chi = 0.0;
        n = 0u;
    }
};

// `all` = the state's mode flag: false sums the mean-filtered set (chi < mean, strict,
// src/ProjectiveICP.py:302), true the whole stage-1 set.
template <bool PINHOLE>
__device__ __forceinline__ void stage2_point(const double* __restrict__ T, const CameraK& cam, double thr, double mean,
                                             bool all, double mu, double mv, double x, double y, double z,
                                             Stage2Sums& s) {
    const Projected p = project_point<PINHOLE>(T, cam, x, y, z);
    const double e0 = mu - p.u, e1 = mv - p.v;
    const double chi = chi_of(e0, e1);
    const bool keep = p.inside && (chi <= thr) && (all || chi < mean);
    if (keep) {
        double J0[6], J1[6];
        jacobian_rows<PINHOLE>(p, cam, J0, J1);
        s.S.add<PINHOLE>(J0, J1, e0, e1);
        s.chi += chi;
        s.n += 1u;
    }
}

__device__ __forceinline__ void pack_sets(const Stage2Sums& s, double (&vals)[S2_USED]) {
#pragma unroll
    for (int i = 0; i < 21; ++i) vals[S2_H + i] = s.S.H[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) vals[S2_B + i] = s.S.b[i];
    vals[S2_CHI] = s.chi;
    vals[S2_N] = (double)s.n;
}

// ---- streaming the correspondence arrays --------------------------------------------------------
// Five 16-byte loads = two correspondences per lane.
// NONTEMPORAL: the loads carry the non-temporal hint (global_load_dwordx4 ... nt).  For arrays far beyond the
// 256 MB memory-side cache nothing that is read will be there when the next scan comes round, and loads that
// do not try to keep it stream faster: 2^24 correspondences (671 MB) 221 -> 199 us per iteration, 2^26 860 ->
// 800 us (6.7 TB/s).  Below about 512 MB the cache does serve part of every scan and the hint costs that:
// 2^23 (335 MB) 86 -> 101 us.
struct Corr2 {
    double2 u, v, x, y, z;
};
template <bool NONTEMPORAL = false>
__device__ __forceinline__ Corr2 load2(const double2* __restrict__ u, const double2* __restrict__ v,
                                       const double2* __restrict__ x, const double2* __restrict__ y,
                                       const double2* __restrict__ z, int64_t i) {
    Corr2 c;
    if constexpr (NONTEMPORAL) {
        typedef double nt2 __attribute__((ext_vector_type(2)));
        auto nt = [](const double2* p) {
            const nt2 t = __builtin_nontemporal_load(reinterpret_cast<const nt2*>(p));
            return double2{t.x, t.y};
        };
        c.u = nt(u + i);
        c.v = nt(v + i);
        c.x = nt(x + i);
        c.y = nt(y + i);
        c.z = nt(z + i);
    } else {
        c.u = u[i];
        c.v = v[i];
        c.x = x[i];
        c.y = y[i];
        c.z = z[i];
    }
    return c;
}

// How a scan walks the arrays (CorrStream's ORDER).
enum : int { SCAN_FORWARD = 0, SCAN_REVERSE = 1, SCAN_FORWARD_NONTEMPORAL = 2 };

// Streams a workgroup grid's share of the correspondences through `point(mu, mv, x, y, z)`.
// VEC == 2: 16-byte loads; the loads of the next grid-stride step are issued before the current
// pair of correspondences is processed, so a wave always has a full step of HBM requests in flight
// while it computes.  start() issues the very first loads; kernels call it before their prologue
// (state, partial records), because the addresses depend on nothing but the thread index.
// REVERSE: every thread visits its grid-stride steps from the last to the first.  A scan that runs
// against the direction of the scan before it starts with the data that scan read last, which the
// memory-side cache still holds when the arrays are larger than the cache.
// ORDER = SCAN_FORWARD_NONTEMPORAL: forward, with non-temporal loads (load2).
template <int VEC, int THREADS, int ORDER = SCAN_FORWARD>
struct CorrStream {
    static constexpr bool REVERSE = (ORDER == SCAN_REVERSE);
    static constexpr bool NONTEMPORAL = (ORDER == SCAN_FORWARD_NONTEMPORAL);
    const double *u, *v, *x, *y, *z;
    int64_t M, stride, tid, head;
    Corr2 A;
    __device__ __forceinline__ void start(const double* u_, const double* v_, const double* x_, const double* y_,
                                          const double* z_, int64_t M_) {
        u = u_, v = v_, x = x_, y = y_, z = z_, M = M_;
        stride = (int64_t)gridDim.x * THREADS;
        tid = (int64_t)blockIdx.x * THREADS + threadIdx.x;
        const int64_t n2 = M >> 1;
        // the thread's first step: index tid going up, its last grid-stride index going down
        head = (REVERSE && tid < n2) ? tid + (n2 - 1 - tid) / stride * stride : tid;
        if (VEC == 2 && tid < n2)
            A = load2<NONTEMPORAL>(reinterpret_cast<const double2*>(u), reinterpret_cast<const double2*>(v),
                      reinterpret_cast<const double2*>(x), reinterpret_cast<const double2*>(y),
                      reinterpret_cast<const double2*>(z), head);
    }
    template <typename F>
    __device__ __forceinline__ void run(F&& visit) {
        run(visit, [] {});
    }
    // between(): called after every step's two correspondences, while the next step's loads are in flight --
    // for work that needs no memory (correspondences a kernel holds on chip).
    template <typename F, typename G>
    __device__ __forceinline__ void run(F&& visit, G&& between) {
        auto point = [&](double mu, double mv, double px, double py, double pz) {
            homogeneous_nan(px, py, pz);
            visit(mu, mv, px, py, pz);
        };
        if (VEC == 2) {
            const int64_t n2 = M >> 1;
            const double2* u2 = reinterpret_cast<const double2*>(u);
            const double2* v2 = reinterpret_cast<const double2*>(v);
            const double2* x2 = reinterpret_cast<const double2*>(x);
            const double2* y2 = reinterpret_cast<const double2*>(y);
            const double2* z2 = reinterpret_cast<const double2*>(z);
            // Two register sets in ping-pong: while set A is processed the loads of set B are in
            // flight and vice versa, with no register copies between steps.  The prefetch is
            // unconditional (past the end it re-reads the current step, a cache hit whose data is
            // dropped): a load under a lane-dependent branch would make the compiler wait for every
            // outstanding load at the join, the prefetched ones included.
            const int64_t hop = REVERSE ? -stride : stride;
            auto more = [&](int64_t j) { return REVERSE ? (j >= 0) : (j < n2); };
            int64_t i = head;
            if (tid < n2) {
                Corr2 B;
                for (;;) {
                    const int64_t j = i + hop;
                    const bool hasB = more(j);
                    B = load2<NONTEMPORAL>(u2, v2, x2, y2, z2, hasB ? j : i);
                    point(A.u.x, A.v.x, A.x.x, A.y.x, A.z.x);
                    point(A.u.y, A.v.y, A.x.y, A.y.y, A.z.y);
                    between();
                    if (!hasB) break;
                    const int64_t k = j + hop;
                    const bool hasA = more(k);
                    A = load2<NONTEMPORAL>(u2, v2, x2, y2, z2, hasA ? k : j);
                    point(B.u.x, B.v.x, B.x.x, B.y.x, B.z.x);
                    point(B.u.y, B.v.y, B.x.y, B.y.y, B.z.y);
                    between();
                    if (!hasA) break;
                    i = k;
                }
            }
            if ((M & 1) && tid == 0) point(u[M - 1], v[M - 1], x[M - 1], y[M - 1], z[M - 1]);
        } else {
            for (int64_t i = tid; i < M; i += stride) point(u[i], v[i], x[i], y[i], z[i]);
        }
    }
};

} // namespace picp
