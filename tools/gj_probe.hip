// Probe: cycles of the wave-parallel 6x6 Gauss-Jordan elimination of csrc/picp_device.h (step_pose_wave)
// in isolation, in a few variants.  Build: hipcc -O3 --offload-arch=gfx950 -I. tools/gj_probe.hip -o tools/gj_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

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
__device__ __forceinline__ double recip_nr(double p) {
    double r = __builtin_amdgcn_rcp(p);
    r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
    return r;
}

template <int VARIANT>
__device__ __forceinline__ double eliminate(double a, int i, int k) {
#define STEP(J)                                                                          \
    {                                                                                    \
        const double p = lane_bcast<(J) * 8 + (J)>(a);                                   \
        const double r = (VARIANT == 0) ? 1.0 / p : recip_nr(p);                         \
        const double aij = lane_gather(a, i * 8 + (J));                                  \
        const double ajk = lane_gather(a, (J) * 8 + k);                                  \
        a = (i == (J)) ? a * r : __builtin_fma(-(aij * r), ajk, a);                      \
    }
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5)
#undef STEP
    return a;
}

// variant 2: the matrix in LDS, every lane reads what it needs (no bpermute, no readlane)
__device__ __forceinline__ double eliminate_lds(double a, int i, int k, double* m /*64*/, int lane) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        m[lane] = a;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const double p = m[j * 8 + j], aij = m[i * 8 + j], ajk = m[j * 8 + k];
        const double r = recip_nr(p);
        a = (i == j) ? a * r : __builtin_fma(-(aij * r), ajk, a);
        __builtin_amdgcn_wave_barrier();
    }
    return a;
}

// variant 3: no cross-lane traffic at all -- every lane holds the whole (uniform) system in registers and runs an
// LDL^T factorisation with the right-hand side carried along as a seventh row, then the back substitution.
// A: upper triangle row-major (00,01,..,05,11,..,55) of the SPD matrix, rhs[6]; x receives the solution.
__device__ __forceinline__ constexpr int tri(int i, int j) { return i <= j ? i * 6 - (i * (i - 1)) / 2 + (j - i) : j * 6 - (j * (j - 1)) / 2 + (i - j); }
__device__ __forceinline__ void solve_ldlt(const double (&A)[21], const double (&rhs)[6], double (&x)[6]) {
    double L[7][6]; // L[i][j], i > j: unit lower factor; row 6: the right-hand side's forward substitution (y^T)
    double d[6], inv[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        double v[6];
        double dj = A[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; ++k) {
            v[k] = L[j][k] * d[k];
            dj = __builtin_fma(-L[j][k], v[k], dj);
        }
        d[j] = dj;
        inv[j] = recip_nr(dj);
#pragma unroll
        for (int i = j + 1; i < 7; ++i) {
            double s = (i < 6) ? A[tri(i, j)] : rhs[j];
#pragma unroll
            for (int k = 0; k < j; ++k) s = __builtin_fma(-L[i][k], v[k], s);
            L[i][j] = s * inv[j];
        }
    }
    // L[6][j] = y_j / d_j = z_j; back substitution L^T x = z
#pragma unroll
    for (int j = 5; j >= 0; --j) {
        double s = L[6][j];
#pragma unroll
        for (int i = j + 1; i < 6; ++i) s = __builtin_fma(-L[i][j], x[i], s);
        x[j] = s;
    }
}

__global__ void probe(double* out, const double* H) {
    __shared__ double m[64];
    const int lane = threadIdx.x & 63, i = lane >> 3, k = lane & 7;
    double a0 = 0.0;
    if (i < 6 && k < 7) a0 = H[i * 7 + k];
    double acc = 0.0;
    unsigned long long t[4];
    t[0] = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) acc += eliminate<0>(a0 + acc * 1e-30, i, k);
    t[1] = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) acc += eliminate<1>(a0 + acc * 1e-30, i, k);
    t[2] = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) acc += eliminate_lds(a0 + acc * 1e-30, i, k, m, lane);
    t[3] = __builtin_amdgcn_s_memtime();
    double A[21], rhs[6], x[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < 6; ++r) {
        for (int c = r; c < 6; ++c) A[tri(r, c)] = H[r * 7 + c];
        rhs[r] = H[r * 7 + 6];
    }
    double acc2 = 0.0;
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) {
        A[0] += acc2 * 1e-30;
        solve_ldlt(A, rhs, x);
        acc2 += x[0] + x[1] + x[2] + x[3] + x[4] + x[5];
    }
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = (t[1] - t[0]) / 8.0;
        out[1] = (t[2] - t[1]) / 8.0;
        out[2] = (t[3] - t[2]) / 8.0;
        out[3] = (t5 - t4) / 8.0;
        for (int q = 0; q < 6; ++q) out[80 + q] = x[q];
    }
    out[8 + threadIdx.x] = acc + acc2 * 1e-30;
}

// The per-lane solve as it sits in the loop kernels: a 512-thread workgroup, wave 0 solves, the other seven
// waves wait at the workgroup barrier behind it.
__global__ __launch_bounds__(512) void probe_in_workgroup(double* out, const double* H, int waiters_spin) {
    double A[21], rhs[6], x[6] = {0, 0, 0, 0, 0, 0};
    for (int r = 0; r < 6; ++r) {
        for (int c = r; c < 6; ++c) A[tri(r, c)] = H[r * 7 + c];
        rhs[r] = H[r * 7 + 6];
    }
    double acc2 = 0.0;
    __syncthreads();
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    for (int rep = 0; rep < 8; ++rep) {
        if (threadIdx.x < 64) {
            A[0] += acc2 * 1e-30;
            solve_ldlt(A, rhs, x);
            acc2 += x[0] + x[1] + x[2] + x[3] + x[4] + x[5];
        }
        __syncthreads();
    }
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = (t5 - t4) / 8.0;
    if (threadIdx.x < 64) out[8 + threadIdx.x] = acc2;
}

int main() {
    double hH[42];
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 7; ++k) hH[i * 7 + k] = (i == k) ? 10.0 + i : (k == 6 ? 1.0 + i : 1.0 / (1 + i + k));
    double *d, *dH;
    hipMalloc(&d, 8 * 128);
    hipMemset(d, 0, 8 * 128);
    hipMalloc(&dH, sizeof(hH));
    hipMemcpy(dH, hH, sizeof(hH), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dH);
        hipDeviceSynchronize();
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe_in_workgroup, dim3(1), dim3(512), 0, 0, d, dH, 0);
        hipDeviceSynchronize();
    }
    double h[96];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("cycles per 6x6 elimination: IEEE division %.0f, rcp + 2 Newton %.0f, through LDS %.0f, per-lane LDL^T %.0f;  x = %.6f %.6f %.6f\n", h[0],
           h[1], h[2], h[3], h[8 + 6], h[8 + 14], h[8 + 22]);
    printf("per-lane LDL^T by wave 0 of a 512-thread workgroup, seven waves waiting at the barrier: %.0f cycles\n", h[4]);
    printf("per-lane LDL^T solution: %.9f %.9f %.9f %.9f %.9f %.9f\n", h[80], h[81], h[82], h[83], h[84], h[85]);
    return 0;
}
