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
    if (threadIdx.x == 0) {
        out[0] = (t[1] - t[0]) / 8.0;
        out[1] = (t[2] - t[1]) / 8.0;
        out[2] = (t[3] - t[2]) / 8.0;
    }
    out[8 + threadIdx.x] = acc;
}

int main() {
    double hH[42];
    for (int i = 0; i < 6; ++i)
        for (int k = 0; k < 7; ++k) hH[i * 7 + k] = (i == k) ? 10.0 + i : (k == 6 ? 1.0 + i : 1.0 / (1 + i + k));
    double *d, *dH;
    hipMalloc(&d, 8 * 128);
    hipMalloc(&dH, sizeof(hH));
    hipMemcpy(dH, hH, sizeof(hH), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, dH);
        hipDeviceSynchronize();
    }
    double h[72];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("cycles per 6x6 elimination: IEEE division %.0f, rcp + 2 Newton %.0f, through LDS %.0f;  x = %.6f %.6f %.6f\n", h[0],
           h[1], h[2], h[8 + 6], h[8 + 14], h[8 + 22]);
    return 0;
}
