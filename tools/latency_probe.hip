// Probe: latencies of dependent instruction chains on one wave of gfx950 (clock cycles per link).
// Build: hipcc -O2 --offload-arch=gfx950 tools/latency_probe.hip -o tools/latency_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(double* out, double seed, int n) {
    __shared__ double lds[64];
    double x = seed + threadIdx.x * 1e-9;
    unsigned long long t0, t1;
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    // 1. dependent fma chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) x = __builtin_fma(x, 1.0000001, 1e-9);
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[0] = (double)(t1 - t0) / n;
    // 2. dependent IEEE division chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) x = 1.0 / (x + 1.5);
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[1] = (double)(t1 - t0) / n;
    // 3. rcp + two Newton steps
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        const double p = x + 1.5;
        double r = __builtin_amdgcn_rcp(p);
        r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
        r = __builtin_fma(r, __builtin_fma(-p, r, 1.0), r);
        x = r;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[2] = (double)(t1 - t0) / n;
    // 4. bpermute round trip of a double
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        const int lo = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, __double2loint(x));
        const int hi = __builtin_amdgcn_ds_bpermute(((threadIdx.x + 1) & 63) << 2, __double2hiint(x));
        x = __hiloint2double(hi, lo);
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[3] = (double)(t1 - t0) / n;
    // 5. LDS store + load round trip
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        lds[threadIdx.x] = x;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        x = lds[(threadIdx.x + 1) & 63] + 1e-9;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[4] = (double)(t1 - t0) / n;
    // 6. sqrt chain
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) x = sqrt(x + 2.0);
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[5] = (double)(t1 - t0) / n;
    // 7. readlane broadcast + add
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5);
        const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
        x = __hiloint2double(hi, lo) + 1e-9 * threadIdx.x;
    }
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[6] = (double)(t1 - t0) / n;
    // 8. barrier cost with 8 waves: measured by a second launch configuration
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) __syncthreads();
    t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[7] = (double)(t1 - t0) / n;
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        out[8] = (double)(r1 - r0) * 10.0; // ns
        out[9] = x;
    }
}

int main() {
    double* d;
    hipMalloc(&d, 128);
    for (int threads : {64, 512}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(threads), 0, 0, d, 1.25, 2000);
            hipDeviceSynchronize();
        }
        double h[10];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%d threads: fma %.1f  div %.1f  rcp+2NR %.1f  bpermute(double) %.1f  lds store+load %.1f  sqrt %.1f  readlane+add %.1f  barrier %.1f cycles per link; total %.0f ns\n",
               threads, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]);
    }
    return 0;
}
