// Probe: one-way latency of a flag handed from one workgroup to another through device memory, for a
// pair of workgroups on the same XCD and a pair on different XCDs (each XCD has its own L2), with
// device-scope accesses.  Also prints which XCD every workgroup of a small grid landed on.
// (Workgroup-scope accesses are not an alternative inside an XCD: the flag never becomes visible to
// the other compute unit -- the loads are served by its own first-level cache -- tried once, the
// run had to be killed.)
// Build: hipcc -O2 --offload-arch=gfx950 tools/hop_probe.hip -o tools/hop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// Workgroups a and b bounce a counter n times: a writes ping[0] = i, b waits for it and writes
// pong[0] = i, a waits for that.  Everyone else only reports its XCD.  Bounded spins.
template <int SCOPE>
__global__ void bounce(unsigned* ping, unsigned* pong, int a, int b, int n, unsigned* xcc, double* ns) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || ((int)blockIdx.x != a && (int)blockIdx.x != b)) return;
    const bool first = ((int)blockIdx.x == a);
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= n; ++i) {
        if (first) {
            __hip_atomic_store(ping, (unsigned)i, __ATOMIC_RELAXED, SCOPE);
            unsigned spins = 0;
            while (__hip_atomic_load(pong, __ATOMIC_RELAXED, SCOPE) != (unsigned)i && ++spins < (1u << 16)) {}
        } else {
            unsigned spins = 0;
            while (__hip_atomic_load(ping, __ATOMIC_RELAXED, SCOPE) != (unsigned)i && ++spins < (1u << 16)) {}
            __hip_atomic_store(pong, (unsigned)i, __ATOMIC_RELAXED, SCOPE);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (first) *ns = (double)(t1 - t0) * 10.0 / (2.0 * n); // 100 MHz clock; two hops per round
}

int main() {
    unsigned *flags, *xcc;
    double* ns;
    hipMalloc(&flags, 4096);
    hipMalloc(&xcc, 64 * sizeof(unsigned));
    hipMalloc(&ns, sizeof(double));
    const int G = 32, N = 2000;
    unsigned hx[64];
    for (int rep = 0; rep < 2; ++rep) {
        for (int pair = 0; pair < 3; ++pair) {
            const int a = 0, b = pair == 0 ? 8 : (pair == 1 ? 1 : 16); // 0/8 and 0/16: same XCD if dispatch is round-robin
            hipMemset(flags, 0, 4096);
            hipLaunchKernelGGL(bounce<__HIP_MEMORY_SCOPE_AGENT>, dim3(G), dim3(64), 0, 0, flags, flags + 512, a, b, N, xcc, ns);
            hipDeviceSynchronize();
            double h;
            hipMemcpy(&h, ns, sizeof(h), hipMemcpyDeviceToHost);
            hipMemcpy(hx, xcc, G * sizeof(unsigned), hipMemcpyDeviceToHost);
            printf("workgroups %d (XCD %u) and %d (XCD %u), device scope: %.0f ns per hop\n", a, hx[a], b, hx[b], h);
        }
    }
    printf("XCD of workgroups 0..%d:", G - 1);
    for (int i = 0; i < G; ++i) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
