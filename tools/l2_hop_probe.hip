// Probe: one-way latency of a flag handed between two workgroups of the SAME XCD through that XCD's L2 --
// stores and loads with the sc0 bit only (first-level cache bypassed, L2 not), written as the instructions
// because HIP's workgroup scope compiles to plain accesses, which the reader's own first-level cache serves --
// beside the device-scope hop (sc1) that the loop's exchanges use.  Bounded spins.
// Build: hipcc -O2 --offload-arch=gfx950 tools/l2_hop_probe.hip -o tools/l2_hop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int MODE> // 0: sc0 (L2 of the XCD), 1: sc1 (device)
__device__ __forceinline__ unsigned load_flag(const unsigned* p) {
    unsigned v;
    if (MODE == 0)
        asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 2)
        asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 3)
        asm volatile("global_load_dword %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (MODE == 4)
        asm volatile("buffer_inv sc0\n\tglobal_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else
        asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int MODE>
__device__ __forceinline__ void store_flag(unsigned* p, unsigned v) {
    if (MODE == 0 || MODE == 4)
        asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 2)
        asm volatile("global_store_dword %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else if (MODE == 3)
        asm volatile("global_store_dword %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    else
        asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void bounce(unsigned* ping, unsigned* pong, int a, int b, int n, unsigned* xcc, double* ns, unsigned* fails) {
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || ((int)blockIdx.x != a && (int)blockIdx.x != b)) return;
    const bool first = ((int)blockIdx.x == a);
    unsigned failed = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 1; i <= n && !failed; ++i) {
        unsigned spins = 0;
        if (first) {
            store_flag<MODE>(ping, (unsigned)i);
            while (load_flag<MODE>(pong) != (unsigned)i && ++spins < (1u << 14)) {}
        } else {
            while (load_flag<MODE>(ping) != (unsigned)i && ++spins < (1u << 14)) {}
            store_flag<MODE>(pong, (unsigned)i);
        }
        if (spins >= (1u << 14)) failed = (unsigned)i;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (first) *ns = (double)(t1 - t0) * 10.0 / (2.0 * n);
    if (failed) atomicMax(fails, failed);
}

int main() {
    unsigned *flags, *xcc, *fails;
    double* ns;
    hipMalloc(&flags, 8192);
    hipMalloc(&xcc, 64 * sizeof(unsigned));
    hipMalloc(&ns, sizeof(double));
    hipMalloc(&fails, sizeof(unsigned));
    const int G = 32, N = 2000;
    unsigned hx[64];
    for (int rep = 0; rep < 2; ++rep) {
        for (int mode = 0; mode < 5; ++mode) {
            for (int pair = 0; pair < 2; ++pair) {
                const int a = 0, b = pair == 0 ? 8 : 1; // 0/8: same XCD when dispatch is round-robin; 0/1: neighbours
                if (mode != 1 && pair == 1) continue;   // the L2 of one XCD is not visible from another
                hipMemset(flags, 0, 8192);
                hipMemset(fails, 0, sizeof(unsigned));
                auto k = mode == 0 ? bounce<0> : mode == 1 ? bounce<1> : mode == 2 ? bounce<2> : mode == 3 ? bounce<3> : bounce<4>;
                hipLaunchKernelGGL(k, dim3(G), dim3(64), 0, 0, flags, flags + 1024, a, b, N, xcc, ns, fails);
                hipDeviceSynchronize();
                double h;
                unsigned f;
                hipMemcpy(&h, ns, sizeof(h), hipMemcpyDeviceToHost);
                hipMemcpy(&f, fails, sizeof(f), hipMemcpyDeviceToHost);
                hipMemcpy(hx, xcc, G * sizeof(unsigned), hipMemcpyDeviceToHost);
                printf("workgroups %d (XCD %u) and %d (XCD %u), %s: %.0f ns per hop%s\n", a, hx[a], b, hx[b],
                       mode == 0 ? "sc0" : mode == 1 ? "sc1 (device scope)" : mode == 2 ? "nt" : mode == 3 ? "sc0 nt" : "buffer_inv sc0 + sc0 load", h, f ? "  ** a wait ran into its bound: not visible **" : "");
            }
        }
    }
    return 0;
}
