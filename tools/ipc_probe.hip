// Probe: two processes on ONE GPU exchange values through each other's device memory (HIP IPC)
// from inside kernels that run at the same time.  Answers, on the box it runs on:
//   - can uncached / fine-grained device memory be shared through hipIpcGetMemHandle,
//   - do kernels of two processes run concurrently when each fits half the CUs,
//   - what one publish -> seen-by-every-workgroup hop costs.
// Build: hipcc -O2 --offload-arch=gfx950 tools/ipc_probe.hip -o tools/ipc_probe
// Run:   tools/ipc_probe [workgroups per process] [rounds] [alloc: 0 hipMalloc, 1 uncached, 2 fine-grained]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "[%d] %s -> %s\n", g_rank, #x, hipGetErrorString(e_));        \
            return 10;                                                                    \
        }                                                                                 \
    } while (0)

static int g_rank = 0;

// Every workgroup waits, round after round, until the peer's value for that round has arrived in
// this process's own buffer; workgroup 0 publishes this process's value into the peer's buffer.
__global__ __launch_bounds__(512) void pingpong(unsigned long long* mine, unsigned long long* peers, int rounds,
                                                unsigned spin_limit, int* fail, unsigned long long* cycles) {
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 1; r <= rounds; ++r) {
        if (blockIdx.x == 0 && threadIdx.x == 0)
            __hip_atomic_store(peers, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (threadIdx.x == 0) {
            unsigned spins = 0;
            while (__hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (unsigned long long)r) {
                if (++spins > spin_limit) {
                    bad = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        if (bad) {
            if (threadIdx.x == 0) atomicAdd(fail, 1);
            return;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *cycles = __builtin_amdgcn_s_memrealtime() - t0;
}

static int run(int rank, int rfd, int wfd, int wgs, int rounds, int alloc) {
    g_rank = rank;
    CHECK(hipSetDevice(0));
    unsigned long long* buf = nullptr;
    if (alloc == 0) CHECK(hipMalloc(&buf, 4096));
    if (alloc == 1) CHECK(hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocUncached));
    if (alloc == 2) CHECK(hipExtMallocWithFlags((void**)&buf, 4096, hipDeviceMallocFinegrained));
    CHECK(hipMemset(buf, 0, 4096));
    CHECK(hipDeviceSynchronize());
    hipIpcMemHandle_t h, hp;
    CHECK(hipIpcGetMemHandle(&h, buf));
    if (write(wfd, &h, sizeof(h)) != (ssize_t)sizeof(h)) return 11;
    if (read(rfd, &hp, sizeof(hp)) != (ssize_t)sizeof(hp)) return 12;
    unsigned long long* peer = nullptr;
    CHECK(hipIpcOpenMemHandle((void**)&peer, hp, hipIpcMemLazyEnablePeerAccess));
    int* fail = nullptr;
    unsigned long long* cyc = nullptr;
    CHECK(hipMalloc(&fail, 4));
    CHECK(hipMalloc(&cyc, 8));
    CHECK(hipMemset(fail, 0, 4));
    CHECK(hipMemset(cyc, 0, 8));
    CHECK(hipDeviceSynchronize());
    // both processes are set up before either launches
    char c = 'x';
    if (write(wfd, &c, 1) != 1 || read(rfd, &c, 1) != 1) return 13;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    hipLaunchKernelGGL(pingpong, dim3(wgs), dim3(512), 0, 0, buf, peer, rounds, 1u << 20, fail, cyc);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    clock_gettime(CLOCK_MONOTONIC, &b);
    int hfail = 0;
    unsigned long long hcyc = 0;
    CHECK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&hcyc, cyc, 8, hipMemcpyDeviceToHost));
    const double ms = (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6;
    printf("[%d] alloc %d, %d workgroups, %d rounds: workgroups that gave up %d, wall %.3f ms, in-kernel %.3f us per round\n",
           rank, alloc, wgs, rounds, hfail, ms, hcyc * 0.01 / rounds);
    fflush(stdout);
    if (write(wfd, &c, 1) != 1 || read(rfd, &c, 1) != 1) return 14; // keep the mapping alive until both are done
    CHECK(hipIpcCloseMemHandle(peer));
    CHECK(hipFree(buf));
    return hfail ? 1 : 0;
}

int main(int argc, char** argv) {
    const int wgs = argc > 1 ? atoi(argv[1]) : 100;
    const int rounds = argc > 2 ? atoi(argv[2]) : 1000;
    const int alloc = argc > 3 ? atoi(argv[3]) : 1;
    int p2c[2], c2p[2];
    if (pipe(p2c) || pipe(c2p)) return 2;
    const pid_t pid = fork(); // before any HIP call
    if (pid == 0) return run(1, p2c[0], c2p[1], wgs, rounds, alloc);
    const int rc = run(0, c2p[0], p2c[1], wgs, rounds, alloc);
    int st = 0;
    waitpid(pid, &st, 0);
    printf("parent rc %d, child rc %d\n", rc, WIFEXITED(st) ? WEXITSTATUS(st) : -1);
    return rc || !WIFEXITED(st) || WEXITSTATUS(st);
}
