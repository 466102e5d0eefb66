// Probe: the two stages of the on-chip loop (csrc/picp_resident.hip: stage1_keep, stage2_kept) alone -- one
// 512-thread workgroup per compute unit, R correspondences per thread in registers and LDS, no exchanges, no solve --
// timed per pass with the shader clock, so that variants of the per-correspondence code can be compared in seconds.
// The pose of a pass depends on the previous pass's sums (by a factor that changes nothing), as in the loop.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp \
//        -Wno-unused-function tools/stage_probe.hip -o tools/stage_probe
#define PICP_STAGE_PROBE
#include "../monocular-visual-odometry_amd/csrc/picp_resident.hip"

using namespace picp;

#ifndef PROBE_R
#define PROBE_R 8
#endif
constexpr int PASSES = 200;

template <int R, bool PINHOLE, bool MODERATE, int WHAT> // WHAT: 1 stage 1 only, 2 stage 2 only, 3 both
__global__ __launch_bounds__(NTR) void stage_probe(const double* __restrict__ u, const double* __restrict__ v,
                                                   const double* __restrict__ x, const double* __restrict__ y,
                                                   const double* __restrict__ z, Pose12 pose, CameraK cam, double thr,
                                                   double* out, unsigned long long* cycles) {
    extern __shared__ double dyn[];
    __shared__ double pose_lds[12];
    double* muv = dyn;
    double* kept = dyn + R * 2 * NTR;
    double rx[R], ry[R], rz[R];
    const int64_t first = (int64_t)blockIdx.x * NTR + threadIdx.x;
    const int64_t step = (int64_t)gridDim.x * NTR;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int64_t i = first + r * step;
        muv[(r * 2 + 0) * NTR + threadIdx.x] = u[i];
        muv[(r * 2 + 1) * NTR + threadIdx.x] = v[i];
        rx[r] = x[i];
        ry[r] = y[i];
        rz[r] = z[i];
        kept[(r * 2 + 0) * NTR + threadIdx.x] = 1.0;
        kept[(r * 2 + 1) * NTR + threadIdx.x] = 1.0;
    }
    const int mine = R;
    if (threadIdx.x < 12) pose_lds[threadIdx.x] = pose.m[threadIdx.x];
    __syncthreads();
    double carry = 0.0, mean = 1.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int pass = 0; pass < PASSES; ++pass) {
        double T[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) T[i] = uniform(pose_lds[i] + carry * 1e-300);
        Stage1Sums a{0.0, 0u, 0u};
        if (WHAT & 1) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double keep[2];
                stage1_keep<PINHOLE>(T, cam, thr, muv[(r * 2 + 0) * NTR + threadIdx.x], muv[(r * 2 + 1) * NTR + threadIdx.x],
                                     rx[r], ry[r], rz[r], r < mine, a, keep);
                kept[(r * 2 + 0) * NTR + threadIdx.x] = keep[0];
                kept[(r * 2 + 1) * NTR + threadIdx.x] = keep[1];
            }
            carry = wave_sums4(a.sum_chi, (double)a.n1, (double)a.n_inside, 0.0);
            mean = uniform(lane_bcast<0>(carry) / fmax(lane_bcast<16>(carry), 1.0));
        }
        if (WHAT & 2) {
            Stage2Sums s;
            s.clear();
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double chi = kept[(r * 2 + 0) * NTR + threadIdx.x];
                const bool in_set = (r < mine) && (chi <= thr) && (chi < mean);
                stage2_kept<PINHOLE, true, MODERATE>(T, cam, muv[(r * 2 + 0) * NTR + threadIdx.x],
                                                     muv[(r * 2 + 1) * NTR + threadIdx.x], rx[r], ry[r], rz[r], chi,
                                                     kept[(r * 2 + 1) * NTR + threadIdx.x], in_set, s);
            }
            double vals[S2_USED];
            pack_sets(s, vals);
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < S2_USED; ++i) acc += vals[i];
            carry = wave_sum(acc);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * NTR + threadIdx.x] = carry + mean;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (NTR / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int WHAT, bool MODERATE>
static void run(const char* what, const double* dev, int64_t n, const Pose12& pose, const CameraK& cam, double* out,
                unsigned long long* cyc) {
    constexpr int R = PROBE_R;
    auto kern = stage_probe<R, true, MODERATE, WHAT>;
    const unsigned dyn_bytes = R * 4 * NTR * sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_bytes);
    for (int rep = 0; rep < 3; ++rep)
        hipLaunchKernelGGL(kern, dim3(256), dim3(NTR), dyn_bytes, 0, dev, dev + n, dev + 2 * n, dev + 3 * n, dev + 4 * n, pose, cam,
                           1000.0, out, cyc);
    if (hipDeviceSynchronize() != hipSuccess) {
        printf("%s: launch failed: %s\n", what, hipGetErrorString(hipGetLastError()));
        return;
    }
    static unsigned long long h[256 * 8];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double older = 0, younger = 0, worst = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < 8; ++w) {
            const double c = (double)h[b * 8 + w] / PASSES;
            (w < 4 ? older : younger) += c / (256 * 4);
            worst = c > worst ? c : worst;
        }
    printf("%-44s %d per thread: %7.0f cycles per pass (older waves %7.0f, younger %7.0f) = %6.1f per correspondence and SIMD\n", what, R,
           worst, older, younger, worst / (2.0 * R));
    fflush(stdout);
}

int main() {
    constexpr int R = PROBE_R;
    const int64_t n = (int64_t)256 * NTR * R;
    double* host = (double*)malloc(sizeof(double) * 5 * n);
    unsigned long long seed = 12345;
    auto rnd = [&]() {
        seed = seed * 6364136223846793005ull + 1442695040888963407ull;
        return (double)(seed >> 11) / 9007199254740992.0;
    };
    for (int64_t i = 0; i < n; ++i) {
        const double X = rnd() * 8 - 4, Y = rnd() * 6 - 3, Z = 2 + rnd() * 8;
        host[2 * n + i] = X, host[3 * n + i] = Y, host[4 * n + i] = Z;
        host[i] = 180 * X / Z + 320 + (rnd() - 0.5), host[n + i] = 180 * Y / Z + 240 + (rnd() - 0.5);
    }
    double *dev, *out;
    unsigned long long* cyc;
    (void)hipMalloc(&dev, sizeof(double) * 5 * n);
    (void)hipMalloc(&out, sizeof(double) * 256 * NTR);
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8);
    (void)hipMemcpy(dev, host, sizeof(double) * 5 * n, hipMemcpyHostToDevice);
    Pose12 pose = {{1, 0, 0, 0.01, 0, 1, 0, -0.02, 0, 0, 1, 0.03}};
    CameraK cam = {{180, 0, 320, 0, 180, 240, 0, 0, 1}, 640, 480};
    run<1, false>("stage 1", dev, n, pose, cam, out, cyc);
    run<2, false>("stage 2, general form", dev, n, pose, cam, out, cyc);
    run<2, true>("stage 2, lean form", dev, n, pose, cam, out, cyc);
    run<3, false>("both stages, general form", dev, n, pose, cam, out, cyc);
    return 0;
}
