// Probe: what a float64 vector instruction costs a SIMD of gfx950 with one and with two waves on it, alone and
// in the patterns the stages of csrc/picp_resident.hip consist of (dependent chains, compares whose lane masks are
// combined on the scalar unit, selects on such masks).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe ; run: tools/valu_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)

constexpr int LOOPS = 200;

// every pattern: 64 vector instructions per loop trip
template <int PATTERN>
__global__ void probe(double* out, unsigned long long* cycles) {
    double a = threadIdx.x * 1e-3 + 1.0, b = 1.0000001, c = 0.5, d = 0.25, e = 0.125, f = 2.0, g = 3.0, h = 4.0;
    double k = 0.999999;
    int p = threadIdx.x, q = 7; // 32-bit operands of the selects
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < LOOPS; ++i) {
        if (PATTERN == 0) { // one dependent chain of fma
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %1\n\t") : "+v"(a) : "v"(k));
        } else if (PATTERN == 1) { // two chains
            asm volatile(REP16(REP4("v_fma_f64 %0, %0, %2, %2\n\tv_fma_f64 %1, %1, %2, %2\n\t")) : "+v"(a), "+v"(b) : "v"(k));
            i++;
        } else if (PATTERN == 2) { // four chains
            asm volatile(REP16("v_fma_f64 %0, %0, %4, %4\n\tv_fma_f64 %1, %1, %4, %4\n\tv_fma_f64 %2, %2, %4, %4\n\tv_fma_f64 %3, %3, %4, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                         : "v"(k));
        } else if (PATTERN == 3) { // eight chains
            asm volatile(REP4(REP4("v_fma_f64 %0, %0, %8, %8\n\tv_fma_f64 %1, %1, %8, %8\n\tv_fma_f64 %2, %2, %8, %8\n\tv_fma_f64 %3, %3, %8, %8\n\t")
                              REP4("v_fma_f64 %4, %4, %8, %8\n\tv_fma_f64 %5, %5, %8, %8\n\tv_fma_f64 %6, %6, %8, %8\n\tv_fma_f64 %7, %7, %8, %8\n\t"))
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
                         : "v"(k));
            i++;
        } else if (PATTERN == 4) { // one chain of 32-bit fma, for comparison
            float x = (float)a;
            asm volatile(REP64("v_fma_f32 %0, %0, %1, %1\n\t") : "+v"(x) : "v"((float)k));
            a = x;
        } else if (PATTERN == 5) { // four chains of 32-bit fma
            float x = (float)a, y = (float)b, z = (float)c, w = (float)d;
            asm volatile(REP16("v_fma_f32 %0, %0, %4, %4\n\tv_fma_f32 %1, %1, %4, %4\n\tv_fma_f32 %2, %2, %4, %4\n\tv_fma_f32 %3, %3, %4, %4\n\t")
                         : "+v"(x), "+v"(y), "+v"(z), "+v"(w)
                         : "v"((float)k));
            a = x, b = y, c = z, d = w;
        } else if (PATTERN == 6) {
            // compare -> scalar and with another mask -> select on it, between independent fma (4 chains):
            // 16 x (cmp, s_and, cndmask x2 (a 64-bit select), fma) = 64 vector instructions and 16 scalar
            asm volatile(REP16("v_cmp_lt_f64 s[20:21], %0, %4\n\t"
                               "s_and_b64 s[20:21], s[20:21], exec\n\t"
                               "v_cndmask_b32 %1, %1, %2, s[20:21]\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t")
                         : "+v"(a), "+v"(p), "+v"(q), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "scc");
            // (16 x 4 vector instructions)
        } else if (PATTERN == 7) {
            // the same without the scalar instruction in between: the select takes the compare's mask directly
            asm volatile(REP16("v_cmp_lt_f64 s[20:21], %0, %4\n\t"
                               "v_cndmask_b32 %1, %1, %2, s[20:21]\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t")
                         : "+v"(a), "+v"(p), "+v"(q), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "scc");
        } else if (PATTERN == 8) {
            // compare and scalar and, the select eight instructions later (the compiler's usual distance)
            asm volatile(REP16("v_cmp_lt_f64 s[20:21], %0, %4\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "s_and_b64 s[22:23], s[20:21], exec\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t"
                               "v_cndmask_b32 %1, %1, %2, s[22:23]\n\t")
                         : "+v"(a), "+v"(p), "+v"(q), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "s22", "s23", "scc");
        } else if (PATTERN == 12) {
            // what a correspondence's tests look like: four compares, their masks combined by three scalar ands,
            // one select on the result; 4 x (4 cmp + 1 select + 11 fma) = 64 vector instructions
            asm volatile(REP4("v_cmp_lt_f64 s[20:21], %0, %4\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t"
                              "v_cmp_gt_f64 s[22:23], %3, %4\n\t"
                              "v_fma_f64 %3, %3, %4, %4\n\t"
                              "v_cmp_lt_f64 s[24:25], %3, %0\n\t"
                              "s_and_b64 s[20:21], s[20:21], s[22:23]\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t"
                              "v_cmp_ge_f64 s[26:27], %0, %3\n\t"
                              "s_and_b64 s[20:21], s[20:21], s[24:25]\n\t"
                              "v_fma_f64 %3, %3, %4, %4\n\t"
                              "s_and_b64 s[20:21], s[20:21], s[26:27]\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t"
                              "v_cndmask_b32 %1, %1, %2, s[20:21]\n\t"
                              "v_fma_f64 %3, %3, %4, %4\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t"
                              "v_fma_f64 %3, %3, %4, %4\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t"
                              "v_fma_f64 %3, %3, %4, %4\n\t"
                              "v_fma_f64 %0, %0, %4, %4\n\t")
                         : "+v"(a), "+v"(p), "+v"(q), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        } else if (PATTERN == 13) { // lane reads into scalar registers (a mask kept in a vector register's lanes)
            asm volatile(REP16("v_readlane_b32 s20, %1, 3\n\t"
                               "v_readlane_b32 s21, %1, 4\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t")
                         : "+v"(a), "+v"(p), "+v"(q), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "scc");
        } else if (PATTERN == 14) { // an LDS read whose value is needed at once
            extern __shared__ double lds_probe[];
            double* addr = lds_probe + threadIdx.x;
            asm volatile(REP16("ds_read_b64 %1, %5\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "s_waitcnt lgkmcnt(0)\n\t"
                               "v_fma_f64 %2, %1, %4, %2\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                         : "v"(k), "v"((unsigned)(size_t)addr)
                         : "memory");
        } else if (PATTERN == 9) { // reciprocal (a transcendental-rate instruction) between fma
            asm volatile(REP16("v_rcp_f64 %1, %0\n\t"
                               "v_fma_f64 %0, %0, %4, %4\n\t"
                               "v_fma_f64 %2, %2, %4, %4\n\t"
                               "v_fma_f64 %3, %3, %4, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                         : "v"(k));
        } else if (PATTERN == 10) { // mul / add mix, four chains
            asm volatile(REP16("v_mul_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                         : "v"(k));
        } else if (PATTERN == 11) { // fma with a scalar-register operand (the pose sits in scalar registers)
            asm volatile("v_readfirstlane_b32 s20, %0\n\tv_readfirstlane_b32 s21, %1\n\t" ::"v"(__double2loint(k)), "v"(__double2hiint(k)) : "s20", "s21", "scc");
            asm volatile(REP16("v_fma_f64 %0, %0, s[20:21], %4\n\tv_fma_f64 %1, %1, s[20:21], %4\n\tv_fma_f64 %2, %2, s[20:21], %4\n\tv_fma_f64 %3, %3, s[20:21], %4\n\t")
                         : "+v"(a), "+v"(b), "+v"(c), "+v"(d)
                         : "v"(k)
                         : "s20", "s21", "scc");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + p + q;
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

static double* g_out;
static unsigned long long* g_cyc;

template <int PATTERN>
static void run(const char* what) {
    for (int threads : {256, 512}) {
        const int blocks = 256;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<PATTERN>, dim3(blocks), dim3(threads), 8192, 0, g_out, g_cyc);
        if (hipDeviceSynchronize() != hipSuccess) {
            printf("launch failed\n");
            return;
        }
        static unsigned long long h[256 * 16];
        (void)hipMemcpy(h, g_cyc, sizeof(unsigned long long) * blocks * (threads / 64), hipMemcpyDeviceToHost);
        unsigned long long worst = 0;
        for (int i = 0; i < blocks * (threads / 64); ++i) worst = h[i] > worst ? h[i] : worst;
        const double instr = 64.0 * LOOPS;
        // cycles per vector instruction of ONE wave = wall cycles / instructions of one wave; divided by the waves on
        // the SIMD: cycles per instruction issued by the SIMD
        printf("%-58s %d wave(s) per SIMD: %6.2f cycles per instruction of a wave, %5.2f per instruction of the SIMD\n", what,
               threads / 256, worst / instr, worst / instr / (threads / 256));
        fflush(stdout);
    }
}

int main() {
    (void)hipMalloc(&g_out, sizeof(double) * 256 * 1024);
    (void)hipMalloc(&g_cyc, sizeof(unsigned long long) * 256 * 16);
    run<7>("cmp f64 -> select, + 2 fma f64");
    run<6>("cmp f64 -> s_and -> select, + 2 fma f64");
    run<8>("cmp f64, fma, s_and, fma, select");
    run<12>("cmp f64 x4 -> s_and x3 -> select (+ 11 fma f64 per 16)");
    run<13>("readlane x2 of a vector register, + 2 fma f64");
    run<14>("ds_read_b64 + wait + 3 fma f64");
    run<0>("fma f64, one dependent chain");
    run<2>("fma f64, four chains");
    run<4>("fma f32, one dependent chain");
    run<5>("fma f32, four chains");
    run<11>("fma f64 with a scalar-register operand, four chains");
    run<9>("rcp f64 + 3 fma f64");
    return 0;
}
