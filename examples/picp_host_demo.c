/*
 * A host program in plain C on top of the C ABI (include/picp_hip.h): no Python, no PyTorch.
 *
 *   picp_host_demo <input.bin> <output.bin>
 *
 * input.bin : int64 M, int64 num_iterations, double pose[16], then five planes of M doubles
 *             (u, v, x, y, z) -- the struct-of-arrays layout of the library.
 * output.bin: picp_result, then num_iterations history rows of PICP_HIST_STRIDE doubles.
 *
 * The loop parameters are the reference's defaults (config/config.yaml:11-22) and the camera is the
 * bundled one (data/camera.dat:2-13).  Device memory comes from hipMalloc; the library is handed
 * raw device pointers, which is all it ever sees from any host language.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/picp_hip.h"

#define CHECK_HIP(x)                                                             \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
            return 2;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s input.bin output.bin\n", argv[0]);
        return 1;
    }
    FILE* in = fopen(argv[1], "rb");
    if (!in) {
        perror(argv[1]);
        return 1;
    }
    int64_t M = 0, num_iterations = 0;
    double pose[16];
    if (fread(&M, sizeof M, 1, in) != 1 || fread(&num_iterations, sizeof num_iterations, 1, in) != 1 ||
        fread(pose, sizeof(double), 16, in) != 16 || M < 0 || num_iterations < 0) {
        fprintf(stderr, "bad header\n");
        return 1;
    }
    const size_t plane = (size_t)(M > 0 ? M : 1);
    double* host = (double*)malloc(5 * plane * sizeof(double));
    if (M > 0 && fread(host, sizeof(double), 5 * (size_t)M, in) != 5 * (size_t)M) {
        fprintf(stderr, "short input\n");
        return 1;
    }
    fclose(in);

    picp_ctx* ctx = NULL;
    if (picp_create(&ctx, 0, NULL) != PICP_OK) {
        fprintf(stderr, "picp_create: %s\n", picp_last_error());
        return 3;
    }
    double *planes = NULL, *state = NULL, *hist = NULL;
    const int rows = num_iterations > 0 ? (int)num_iterations : 1;
    CHECK_HIP(hipMalloc((void**)&planes, 5 * plane * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&state, PICP_STATE_DOUBLES * sizeof(double)));
    CHECK_HIP(hipMalloc((void**)&hist, (size_t)rows * PICP_HIST_STRIDE * sizeof(double)));
    CHECK_HIP(hipMemset(hist, 0, (size_t)rows * PICP_HIST_STRIDE * sizeof(double)));
    if (M > 0) CHECK_HIP(hipMemcpy(planes, host, 5 * (size_t)M * sizeof(double), hipMemcpyHostToDevice));

    picp_camera cam = {{180, 0, 320, 0, 180, 240, 0, 0, 1}, 640, 480};
    picp_schedule sched = {(int32_t)num_iterations, 10, 1000.0, 10.0, 2000.0, 64.0, 1.0, 64.0, 0, 0};
    picp_result res;
    memset(&res, 0, sizeof res);
    const int rc = picp_run(ctx, planes, planes + plane, planes + 2 * plane, planes + 3 * plane, planes + 4 * plane, M, pose,
                            &cam, &sched, state, hist, rows, &res);
    if (rc != PICP_OK) {
        fprintf(stderr, "picp_run: %s\n", picp_last_error());
        return 3;
    }
    double* hist_host = (double*)malloc((size_t)rows * PICP_HIST_STRIDE * sizeof(double));
    CHECK_HIP(hipMemcpy(hist_host, hist, (size_t)rows * PICP_HIST_STRIDE * sizeof(double), hipMemcpyDeviceToHost));
    FILE* out = fopen(argv[2], "wb");
    if (!out) {
        perror(argv[2]);
        return 1;
    }
    fwrite(&res, sizeof res, 1, out);
    fwrite(hist_host, sizeof(double), (size_t)res.num_iterations * PICP_HIST_STRIDE, out);
    fclose(out);
    printf("iterations %d valid %d best_index %d best_error %.17g\n", res.num_iterations, res.valid, res.best_index,
           res.best_error);
    (void)hipFree(planes);
    (void)hipFree(state);
    (void)hipFree(hist);
    picp_destroy(ctx);
    free(host);
    free(hist_host);
    return 0;
}
