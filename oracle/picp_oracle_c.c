/*
 * CPU restatement, in plain C, of the reference's projective-ICP loop.
 *
 * THIS FILE IS TEST INFRASTRUCTURE: a second, independent checker next to oracle/picp_oracle.py and
 * the scalar CPU baseline of bench.py.  The shipped package never loads it.
 *
 * Unlike the numpy restatement, which is vectorised, this one follows the reference's own order of
 * work: one correspondence at a time in index order, sums in index order (src/ProjectiveICP.py:285-317),
 * numpy's pairwise summation where the reference calls np.mean / np.std (:212-213, :301, :321).
 * Each function names the reference lines it restates (paths relative to the reference checkout).
 *
 * Parity status: PINNED -- tests/test_oracle_c.py checks it against the vectors captured from the
 * reference itself (tests/golden/ *.npz, produced by oracle/gen_golden.py).
 *
 * Build: gcc -O2 -fPIC -shared -ffp-contract=off picp_oracle_c.c -o _build/liboracle.so -lm
 * (contraction is off so that the only fused multiply-adds are the ones written out below).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OC_HIST_STRIDE 80 /* same row layout as the device history (include/picp_hip.h, PICP_H_*) */
enum { H_T = 0, H_ERROR = 16, H_NUM_INLIERS = 17, H_KERNEL_THR = 18, H_DAMPING = 19, H_N_INSIDE = 20, H_DONE = 21,
       H_N_STAGE1 = 22, H_N_STAGE2 = 23, H_USED_STAGE2 = 24, H_MEAN_STAGE1 = 25, H_DX = 26, H_B = 32, H_H = 38 };

typedef struct oc_params {
    double K[9];
    double width, height;
    int32_t num_iterations, min_inliers;
    double kernel_threshold, min_kernel_threshold, max_kernel_threshold;
    double damping, min_damping, max_damping;
    int32_t disable_stop, reserved;
} oc_params;

/* ---- numpy's pairwise summation (numpy/core/src/umath/loops_utils.h.src, pairwise_sum): np.mean and
 * np.std reduce with it, so the mean the reference compares chi with (:301-302) has this rounding. */
static double pairwise_sum(const double* a, int64_t n) {
    if (n < 8) {
        double res = 0.0;
        for (int64_t i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int64_t i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
}

/* ---- float32 sine / cosine as numpy computes them on x86 with FMA (its AVX2 / AVX-512 float32 loops;
 * the reference takes np.cos / np.sin of float32 angles, src/geometry_utils.py:41).  Cody-Waite
 * reduction in three float32 pieces, degree-9 / degree-8 polynomials; beyond the reduction's range
 * numpy falls back to libm. */
static void numpy_sincosf(float x, float* sin_out, float* cos_out) {
    const float two_over_pi = 0x1.45f306p-1f, magic = 0x1.8p+23f;
    const float pio2_hi = -0x1.921fb0p+00f, pio2_med = -0x1.5110b4p-22f, pio2_lo = -0x1.846988p-48f;
    volatile float qv = x * two_over_pi; /* volatile: keep the float32 roundings the algorithm relies on */
    qv = qv + magic;
    qv = qv - magic;
    const float q = qv;
    float r = fmaf(q, pio2_hi, x);
    r = fmaf(q, pio2_med, r);
    r = fmaf(q, pio2_lo, r);
    const float r2 = r * r;
    float c = fmaf(0x1.98e616p-16f, r2, -0x1.6c06dcp-10f);
    c = fmaf(c, r2, 0x1.55553cp-05f);
    c = fmaf(c, r2, -0x1.000000p-01f);
    c = fmaf(c, r2, 0x1.000000p+00f);
    float sn = fmaf(0x1.7d3bbcp-19f, r2, -0x1.a06bbap-13f);
    sn = fmaf(sn, r2, 0x1.11119ap-07f);
    sn = fmaf(sn, r2, -0x1.555556p-03f);
    sn = fmaf(sn, r2, 0.0f);
    sn = fmaf(sn, r, r);
    const int iq = (int)q;
    float s_res = ((iq & 1) == 0) ? sn : c;
    s_res = ((iq & 2) == 2) ? (0.0f - s_res) : s_res;
    const int iqc = iq + 1;
    float c_res = ((iqc & 1) == 0) ? sn : c;
    c_res = ((iqc & 2) == 2) ? (0.0f - c_res) : c_res;
    const float ax = fabsf(x);
    if (!(ax <= 117435.992f)) s_res = sinf(x);
    if (!(ax <= 71476.0625f)) c_res = cosf(x);
    *sin_out = s_res;
    *cos_out = c_res;
}

static void matmul3(const double* A, const double* B, double* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0.0;
            for (int k = 0; k < 3; ++k) s += A[i * 3 + k] * B[k * 3 + j];
            C[i * 3 + j] = s;
        }
}

/* v2T (src/geometry_utils.py:38-45): R = Rx Ry Rz (:19-23) from float32 sines and cosines (:41). */
void oc_v2T(const double* v, double* T) {
    float sx, cx, sy, cy, sz, cz;
    numpy_sincosf((float)v[3], &sx, &cx);
    numpy_sincosf((float)v[4], &sy, &cy);
    numpy_sincosf((float)v[5], &sz, &cz);
    const double Rx[9] = {1, 0, 0, 0, cx, -(double)sx, 0, sx, cx};
    const double Ry[9] = {cy, 0, sy, 0, 1, 0, -(double)sy, 0, cy};
    const double Rz[9] = {cz, -(double)sz, 0, sz, cz, 0, 0, 0, 1};
    double Rxy[9], R[9];
    matmul3(Rx, Ry, Rxy);
    matmul3(Rxy, Rz, R);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[i * 4 + j] = R[i * 3 + j];
        T[i * 4 + 3] = v[i];
    }
    T[12] = T[13] = T[14] = 0.0;
    T[15] = 1.0;
}

/* General inverse of a 4x4 matrix by Gauss-Jordan elimination with partial pivoting (the reference
 * calls np.linalg.inv, src/ProjectiveICP.py:232). */
static int invert4(const double* A, double* Ainv) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = A[i * 4 + j];
            m[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(m[r][c]) > fabs(m[p][c])) p = r;
        if (m[p][c] == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < 8; ++j) {
                const double t = m[c][j];
                m[c][j] = m[p][j];
                m[p][j] = t;
            }
        const double d = m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = m[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) Ainv[i * 4 + j] = m[i][4 + j];
    return 0;
}

/* Solves the symmetric positive definite system A x = rhs (6x6) by Cholesky; the reference calls
 * np.linalg.lstsq on H + lambda I (src/ProjectiveICP.py:269-270), whose solution this is. */
static void solve6(const double* A, const double* rhs, double* x) {
    double L[6][6];
    memset(L, 0, sizeof L);
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= L[j][k] * L[j][k];
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
            L[i][j] = s / L[j][j];
        }
    }
    double y[6];
    for (int i = 0; i < 6; ++i) {
        double s = rhs[i];
        for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
        y[i] = s / L[i][i];
    }
    for (int i = 5; i >= 0; --i) {
        double s = y[i];
        for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
        x[i] = s / L[i][i];
    }
}

/* Camera.project_point (src/Camera.py:104-119).  Returns 1 when the point is in the image. */
static int project_point(const double* c_T_w, const oc_params* p, const double* pw, double* uv, double* pc) {
    double ph[4];
    for (int r = 0; r < 4; ++r)
        ph[r] = c_T_w[r * 4 + 0] * pw[0] + c_T_w[r * 4 + 1] * pw[1] + c_T_w[r * 4 + 2] * pw[2] + c_T_w[r * 4 + 3];
    for (int r = 0; r < 3; ++r) pc[r] = ph[r] / ph[3]; /* :106 */
    double h[3];
    for (int r = 0; r < 3; ++r) h[r] = p->K[r * 3 + 0] * pc[0] + p->K[r * 3 + 1] * pc[1] + p->K[r * 3 + 2] * pc[2];
    uv[0] = h[0] / h[2]; /* :108-109, before any test */
    uv[1] = h[1] / h[2];
    if (pc[2] <= 0.0) return 0;                                                          /* :111 */
    if (uv[0] < 0.0 || uv[0] >= p->width || uv[1] < 0.0 || uv[1] >= p->height) return 0; /* :117-119 */
    return 1;
}

/* __error_and_jacobian (src/ProjectiveICP.py:326-353).  J is 2x6 row-major. */
static int error_and_jacobian(const double* c_T_w, const oc_params* p, const double* uv_meas, const double* pw,
                              double* e, double* J) {
    double uv[2], ph[3];
    if (!project_point(c_T_w, p, pw, uv, ph)) return 0; /* :329-331 */
    e[0] = uv_meas[0] - uv[0];                          /* :333 */
    e[1] = uv_meas[1] - uv[1];
    double pcam[3]; /* :338 */
    for (int r = 0; r < 3; ++r) pcam[r] = p->K[r * 3 + 0] * ph[0] + p->K[r * 3 + 1] * ph[1] + p->K[r * 3 + 2] * ph[2];
    const double zi = 1.0 / pcam[2], zi2 = zi * zi;     /* :345-346 */
    const double Jp[2][3] = {{zi, 0.0, -pcam[0] * zi2}, {0.0, zi, -pcam[1] * zi2}}; /* :348-349 */
    double A[2][3]; /* J_proj @ K */
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
            A[i][j] = Jp[i][0] * p->K[0 * 3 + j] + Jp[i][1] * p->K[1 * 3 + j] + Jp[i][2] * p->K[2 * 3 + j];
    /* J_icp = [I | -skew(p_hat)] (:341-343) */
    const double x = ph[0], y = ph[1], z = ph[2];
    const double Ji[3][6] = {{1, 0, 0, 0, z, -y}, {0, 1, 0, -z, 0, x}, {0, 0, 1, y, -x, 0}};
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 6; ++j) J[i * 6 + j] = A[i][0] * Ji[0][j] + A[i][1] * Ji[1][j] + A[i][2] * Ji[2][j]; /* :351 */
    return 1;
}

/* __linearize (src/ProjectiveICP.py:276-323).  work: at least 15 * M doubles + M bytes.
 * aux (8 doubles): n_inside, n_stage1, n_stage2, used_stage2, mean_stage1. */
static void linearize(const double* uv, const double* pw, int64_t M, double thr, const double* c_T_w,
                      const oc_params* p, double* H, double* b, int64_t* n_inliers, double* error, double* aux,
                      double* work) {
    double* chi = work;            /* [M] of the stage-1 survivors, in order */
    double* es = work + M;         /* [M][2] */
    double* Js = work + 3 * M;     /* [M][12] */
    memset(H, 0, 36 * sizeof(double));
    memset(b, 0, 6 * sizeof(double));
    for (int k = 0; k < 8; ++k) aux[k] = 0.0;
    int64_t n1 = 0, n_inside = 0;
    for (int64_t i = 0; i < M; ++i) { /* :285 */
        double e[2], J[12];
        if (!error_and_jacobian(c_T_w, p, uv + 2 * i, pw + 3 * i, e, J)) continue; /* :290 */
        n_inside += 1;
        const double c = fma(e[1], e[1], e[0] * e[0]); /* :293 error.T @ error: numpy's dot of two elements */
        if (c <= thr) {                                /* :294 */
            chi[n1] = c;
            es[2 * n1] = e[0];
            es[2 * n1 + 1] = e[1];
            memcpy(Js + 12 * n1, J, sizeof J);
            n1 += 1;
        }
    }
    aux[0] = (double)n_inside;
    aux[1] = (double)n1;
    if (n1 == 0) { /* :300 */
        *n_inliers = 0;
        *error = INFINITY;
        return;
    }
    const double mean = pairwise_sum(chi, n1) / (double)n1; /* :301 np.mean */
    int64_t n2 = 0;
    for (int64_t k = 0; k < n1; ++k) n2 += (chi[k] < mean) ? 1 : 0; /* :302 */
    const int use2 = n2 >= p->min_inliers;                           /* :307 */
    aux[2] = (double)n2;
    aux[3] = use2 ? 1.0 : 0.0;
    aux[4] = mean;
    /* H, b over the final set in index order (:313-317); its chi values for the error (:319-321) */
    double* final_chi = work + 15 * M; /* [M] */
    int64_t n = 0;
    for (int64_t k = 0; k < n1; ++k) {
        if (use2 && !(chi[k] < mean)) continue;
        const double* J = Js + 12 * k;
        const double* e = es + 2 * k;
        for (int i = 0; i < 6; ++i) {
            for (int j = 0; j < 6; ++j) H[i * 6 + j] += J[i] * J[j] + J[6 + i] * J[6 + j]; /* J.T @ J */
            b[i] += J[i] * e[0] + J[6 + i] * e[1];                                          /* J.T @ e */
        }
        final_chi[n++] = chi[k];
    }
    *n_inliers = n;
    *error = (n > 0) ? pairwise_sum(final_chi, n) / (double)n : INFINITY; /* np.mean, :321 */
}

/* __projective_ICP on pre-associated pairs (src/ProjectiveICP.py:154-257).
 * uv: [M][2], pw: [M][3] row-major.  hist: [hist_rows][OC_HIST_STRIDE] or NULL.
 * Returns 0, or -1 when memory could not be obtained. */
int oc_projective_icp(const double* uv, const double* pw, int64_t M, const double* w_T_start, const oc_params* p,
                      double* T_out, int32_t* valid, int32_t* num_iterations, int32_t* best_index, double* hist,
                      int32_t hist_rows) {
    double* work = (double*)malloc((size_t)(16 * (M > 0 ? M : 1)) * sizeof(double));
    double* errors = (double*)malloc((size_t)(p->num_iterations > 0 ? p->num_iterations : 1) * sizeof(double));
    double* poses = (double*)malloc((size_t)(p->num_iterations > 0 ? p->num_iterations : 1) * 16 * sizeof(double));
    if (!work || !errors || !poses) {
        free(work);
        free(errors);
        free(poses);
        return -1;
    }
    double w_T_c0[16], c_T_w[16];
    memcpy(w_T_c0, w_T_start, sizeof w_T_c0);
    invert4(w_T_c0, c_T_w);
    double thr = p->kernel_threshold, lam = p->damping;  /* :157-158 */
    double error_prev = INFINITY, mean_slope = 0.0, sigma_slope = 0.0; /* :160-163 */
    double ring[10] = {0};                                               /* :166 */
    int stuck = 0, flicker = 0;                                          /* :167-168 */
    int it_done = 0;
    for (int it = 1; it <= p->num_iterations; ++it) { /* :178-179: icp_iteration is 1-based */
        double H[36], b[6], dx[6] = {0}, aux[8] = {0}, T1[16], err = INFINITY;
        int64_t n_inl = 0;
        int done = 0;
        /* the association's in-image filter (:182, src/data_association.py:44-46) uses the same
         * camera state as the linearisation, so the pairs that reach __one_step are the in-image ones */
        int64_t n_inside = 0;
        {
            double uvp[2], pc[3];
            for (int64_t i = 0; i < M; ++i) n_inside += project_point(c_T_w, p, pw + 3 * i, uvp, pc);
        }
        if (n_inside == 0) { /* __one_step, :263 */
            memcpy(T1, w_T_c0, sizeof T1);
            memset(H, 0, sizeof H);
            memset(b, 0, sizeof b);
        } else {
            linearize(uv, pw, M, thr, c_T_w, p, H, b, &n_inl, &err, aux, work); /* :266 */
            double A[36], rhs[6];
            memcpy(A, H, sizeof A);
            for (int i = 0; i < 6; ++i) {
                A[i * 6 + i] += lam; /* :269 */
                rhs[i] = -b[i];
            }
            solve6(A, rhs, dx);      /* :270 */
            double dT[16];
            oc_v2T(dx, dT);
            for (int i = 0; i < 4; ++i) /* :271 w_T_c1 = w_T_c0 @ v2T(dx) */
                for (int j = 0; j < 4; ++j) {
                    double s = 0.0;
                    for (int k = 0; k < 4; ++k) s += w_T_c0[i * 4 + k] * dT[k * 4 + j];
                    T1[i * 4 + j] = s;
                }
            done = 1;
        }
        /* kernel threshold (:200-205) */
        if (n_inl == n_inside) thr = p->min_kernel_threshold;
        else if (n_inl < p->min_inliers) {
            if (thr + 10.0 <= p->max_kernel_threshold) thr += 10.0;
        } else {
            if (thr - 10.0 >= p->min_kernel_threshold) thr -= 10.0;
        }
        /* error slope statistics (:208-213) */
        if (it > 1) {
            const double slope = (isinf(err) || isinf(error_prev)) ? 1e3 : fabs(error_prev - err);
            ring[it % 10] = slope;
            mean_slope = pairwise_sum(ring, 10) / 10.0;
            double dev[10];
            for (int k = 0; k < 10; ++k) dev[k] = (ring[k] - mean_slope) * (ring[k] - mean_slope);
            sigma_slope = sqrt(pairwise_sum(dev, 10) / 10.0);
        }
        stuck = (done && mean_slope < 1e-2 && sigma_slope < 1e-2) ? stuck + 1 : 0;   /* :216-217 */
        flicker = (done && mean_slope > 1.0 && sigma_slope > 1.0) ? flicker + 1 : 0; /* :220-221 */
        if (lam / 2.0 >= p->min_damping && (stuck > 10 || (stuck == 0 && flicker == 0))) lam /= 2.0; /* :224 */
        if (lam * 2.0 <= p->max_damping && flicker > 10) lam *= 2.0;                                  /* :225 */
        const int stop = done && (err < 0.01 || stuck > 100) && !p->disable_stop;                     /* :228 */
        /* state update and result lists (:231-238) */
        error_prev = err;
        memcpy(w_T_c0, T1, sizeof T1);
        invert4(w_T_c0, c_T_w);
        errors[it - 1] = err;
        memcpy(poses + 16 * (it - 1), T1, sizeof T1);
        if (hist != NULL && it - 1 < hist_rows) {
            double* row = hist + (size_t)(it - 1) * OC_HIST_STRIDE;
            memset(row, 0, OC_HIST_STRIDE * sizeof(double));
            memcpy(row + H_T, T1, sizeof T1);
            row[H_ERROR] = err;
            row[H_NUM_INLIERS] = (double)n_inl;
            row[H_KERNEL_THR] = thr;
            row[H_DAMPING] = lam;
            row[H_N_INSIDE] = (double)n_inside;
            row[H_DONE] = (double)done;
            row[H_N_STAGE1] = aux[1];
            row[H_N_STAGE2] = aux[2];
            row[H_USED_STAGE2] = aux[3];
            row[H_MEAN_STAGE1] = aux[4];
            memcpy(row + H_DX, dx, sizeof dx);
            memcpy(row + H_B, b, sizeof b);
            memcpy(row + H_H, H, sizeof H);
        }
        it_done = it;
        if (stop) break;
    }
    /* best iteration (:249-257): first minimum of the error list; invalid if it is > 5 */
    *num_iterations = it_done;
    int k = 0;
    for (int i = 1; i < it_done; ++i)
        if (errors[i] < errors[k]) k = i;
    *best_index = k;
    if (it_done == 0 || errors[k] > 5.0) {
        *valid = 0;
        memcpy(T_out, w_T_start, 16 * sizeof(double)); /* :250-251: the pose the loop was entered with */
    } else {
        *valid = 1;
        memcpy(T_out, poses + 16 * k, 16 * sizeof(double));
    }
    free(work);
    free(errors);
    free(poses);
    return 0;
}
