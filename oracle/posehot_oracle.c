/*
 * posehot_oracle.c — CPU restatement of the camera-pose hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library (oracle/libposehot_oracle.so).  The product path (pose_estimation_amd/) never does.
 *
 * PARITY STATUS.  The reference (alirezafalah/pose_estimation) has no descriptor matcher, no
 * RANSAC scoring and no triangulation, and the library its arithmetic lives in
 * (opencv-python, unpinned; call sites main.py:95,101-104,122 and validate_pose.py:88,93-96,
 * 183,269,275) is not installed here.  Sections (1)-(3) below are therefore "parity unpinned":
 * they restate the contract of SURVEY.md §8 rows a-9, a-10, a-11 / include/posehot.h, not
 * reference code.  Section (4) follows validate_pose.py:181-184 and :260-288 through the
 * OpenCV-documented 5-coefficient distortion model.
 *
 * Written as plain scalar loops on purpose: every function is the definition, not an
 * optimisation.  f32 decisions use explicit fmaf() and the file is compiled with
 * -ffp-contract=off so that no other contraction happens; the op order is the contract in
 * DESIGN.md §"Arithmetic contracts".  Loops over items / rows are OpenMP-parallel so that the
 * same code serves as bench.py's cpu_baseline ("port", cores = omp threads).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_NO_INDEX (-1)
#define ORC_HAMMING_NONE 0x7fffffff

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------------------------
 * (1) matching                                                        SURVEY.md §8 row a-9
 * ------------------------------------------------------------------------------------------ */

/* 256-bit Hamming, nearest + second nearest, ties -> lowest train index. */
void orc_match_hamming_top2(const uint8_t* q, const uint8_t* t, int nq, int nt, int bytes,
                            int32_t* idx, int32_t* d1, int32_t* d2) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        int best = ORC_HAMMING_NONE, second = ORC_HAMMING_NONE, bi = ORC_NO_INDEX;
        const uint8_t* qi = q + (size_t)i * bytes;
        for (int j = 0; j < nt; ++j) {
            const uint8_t* tj = t + (size_t)j * bytes;
            int d = 0;
            for (int w = 0; w < bytes; w += 8) {
                uint64_t a, b;
                memcpy(&a, qi + w, 8);
                memcpy(&b, tj + w, 8);
                d += __builtin_popcountll(a ^ b);
            }
            if (d < best) { second = best; best = d; bi = j; }
            else if (d < second) { second = d; }
        }
        idx[i] = bi; d1[i] = best; d2[i] = second;
    }
}

/* squared norm of each row as a k-ordered f32 fma chain from 0. */
void orc_row_sqnorm(const float* a, int n, int dim, float* out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        float acc = 0.0f;
        for (int k = 0; k < dim; ++k) acc = fmaf(a[(size_t)i * dim + k], a[(size_t)i * dim + k], acc);
        out[i] = acc;
    }
}

/* L2 top-2 with the library's f32 contract (DESIGN.md "Arithmetic contracts"):
 *   s(i,j) = fma chain of t[j][k]*q[i][k] started at -0.5f*|t_j|^2, taking k in the order
 *            8c + {0,4,1,5,2,6,3,7} for c = 0..dim/8-1 (the order in which the two lane halves
 *            of v_mfma_f32_32x32x2_f32 consume 16-byte operand chunks);
 *   best = argmax_j s (ties -> lowest j);  dsq = max(0, fma(-2, s, |q_i|^2));
 *   |.|^2 = natural-order fma chain from 0 (orc_row_sqnorm). */
static const int kOrder8[8] = {0, 4, 1, 5, 2, 6, 3, 7};
void orc_match_l2_top2(const float* q, const float* t, int nq, int nt, int dim,
                       int32_t* idx, float* d1sq, float* d2sq) {
    float* tn = (float*)malloc(sizeof(float) * (size_t)(nt > 0 ? nt : 1));
    float* qn = (float*)malloc(sizeof(float) * (size_t)(nq > 0 ? nq : 1));
    orc_row_sqnorm(t, nt, dim, tn);
    orc_row_sqnorm(q, nq, dim, qn);
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        float m1 = -INFINITY, m2 = -INFINITY;
        int bi = ORC_NO_INDEX;
        const float* qi = q + (size_t)i * dim;
        for (int j = 0; j < nt; ++j) {
            const float* tj = t + (size_t)j * dim;
            float acc = -0.5f * tn[j];
            for (int c = 0; c < dim; c += 8)
                for (int e = 0; e < 8; ++e) acc = fmaf(tj[c + kOrder8[e]], qi[c + kOrder8[e]], acc);
            if (acc > m1) { m2 = m1; m1 = acc; bi = j; }
            else if (acc > m2) { m2 = acc; }
        }
        idx[i] = bi;
        d1sq[i] = (bi >= 0) ? fmaxf(0.0f, fmaf(-2.0f, m1, qn[i])) : INFINITY;
        d2sq[i] = (nt >= 2) ? fmaxf(0.0f, fmaf(-2.0f, m2, qn[i])) : INFINITY;
    }
    free(tn); free(qn);
}

/* Independent float64 statement: direct sum of squared differences (what a textbook brute-force
 * matcher computes); used for the >= 99.9 % match-set agreement bound of north_star. */
void orc_match_l2_top2_f64(const float* q, const float* t, int nq, int nt, int dim,
                           int32_t* idx, double* d1sq, double* d2sq) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nq; ++i) {
        double m1 = INFINITY, m2 = INFINITY;
        int bi = ORC_NO_INDEX;
        for (int j = 0; j < nt; ++j) {
            double acc = 0.0;
            for (int k = 0; k < dim; ++k) {
                double d = (double)q[(size_t)i * dim + k] - (double)t[(size_t)j * dim + k];
                acc += d * d;
            }
            if (acc < m1) { m2 = m1; m1 = acc; bi = j; }
            else if (acc < m2) { m2 = acc; }
        }
        idx[i] = bi; d1sq[i] = m1; d2sq[i] = m2;
    }
}

void orc_ratio_filter_hamming(const int32_t* idx, const int32_t* d1, const int32_t* d2, int64_t n,
                              float ratio, int32_t* keep) {
    for (int64_t i = 0; i < n; ++i)
        keep[i] = (idx[i] >= 0 && d2[i] != ORC_HAMMING_NONE && (float)d1[i] < ratio * (float)d2[i]) ? 1 : 0;
}
void orc_ratio_filter_l2(const int32_t* idx, const float* d1sq, const float* d2sq, int64_t n,
                         float ratio, int32_t* keep) {
    const float r2 = ratio * ratio;
    for (int64_t i = 0; i < n; ++i)
        keep[i] = (idx[i] >= 0 && isfinite(d2sq[i]) && d1sq[i] < r2 * d2sq[i]) ? 1 : 0;
}

/* stable compaction, one item. */
int orc_compact_matches(const int32_t* idx, const int32_t* keep, const float* kp_q, const float* kp_t,
                        int nq, float* x1, float* y1, float* x2, float* y2,
                        int32_t* match_q, int32_t* match_t) {
    int n = 0;
    for (int i = 0; i < nq; ++i) {
        if (!keep[i]) continue;
        int j = idx[i];
        x1[n] = kp_q[2 * i]; y1[n] = kp_q[2 * i + 1];
        x2[n] = kp_t[2 * j]; y2[n] = kp_t[2 * j + 1];
        if (match_q) match_q[n] = i;
        if (match_t) match_t[n] = j;
        ++n;
    }
    return n;
}

/* ------------------------------------------------------------------------------------------
 * (2) hypothesis scoring                                              SURVEY.md §8 row a-10
 * ------------------------------------------------------------------------------------------ */

static inline int sampson_inlier(const float* E, float x1, float y1, float x2, float y2, float thr_sq) {
    const float e0 = fmaf(E[0], x1, fmaf(E[1], y1, E[2]));
    const float e1 = fmaf(E[3], x1, fmaf(E[4], y1, E[5]));
    const float e2 = fmaf(E[6], x1, fmaf(E[7], y1, E[8]));
    const float t0 = fmaf(E[0], x2, fmaf(E[3], y2, E[6]));
    const float t1 = fmaf(E[1], x2, fmaf(E[4], y2, E[7]));
    const float num = fmaf(x2, e0, fmaf(y2, e1, e2));
    const float den = fmaf(e0, e0, fmaf(e1, e1, fmaf(t0, t0, t1 * t1)));
    return (num * num) < (thr_sq * den);
}

static inline int pnp_inlier(const float* P, float X, float Y, float Z, float u, float v, float thr_sq) {
    const float pu = fmaf(P[0], X, fmaf(P[1], Y, fmaf(P[2], Z, P[3])));
    const float pv = fmaf(P[4], X, fmaf(P[5], Y, fmaf(P[6], Z, P[7])));
    const float w = fmaf(P[8], X, fmaf(P[9], Y, fmaf(P[10], Z, P[11])));
    const float du = fmaf(-u, w, pu);
    const float dv = fmaf(-v, w, pv);
    const float e2 = fmaf(du, du, dv * dv);
    return (w > 0.0f) && (e2 < thr_sq * (w * w));
}

/* one item: H hypotheses x n points -> count[H] */
void orc_score_essential(const float* E, const float* x1, const float* y1, const float* x2,
                         const float* y2, int H, int n, float thr_sq, int32_t* count) {
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; ++h) {
        int c = 0;
        for (int i = 0; i < n; ++i) c += sampson_inlier(E + 9 * (size_t)h, x1[i], y1[i], x2[i], y2[i], thr_sq);
        count[h] = c;
    }
}
void orc_score_pnp(const float* P, const float* X, const float* Y, const float* Z, const float* u,
                   const float* v, int H, int n, float thr_sq, int32_t* count) {
#pragma omp parallel for schedule(static)
    for (int h = 0; h < H; ++h) {
        int c = 0;
        for (int i = 0; i < n; ++i) c += pnp_inlier(P + 12 * (size_t)h, X[i], Y[i], Z[i], u[i], v[i], thr_sq);
        count[h] = c;
    }
}
void orc_select_best(const int32_t* count, int H, int32_t* best_idx, int32_t* best_count) {
    int bi = ORC_NO_INDEX, bc = -1;
    for (int h = 0; h < H; ++h) if (count[h] > bc) { bc = count[h]; bi = h; }
    *best_idx = bi; *best_count = (bi >= 0) ? bc : 0;
}
void orc_inlier_mask_essential(const float* E, const float* x1, const float* y1, const float* x2,
                               const float* y2, int n, int N, float thr_sq, uint64_t* mask) {
    const int words = (N + 63) / 64;
    for (int w = 0; w < words; ++w) mask[w] = 0;
    for (int i = 0; i < n; ++i)
        if (sampson_inlier(E, x1[i], y1[i], x2[i], y2[i], thr_sq)) mask[i >> 6] |= (uint64_t)1 << (i & 63);
}
void orc_inlier_mask_pnp(const float* P, const float* X, const float* Y, const float* Z, const float* u,
                         const float* v, int n, int N, float thr_sq, uint64_t* mask) {
    const int words = (N + 63) / 64;
    for (int w = 0; w < words; ++w) mask[w] = 0;
    for (int i = 0; i < n; ++i)
        if (pnp_inlier(P, X[i], Y[i], Z[i], u[i], v[i], thr_sq)) mask[i >> 6] |= (uint64_t)1 << (i & 63);
}

/* ------------------------------------------------------------------------------------------
 * (3) DLT triangulation, float64                                      SURVEY.md §8 row a-11
 * ------------------------------------------------------------------------------------------ */

/* cyclic Jacobi on a symmetric 4x4; V accumulates eigenvectors as columns. */
static void jacobi4(double M[4][4], double V[4][4]) {
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) V[i][j] = (i == j);
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) off += M[p][q] * M[p][q];
        if (off < 1e-300) break;
        for (int p = 0; p < 4; ++p) for (int q = p + 1; q < 4; ++q) {
            if (M[p][q] == 0.0) continue;
            const double theta = (M[q][q] - M[p][p]) / (2.0 * M[p][q]);
            const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
            for (int k = 0; k < 4; ++k) {
                const double mkp = M[k][p], mkq = M[k][q];
                M[k][p] = c * mkp - s * mkq; M[k][q] = s * mkp + c * mkq;
            }
            for (int k = 0; k < 4; ++k) {
                const double mpk = M[p][k], mqk = M[q][k];
                M[p][k] = c * mpk - s * mqk; M[q][k] = s * mpk + c * mqk;
            }
            for (int k = 0; k < 4; ++k) {
                const double vkp = V[k][p], vkq = V[k][q];
                V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq;
            }
        }
    }
}

/* one (P1,P2) pair, n points -> XYZ[n][3]; returns the number of points with positive depth
 * in both views among those selected by mask (NULL = all). */
int orc_triangulate(const float* P1, const float* P2, const float* x1, const float* y1,
                    const float* x2, const float* y2, int n, const uint64_t* mask, float* XYZ) {
    int front = 0;
    for (int i = 0; i < n; ++i) {
        double A[4][4];
        for (int c = 0; c < 4; ++c) {
            A[0][c] = (double)x1[i] * P1[8 + c] - P1[c];
            A[1][c] = (double)y1[i] * P1[8 + c] - P1[4 + c];
            A[2][c] = (double)x2[i] * P2[8 + c] - P2[c];
            A[3][c] = (double)y2[i] * P2[8 + c] - P2[4 + c];
        }
        double M[4][4], V[4][4];
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) {
            double s = 0.0;
            for (int r = 0; r < 4; ++r) s += A[r][a] * A[r][b];
            M[a][b] = s;
        }
        jacobi4(M, V);
        int m = 0;
        for (int k = 1; k < 4; ++k) if (M[k][k] < M[m][m]) m = k;
        const double X = V[0][m] / V[3][m], Y = V[1][m] / V[3][m], Z = V[2][m] / V[3][m];
        if (XYZ) { XYZ[3 * i] = (float)X; XYZ[3 * i + 1] = (float)Y; XYZ[3 * i + 2] = (float)Z; }
        const double z1 = P1[8] * X + P1[9] * Y + P1[10] * Z + P1[11];
        const double z2 = P2[8] * X + P2[9] * Y + P2[10] * Z + P2[11];
        const int sel = mask ? (int)((mask[i >> 6] >> (i & 63)) & 1) : 1;
        if (sel && z1 > 0.0 && z2 > 0.0) ++front;
    }
    return front;
}

/* ------------------------------------------------------------------------------------------
 * (4) projection with distortion + reprojection error, float64
 *     validate_pose.py:181-184 (project, truncate to int32) and :260-288 (error statistics).
 * ------------------------------------------------------------------------------------------ */

void orc_rodrigues(const double r[3], double R[9]) {
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-12) { /* first-order: I + [r]x */
        R[0] = 1; R[1] = -r[2]; R[2] = r[1]; R[3] = r[2]; R[4] = 1; R[5] = -r[0]; R[6] = -r[1]; R[7] = r[0]; R[8] = 1;
        return;
    }
    const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th, c = cos(th), s = sin(th), v = 1 - c;
    R[0] = c + kx * kx * v;      R[1] = kx * ky * v - kz * s; R[2] = kx * kz * v + ky * s;
    R[3] = ky * kx * v + kz * s; R[4] = c + ky * ky * v;      R[5] = ky * kz * v - kx * s;
    R[6] = kz * kx * v - ky * s; R[7] = kz * ky * v + kx * s; R[8] = c + kz * kz * v;
}

void orc_project_points(const float* obj, int n, const double rvec[3], const double tvec[3],
                        const double K4[4], const double dist5[5], double* uv, int32_t* uv_int) {
    double R[9];
    orc_rodrigues(rvec, R);
    const double k1 = dist5[0], k2 = dist5[1], p1 = dist5[2], p2 = dist5[3], k3 = dist5[4];
    for (int i = 0; i < n; ++i) {
        const double X = obj[3 * i], Y = obj[3 * i + 1], Z = obj[3 * i + 2];
        const double xc = R[0] * X + R[1] * Y + R[2] * Z + tvec[0];
        const double yc = R[3] * X + R[4] * Y + R[5] * Z + tvec[1];
        const double zc = R[6] * X + R[7] * Y + R[8] * Z + tvec[2];
        const double x = xc / zc, y = yc / zc, r2 = x * x + y * y;
        const double g = 1 + k1 * r2 + k2 * r2 * r2 + k3 * r2 * r2 * r2;
        const double xd = x * g + 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
        const double yd = y * g + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
        const double u = K4[0] * xd + K4[2], v = K4[1] * yd + K4[3];
        if (uv) { uv[2 * i] = u; uv[2 * i + 1] = v; }
        if (uv_int) { uv_int[2 * i] = (int32_t)u; uv_int[2 * i + 1] = (int32_t)v; } /* C cast truncates, as astype(int32) */
    }
}

void orc_reprojection_error(const float* obj, const float* img, int n, const double rvec[3],
                            const double tvec[3], const double K4[4], const double dist5[5],
                            double* err, double stats[5]) {
    double* uv = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
    orc_project_points(obj, n, rvec, tvec, K4, dist5, uv, NULL);
    double sum = 0, mx = -INFINITY, mn = INFINITY;
    for (int i = 0; i < n; ++i) {
        const double du = uv[2 * i] - img[2 * i], dv = uv[2 * i + 1] - img[2 * i + 1];
        const double e = sqrt(du * du + dv * dv);
        if (err) err[i] = e;
        sum += e; if (e > mx) mx = e; if (e < mn) mn = e;
    }
    const double mean = n ? sum / n : 0.0;
    double var = 0;
    for (int i = 0; i < n; ++i) {
        const double du = uv[2 * i] - img[2 * i], dv = uv[2 * i + 1] - img[2 * i + 1];
        const double e = sqrt(du * du + dv * dv) - mean;
        var += e * e;
    }
    stats[0] = mean; stats[1] = n ? sqrt(var / n) : 0.0; stats[2] = mx; stats[3] = mn; stats[4] = (double)n;
    free(uv);
}
