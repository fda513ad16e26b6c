// fivepoint_core.hpp — Nister 5-point solver for ONE minimal sample (float64), shared by the device
// kernel (fivepoint.hip) and a host debug harness; see fivepoint.hip for the algorithm summary.
#pragma once
#include <math.h>
#include <stdint.h>
#if defined(__HIPCC__)
#define PH_HD __host__ __device__
#else
#define PH_HD
#endif
#define PH_HD_CONST static constexpr

namespace ph5 {

// ---------------------------------------------------------------- polynomial tables (tools/gen_poly_tables.py)
// degree-1 monomials: x y z 1; degree-2: xx xy xz yy yz zz x y z 1; degree-3 graded (20).
PH_HD_CONST int8_t kMul11[4][4] = {{0, 1, 2, 6}, {1, 3, 4, 7}, {2, 4, 5, 8}, {6, 7, 8, 9}};
PH_HD_CONST int8_t kMul21[10][4] = {{0, 1, 2, 10}, {1, 3, 4, 11}, {2, 4, 5, 12}, {3, 6, 7, 13}, {4, 7, 8, 14},
                                     {5, 8, 9, 15}, {10, 11, 12, 16}, {11, 13, 14, 17}, {12, 14, 15, 18}, {16, 17, 18, 19}};
// column c of the elimination matrix (Nistér's order x3 y3 x2y xy2 x2z x2 y2z y2 xyz xy | xz2 xz x yz2 yz y z3 z2 z 1)
PH_HD_CONST int8_t kNister[20] = {0, 6, 1, 3, 2, 10, 7, 13, 4, 11, 5, 12, 16, 8, 14, 17, 9, 15, 18, 19};

PH_HD inline void mul11_acc(const double* a, const double* b, double s, double* out) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out[kMul11[i][j]] += s * a[i] * b[j];
}
PH_HD inline void mul21_acc(const double* a, const double* b, double s, double* out) {
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 4; ++j) out[kMul21[i][j]] += s * a[i] * b[j];
}
// c (ascending, length na+nb-1) += s * a * b
PH_HD inline void pmul_acc(const double* a, int na, const double* b, int nb, double s, double* c) {
    for (int i = 0; i < na; ++i)
        for (int j = 0; j < nb; ++j) c[i + j] += s * a[i] * b[j];
}

struct cplx { double re, im; };
PH_HD inline cplx cmul(cplx a, cplx b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// 1/x for the root iteration.  On the device: v_rcp_f64 (about 2^-27 relative) and two Newton steps, ~6 instructions
// instead of the ~25 of an IEEE-correct division (div_scale / div_fmas / div_fixup); every root is polished afterwards.
PH_HD inline double fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
#else
    return 1.0 / x;
#endif
}
// 1/sqrt(x) the same way: v_rsq_f64 and two Newton steps
PH_HD inline double fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rsq(x);
    r = r * fma(-0.5 * x * r, r, 1.5);
    r = r * fma(-0.5 * x * r, r, 1.5);
    return r;
#else
    return 1.0 / sqrt(x);
#endif
}
PH_HD inline cplx cinv(cplx b) {                       // 1/b with ONE real reciprocal
    const double r = fast_rcp(b.re * b.re + b.im * b.im);
    return {b.re * r, -b.im * r};
}
PH_HD inline cplx cdiv(cplx a, cplx b) { return cmul(a, cinv(b)); }


// pts[r] = (x1, y1, x2, y2) of correspondence r; eo = 90 floats (10 slots), pre-zeroed by the caller.
// Returns the number of real solutions written.
// Everything the root stage needs about one minimal sample.
struct Poly5 {
    double c[11];                        // det B(z), ascending powers; degree n
    double Bx[3][4], By[3][4], B1[3][5]; // rows k, l, m of B(z): x, y and constant parts
    double basis[4][9];                  // null-space basis X, Y, Z, W (E = xX + yY + zZ + W)
    int n;                               // polynomial degree, 0 = degenerate sample (no solutions)
};

// Working arrays of the polynomial stage.  They are indexed dynamically (pivoting), so on the device they
// would live in private memory (HBM-backed); the kernel puts one Work5 per thread in LDS instead.
// sizeof(Work5) = 8 * 373 bytes: an odd number of 8-byte words, so that the lanes of a wave touching the
// same field hit distinct LDS banks.  Q (the 5x9 epipolar system) is dead once the null-space basis is out and
// shares its storage with EEt, which is first written after that.
struct Work5 {
    union {
        double Q[5][9];
        double EEt[3][3][10];
    };
    double Ep[9][4];
    double tr[10];
    double A[10][20];
    double tmp[20], d2[10];
    int perm[10];
    double pad[2];
};
static_assert(sizeof(Work5) == 8 * 373, "Work5 must be an odd number of 8-byte words");

// Sections 1-8: minimal sample -> polynomial.  pts[r] = (x1, y1, x2, y2).
PH_HD inline bool build_polynomial(const double (&pts)[5][4], Poly5& P, Work5& w) {
    double (&Q)[5][9] = w.Q;
    int (&perm)[10] = w.perm;
    double (&Ep)[9][4] = w.Ep;
    double (&EEt)[3][3][10] = w.EEt;
    double (&tr)[10] = w.tr;
    double (&A)[10][20] = w.A;
    double (&tmp)[20] = w.tmp;
    double (&d2)[10] = w.d2;
    P.n = 0;
    double (&basis)[4][9] = P.basis;
    double (&Bx)[3][4] = P.Bx;
    double (&By)[3][4] = P.By;
    double (&B1)[3][5] = P.B1;
    double (&c)[11] = P.c;
    // ---- 1. epipolar rows q . vec(E) = x2' E x1
    for (int r = 0; r < 5; ++r) {
        const double a = pts[r][0], c = pts[r][1], d = pts[r][2], e = pts[r][3];
        Q[r][0] = d * a; Q[r][1] = d * c; Q[r][2] = d; Q[r][3] = e * a; Q[r][4] = e * c; Q[r][5] = e;
        Q[r][6] = a; Q[r][7] = c; Q[r][8] = 1.0;
    }
    // ---- 2. null space: Gauss-Jordan with full pivoting, then modified Gram-Schmidt
    for (int j = 0; j < 9; ++j) perm[j] = j;
    for (int r = 0; r < 5; ++r) {
        int pi = r, pj = r;
        double best = -1.0;
        for (int i = r; i < 5; ++i)
            for (int j = r; j < 9; ++j) { const double v = fabs(Q[i][perm[j]]); if (v > best) { best = v; pi = i; pj = j; } }
        if (!(best > 1e-12)) return false;                                   // degenerate sample
        if (pi != r) for (int j = 0; j < 9; ++j) { const double t = Q[r][j]; Q[r][j] = Q[pi][j]; Q[pi][j] = t; }
        { const int t = perm[r]; perm[r] = perm[pj]; perm[pj] = t; }
        const double inv = 1.0 / Q[r][perm[r]];
        for (int j = 0; j < 9; ++j) Q[r][j] *= inv;
        for (int i = 0; i < 5; ++i) {
            if (i == r) continue;
            const double f = Q[i][perm[r]];
            for (int j = 0; j < 9; ++j) Q[i][j] -= f * Q[r][j];
        }
    }
    for (int k = 0; k < 4; ++k) {
        for (int j = 0; j < 9; ++j) basis[k][j] = 0.0;
        basis[k][perm[5 + k]] = 1.0;
        for (int i = 0; i < 5; ++i) basis[k][perm[i]] = -Q[i][perm[5 + k]];
    }
    for (int k = 0; k < 4; ++k) {
        for (int p = 0; p < k; ++p) {
            double d = 0.0;
            for (int j = 0; j < 9; ++j) d += basis[k][j] * basis[p][j];
            for (int j = 0; j < 9; ++j) basis[k][j] -= d * basis[p][j];
        }
        double n2 = 0.0;
        for (int j = 0; j < 9; ++j) n2 += basis[k][j] * basis[k][j];
        const double inv = 1.0 / sqrt(n2);
        for (int j = 0; j < 9; ++j) basis[k][j] *= inv;
    }
    // ---- 3. E entries as linear polynomials in (x, y, z, 1)
    for (int e = 0; e < 9; ++e)
        for (int k = 0; k < 4; ++k) Ep[e][k] = basis[k][e];
    // ---- 4. EE' (symmetric, 6 entries) and its trace, degree 2
    for (int r = 0; r < 3; ++r)
        for (int c = r; c < 3; ++c) {
            for (int k = 0; k < 10; ++k) EEt[r][c][k] = 0.0;
            for (int k = 0; k < 3; ++k) mul11_acc(Ep[3 * r + k], Ep[3 * c + k], 1.0, EEt[r][c]);
        }
    for (int r = 1; r < 3; ++r)
        for (int c = 0; c < r; ++c)
            for (int k = 0; k < 10; ++k) EEt[r][c][k] = EEt[c][r][k];
    for (int k = 0; k < 10; ++k) tr[k] = EEt[0][0][k] + EEt[1][1][k] + EEt[2][2][k];
    // ---- 5. the 10 cubic constraints, columns in Nistér's order
    {
        auto put = [&](int row) { for (int c = 0; c < 20; ++c) A[row][c] = tmp[kNister[c]]; };
        for (int k = 0; k < 20; ++k) tmp[k] = 0.0;
        const int cof[3][4] = {{1, 5, 2, 4}, {2, 3, 0, 5}, {0, 4, 1, 3}};   // det E = sum_c (minor_c) * E[2][c]
        for (int c = 0; c < 3; ++c) {
            for (int k = 0; k < 10; ++k) d2[k] = 0.0;
            mul11_acc(Ep[cof[c][0]], Ep[cof[c][1]], 1.0, d2);
            mul11_acc(Ep[cof[c][2]], Ep[cof[c][3]], -1.0, d2);
            mul21_acc(d2, Ep[6 + c], 1.0, tmp);
        }
        put(0);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) {
                for (int k = 0; k < 20; ++k) tmp[k] = 0.0;
                for (int k = 0; k < 3; ++k) mul21_acc(EEt[r][k], Ep[3 * k + c], 2.0, tmp);
                mul21_acc(tr, Ep[3 * r + c], -1.0, tmp);
                put(1 + 3 * r + c);
            }
    }
    // ---- 6. Gauss-Jordan on the left 10x10 block (partial pivoting)
    for (int c = 0; c < 10; ++c) {
        int p = c;
        double best = fabs(A[c][c]);
        for (int i = c + 1; i < 10; ++i) { const double v = fabs(A[i][c]); if (v > best) { best = v; p = i; } }
        if (!(best > 1e-14)) return false;
        if (p != c) for (int j = 0; j < 20; ++j) { const double t = A[c][j]; A[c][j] = A[p][j]; A[p][j] = t; }
        const double inv = 1.0 / A[c][c];
        for (int j = c; j < 20; ++j) A[c][j] *= inv;
        for (int i = 0; i < 10; ++i) {
            if (i == c) continue;
            const double f = A[i][c];
            if (f != 0.0) for (int j = c; j < 20; ++j) A[i][j] -= f * A[c][j];
        }
    }
    // ---- 7. k = e - z f, l = g - z h, m = i - z j  (ascending powers of z; x, y parts degree 3, 1 part degree 4)
    for (int q = 0; q < 3; ++q) {
        const double* p = &A[4 + 2 * q][10];
        const double* r = &A[5 + 2 * q][10];
        Bx[q][3] = -r[0]; Bx[q][2] = p[0] - r[1]; Bx[q][1] = p[1] - r[2]; Bx[q][0] = p[2];
        By[q][3] = -r[3]; By[q][2] = p[3] - r[4]; By[q][1] = p[4] - r[5]; By[q][0] = p[5];
        B1[q][4] = -r[6]; B1[q][3] = p[6] - r[7]; B1[q][2] = p[7] - r[8]; B1[q][1] = p[8] - r[9]; B1[q][0] = p[9];
    }
    // ---- 8. det B(z), degree 10
    for (int k = 0; k < 11; ++k) c[k] = 0.0;
    {
        double t7[8], t6[7];
        for (int k = 0; k < 8; ++k) t7[k] = 0.0;
        pmul_acc(By[1], 4, B1[2], 5, 1.0, t7); pmul_acc(B1[1], 5, By[2], 4, -1.0, t7);
        pmul_acc(Bx[0], 4, t7, 8, 1.0, c);
        for (int k = 0; k < 8; ++k) t7[k] = 0.0;
        pmul_acc(Bx[1], 4, B1[2], 5, 1.0, t7); pmul_acc(B1[1], 5, Bx[2], 4, -1.0, t7);
        pmul_acc(By[0], 4, t7, 8, -1.0, c);
        for (int k = 0; k < 7; ++k) t6[k] = 0.0;
        pmul_acc(Bx[1], 4, By[2], 4, 1.0, t6); pmul_acc(By[1], 4, Bx[2], 4, -1.0, t6);
        pmul_acc(B1[0], 5, t6, 7, 1.0, c);
    }
    int n = 10;
    double cmax = 0.0;
    for (int k = 0; k <= 10; ++k) cmax = fmax(cmax, fabs(c[k]));
    if (!(cmax > 0.0) || !(cmax < HUGE_VAL)) return false;
    while (n > 0 && fabs(c[n]) <= 1e-30 * cmax) --n;   // a tiny leading coefficient is a legitimate huge root: keep it
    if (n < 1) return false;
    P.n = n;
    return true;
}

// Iteration control: the roots are polished by Newton steps on the real axis afterwards (root_to_E), so the
// simultaneous iteration only has to separate them; 1e-13 relative is far inside Newton's basin.
PH_HD_CONST int kAberthMaxIter = 60;
PH_HD_CONST double kAberthTol = 1e-13;
// Ill-conditioned roots jitter at 1e-13..1e-11 relative for ever (rounding noise of the degree-10 Horner sums): about
// 2 % of the samples never meet 1e-13, and one such sample keeps its whole wave iterating.  The tolerance therefore
// relaxes by a decade every three iterations from the 13th on (the typical sample is done by then): 1e-12 at 13, 1e-11
// at 16, 1e-9 at 22 — every accepted root is still far inside the basin of the Newton polish that follows.
PH_HD inline double aberth_tol(int it) {
    double t = kAberthTol;
    for (int k = 13; k <= it; k += 3) t *= 10.0;
    return t;
}

// Aberth-Ehrlich start values: n points on a circle of the geometric-mean root radius.
PH_HD inline cplx aberth_start(const Poly5& P, int k) {
    double rad = pow(fabs(P.c[0] / P.c[P.n]) + 1e-300, 1.0 / P.n);
    rad = fmin(fmax(rad, 1e-3), 1e3);
    const double a = 6.283185307179586 * k / P.n + 0.4;
    return {rad * cos(a), rad * sin(a)};
}

// One Aberth correction for root zk given sum_j 1/(zk - zj) over the other roots; returns the step w (z -= w).
PH_HD inline cplx aberth_step(const Poly5& P, cplx zk, cplx sum) {
    const int n = P.n;
    cplx p = {P.c[n], 0.0}, dp = {0.0, 0.0};
    for (int j = n - 1; j >= 0; --j) {
        dp = cmul(dp, zk); dp.re += p.re; dp.im += p.im;
        p = cmul(p, zk); p.re += P.c[j];
    }
    if (p.re == 0.0 && p.im == 0.0) return {0.0, 0.0};
    const cplx nw = cdiv(p, dp);                          // Newton step
    const cplx den = {1.0 - (nw.re * sum.re - nw.im * sum.im), -(nw.re * sum.im + nw.im * sum.re)};
    const cplx w = cdiv(nw, den);
    if (w.re == w.re && w.im == w.im) return w;
    return {0.0, 0.0};
}

// A converged root -> the essential matrix it defines (unit Frobenius norm, f32); false when the root is not real.
PH_HD inline bool root_to_E(const Poly5& P, cplx z, float* e9) {
    const int n = P.n;
    if (!(fabs(z.im) < 1e-6 * fmax(1.0, fabs(z.re)))) return false;
    double zr = z.re;
    for (int it = 0; it < 6; ++it) {                       // polish on the real axis
        double p = P.c[n], dp = 0.0;
        for (int j = n - 1; j >= 0; --j) { dp = dp * zr + p; p = p * zr + P.c[j]; }
        if (dp != 0.0) zr -= p / dp;
    }
    {   // a complex pair with a tiny imaginary part is not a solution: require p(zr) = 0 to rounding
        double p = P.c[n], mag = fabs(P.c[n]);
        const double az = fabs(zr);
        for (int j = n - 1; j >= 0; --j) { p = p * zr + P.c[j]; mag = mag * az + fabs(P.c[j]); }
        if (!(fabs(p) <= 1e-10 * mag)) return false;
    }
    double Bz[3][3];
    for (int q = 0; q < 3; ++q) {
        Bz[q][0] = ((P.Bx[q][3] * zr + P.Bx[q][2]) * zr + P.Bx[q][1]) * zr + P.Bx[q][0];
        Bz[q][1] = ((P.By[q][3] * zr + P.By[q][2]) * zr + P.By[q][1]) * zr + P.By[q][0];
        Bz[q][2] = (((P.B1[q][4] * zr + P.B1[q][3]) * zr + P.B1[q][2]) * zr + P.B1[q][1]) * zr + P.B1[q][0];
    }
    double v[3] = {0, 0, 0}, vbest = -1.0;                 // null vector = largest cross product of two rows
    for (int q = 0; q < 3; ++q) {
        const double* a = Bz[q];
        const double* d = Bz[(q + 1) % 3];
        const double w0 = a[1] * d[2] - a[2] * d[1], w1 = a[2] * d[0] - a[0] * d[2], w2 = a[0] * d[1] - a[1] * d[0];
        const double nn = w0 * w0 + w1 * w1 + w2 * w2;
        if (nn > vbest) { vbest = nn; v[0] = w0; v[1] = w1; v[2] = w2; }
    }
    if (!(fabs(v[2]) > 1e-14 * sqrt(vbest))) return false;
    const double xr = v[0] / v[2], yr = v[1] / v[2];
    double E[9], n2 = 0.0;
    for (int e = 0; e < 9; ++e) {
        E[e] = xr * P.basis[0][e] + yr * P.basis[1][e] + zr * P.basis[2][e] + P.basis[3][e];
        n2 += E[e] * E[e];
    }
    if (!(n2 > 0.0) || !(n2 < HUGE_VAL)) return false;
    const double inv = 1.0 / sqrt(n2);
    for (int e = 0; e < 9; ++e) e9[e] = (float)(E[e] * inv);
    return true;
}

// ---- fixed-length forms for the device's root stage: the coefficients above the degree are ZERO (c[k] = 0 for k > n),
// so every Horner loop runs the full length 10 under constant indices and the polynomial stays in registers (a loop that
// starts at the runtime degree indexes c[] dynamically, which puts it in scratch memory: one L2 round trip per Horner step).
// A leading zero contributes exact zeros, so the values are those of the loops that start at n.
PH_HD inline double lead_coeff(const double (&c)[11], int n) {
    double l = c[0];
#pragma unroll
    for (int k = 1; k <= 10; ++k) l = (k == n) ? c[k] : l;
    return l;
}
PH_HD inline cplx aberth_step10(const double (&c)[11], cplx zk, cplx sum) {
    // p(z) and p'(z) of a REAL polynomial at a complex point by synthetic division with the real quadratic
    // x^2 - r x + s (r = 2 Re z, s = |z|^2) that has z as a root: two real fmas per coefficient instead of a complex
    // Horner step (eleven).  p(x) = (x^2 - r x + s) q(x) + b1 x + (b0 - r b1)  =>  p(z) = b1 z + b0 - r b1, and
    // p'(z) = (2z - r) q(z) + b1 with q evaluated the same way (2z - r = 2i Im z).
    const double r = 2.0 * zk.re, sq = zk.re * zk.re + zk.im * zk.im;
    double b[11];
    b[10] = c[10];
    b[9] = c[9] + r * b[10];
#pragma unroll
    for (int k = 8; k >= 0; --k) b[k] = c[k] + r * b[k + 1] - sq * b[k + 2];
    // q(x) = sum_{k=0..8} b[k+2] x^k
    double d1 = b[10], d0 = b[9] + r * d1;                 // d_8, d_7 of the second division
#pragma unroll
    for (int k = 6; k >= 0; --k) { const double t = b[k + 2] + r * d0 - sq * d1; d1 = d0; d0 = t; }
    // after the loop: d0 = d_0, d1 = d_1  =>  q(z) = d1 z + d0 - r d1
    const cplx p = {b[1] * zk.re + b[0] - r * b[1], b[1] * zk.im};
    const cplx qz = {d1 * zk.re + d0 - r * d1, d1 * zk.im};
    const double tz = 2.0 * zk.im;                         // (2z - r) = i * tz
    const cplx dp = {b[1] - tz * qz.im, tz * qz.re};
    if (p.re == 0.0 && p.im == 0.0) return {0.0, 0.0};
    const cplx nw = cdiv(p, dp);                          // Newton step
    const cplx den = {1.0 - (nw.re * sum.re - nw.im * sum.im), -(nw.re * sum.im + nw.im * sum.re)};
    const cplx w = cdiv(nw, den);
    if (w.re == w.re && w.im == w.im) return w;
    return {0.0, 0.0};
}
// root_to_E with the polynomial in registers and the rest of the record read where it lies (constant indices)
PH_HD inline bool root_to_E10(const double (&c)[11], const Poly5& P, cplx z, float* e9) {
    if (!(fabs(z.im) < 1e-6 * fmax(1.0, fabs(z.re)))) return false;
    double zr = z.re;
#pragma unroll
    for (int it = 0; it < 6; ++it) {                       // polish on the real axis
        double p = c[10], dp = 0.0;
#pragma unroll
        for (int j = 9; j >= 0; --j) { dp = dp * zr + p; p = p * zr + c[j]; }
        if (dp != 0.0) zr -= p * fast_rcp(dp);          // a Newton step forgives its quotient's last bits
    }
    {   // a complex pair with a tiny imaginary part is not a solution: require p(zr) = 0 to rounding
        double p = c[10], mag = fabs(c[10]);
        const double az = fabs(zr);
#pragma unroll
        for (int j = 9; j >= 0; --j) { p = p * zr + c[j]; mag = mag * az + fabs(c[j]); }
        if (!(fabs(p) <= 1e-10 * mag)) return false;
    }
    double Bz[3][3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        Bz[q][0] = ((P.Bx[q][3] * zr + P.Bx[q][2]) * zr + P.Bx[q][1]) * zr + P.Bx[q][0];
        Bz[q][1] = ((P.By[q][3] * zr + P.By[q][2]) * zr + P.By[q][1]) * zr + P.By[q][0];
        Bz[q][2] = (((P.B1[q][4] * zr + P.B1[q][3]) * zr + P.B1[q][2]) * zr + P.B1[q][1]) * zr + P.B1[q][0];
    }
    double v0 = 0, v1 = 0, v2 = 0, vbest = -1.0;           // null vector = largest cross product of two rows
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const double* a = Bz[q];
        const double* d = Bz[(q + 1) % 3];
        const double w0 = a[1] * d[2] - a[2] * d[1], w1 = a[2] * d[0] - a[0] * d[2], w2 = a[0] * d[1] - a[1] * d[0];
        const double nn = w0 * w0 + w1 * w1 + w2 * w2;
        if (nn > vbest) { vbest = nn; v0 = w0; v1 = w1; v2 = w2; }
    }
    if (!(v2 * v2 > 1e-28 * vbest)) return false;          // |v2| > 1e-14 |v|
    const double rv = fast_rcp(v2), xr = v0 * rv, yr = v1 * rv;
    double E[9], n2 = 0.0;
#pragma unroll
    for (int e = 0; e < 9; ++e) {
        E[e] = xr * P.basis[0][e] + yr * P.basis[1][e] + zr * P.basis[2][e] + P.basis[3][e];
        n2 += E[e] * E[e];
    }
    if (!(n2 > 0.0) || !(n2 < HUGE_VAL)) return false;
    const double inv = fast_rsqrt(n2);                     // the result is stored as f32
#pragma unroll
    for (int e = 0; e < 9; ++e) e9[e] = (float)(E[e] * inv);
    return true;
}

// Sequential reference of the whole solver (host debug harness; the device splits it over two kernels).
// eo = 90 floats (10 slots, slot k = root k), pre-zeroed by the caller.  Returns the number of real solutions.
PH_HD inline int solve_fivepoint(const double (&pts)[5][4], float* eo) {
    Poly5 P;
    Work5 w;
    if (!build_polynomial(pts, P, w)) return 0;
    const int n = P.n;
    cplx z[10];
    for (int k = 0; k < n; ++k) z[k] = aberth_start(P, k);
    for (int it = 0; it < kAberthMaxIter; ++it) {
        cplx w[10];
        double worst = 0.0;
        for (int k = 0; k < n; ++k) {
            cplx sum = {0.0, 0.0};
            for (int j = 0; j < n; ++j) {
                if (j == k) continue;
                const cplx d = {z[k].re - z[j].re, z[k].im - z[j].im};
                const cplx inv = cinv(d);
                sum.re += inv.re; sum.im += inv.im;
            }
            w[k] = aberth_step(P, z[k], sum);
        }
        for (int k = 0; k < n; ++k) {                     // simultaneous (Jacobi) update, as the device does
            z[k].re -= w[k].re; z[k].im -= w[k].im;
            worst = fmax(worst, (fabs(w[k].re) + fabs(w[k].im)) / (fabs(z[k].re) + fabs(z[k].im) + 1e-30));
        }
        if (worst < aberth_tol(it)) break;
    }
    int ns = 0;
    for (int k = 0; k < n; ++k) ns += root_to_E(P, z[k], eo + 9 * k) ? 1 : 0;
    return ns;
}

}  // namespace ph5
