// P3P minimal solver core (host + device): absolute camera pose from three 3-D points and their bearings.
// No reference counterpart (the reference calls cv2.solvePnP on all points, main.py:101-104); this is the
// minimal solver north_star's "P3P hypotheses" need (SURVEY.md §8f item 1, §8d cfg 5).
//
// Method (Grunert's substitution, derived here rather than transcribed):
//   depths s1, s2 = u s1, s3 = v s1 along the unit bearings f1, f2, f3; the three law-of-cosines equations
//       s2^2 + s3^2 - 2 s2 s3 ca = a2      (a2 = |X2 - X3|^2, ca = f2.f3)
//       s1^2 + s3^2 - 2 s1 s3 cb = b2      (b2 = |X1 - X3|^2, cb = f1.f3)
//       s1^2 + s2^2 - 2 s1 s2 cg = c2      (c2 = |X1 - X2|^2, cg = f1.f2)
//   give, after eliminating s1^2,  u = N(v) / D(v)  with  N = (a2-c2) Q - b2 (v^2 - 1),  D = 2 b2 (cg - ca v),
//   Q = 1 - 2 cb v + v^2,  and the quartic   b2 (D^2 + N^2 - 2 cg N D) - c2 Q D^2 = 0   in v.
//   Its four roots come from an Aberth-Ehrlich iteration; every real positive root is turned into depths, polished
//   by two Gauss-Newton steps on the three equations, checked against them, and converted to (R, t) by aligning the
//   two triangles' orthonormal frames.  Slot k of the output belongs to root k; a slot without a solution is zero.
#pragma once
#include <cmath>
#include <cstdint>

#include "fivepoint_core.hpp"   // PH_HD, ph5::cplx helpers

namespace p3p {

using ph5::cinv;
using ph5::cmul;
using ph5::cplx;
using ph5::fast_rcp;      // on the device: v_rcp_f64 / v_rsq_f64 + two Newton steps (~2^-50) instead of IEEE division / sqrt
using ph5::fast_rsqrt;    // (~25 dependent instructions each; this solver had about sixty of them per sample); host: exact

PH_HD_CONST int kMaxIter = 40;
PH_HD_CONST double kTol = 1e-14;

PH_HD inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
PH_HD inline void cross3(const double* a, const double* b, double* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

// orthonormal frame (columns e1, e2, e3) of the triangle p1, p2, p3; false when the points are collinear
PH_HD inline bool triangle_frame(const double* p1, const double* p2, const double* p3, double (&E)[3][3]) {
    double d12[3], d13[3];
    for (int k = 0; k < 3; ++k) { d12[k] = p2[k] - p1[k]; d13[k] = p3[k] - p1[k]; }
    const double q1 = dot3(d12, d12);
    if (!(q1 > 0.0)) return false;
    const double r1 = fast_rsqrt(q1);
    for (int k = 0; k < 3; ++k) E[0][k] = d12[k] * r1;
    double w[3];
    cross3(E[0], d13, w);
    const double q3 = dot3(w, w);
    if (!(q3 > 1e-24 * dot3(d13, d13))) return false;      // |w| > 1e-12 |d13|
    const double r3 = fast_rsqrt(q3);
    for (int k = 0; k < 3; ++k) E[2][k] = w[k] * r3;
    cross3(E[2], E[0], E[1]);
    return true;
}

// X: three world points; f: three unit bearings.  Rt: [4][12] row-major [R | t] per root slot (zeroed when the slot
// holds no solution).  Returns the number of solutions.
PH_HD inline int solve(const double (&X)[3][3], const double (&f)[3][3], double (*Rt)[12]) {
    for (int k = 0; k < 4; ++k)
        for (int e = 0; e < 12; ++e) Rt[k][e] = 0.0;
    double d[3];
    for (int k = 0; k < 3; ++k) d[k] = X[1][k] - X[2][k];
    const double a2 = dot3(d, d);
    for (int k = 0; k < 3; ++k) d[k] = X[0][k] - X[2][k];
    const double b2 = dot3(d, d);
    for (int k = 0; k < 3; ++k) d[k] = X[0][k] - X[1][k];
    const double c2 = dot3(d, d);
    const double scale = a2 + b2 + c2;
    if (!(a2 > 0.0) || !(b2 > 0.0) || !(c2 > 0.0) || !(scale < HUGE_VAL)) return 0;
    double EX[3][3];
    if (!triangle_frame(X[0], X[1], X[2], EX)) return 0;
    const double ca = dot3(f[1], f[2]), cb = dot3(f[0], f[2]), cg = dot3(f[0], f[1]);

    // N(v), D(v), Q(v) and the quartic
    const double m = a2 - c2;
    const double N[3] = {m + b2, -2.0 * m * cb, m - b2};
    const double D[2] = {2.0 * b2 * cg, -2.0 * b2 * ca};
    const double Q[3] = {1.0, -2.0 * cb, 1.0};
    const double D2[3] = {D[0] * D[0], 2.0 * D[0] * D[1], D[1] * D[1]};
    double c[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        c[i] += b2 * D2[i];
        for (int j = 0; j < 3; ++j) {
            c[i + j] += b2 * N[i] * N[j];
            c[i + j] -= c2 * Q[i] * D2[j];
        }
        for (int j = 0; j < 2; ++j) c[i + j] -= 2.0 * cg * b2 * N[i] * D[j];
    }
    double cmax = 0.0;
    for (int k = 0; k <= 4; ++k) cmax = fmax(cmax, fabs(c[k]));
    if (!(cmax > 0.0) || !(cmax < HUGE_VAL)) return 0;
    {
        const double rc = fast_rcp(cmax);
        for (int k = 0; k <= 4; ++k) c[k] *= rc;
    }
    constexpr int n = 4;                       // fixed degree keeps every array below in registers on the device
    if (!(fabs(c[4]) > 1e-30)) return 0;       // a vanishing leading coefficient (measure-zero configuration): no hypotheses

    // Aberth-Ehrlich on the four roots
    cplx z[4];
    {
        // fourth root of |c0 / c4|, clamped to [1e-3, 1e3]: two f32 square roots (start values need no more)
        const float x4 = (float)fmin(fmax(fabs(c[0] * fast_rcp(c[n])), 1e-12), 1e12);
        const double rad = (double)sqrtf(sqrtf(x4));
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const double ang = 6.283185307179586 * k / n + 0.4;
            z[k] = {rad * cos(ang), rad * sin(ang)};
        }
    }
    for (int it = 0; it < kMaxIter; ++it) {
        cplx w[4];
#pragma unroll
        for (int k = 0; k < n; ++k) {
            cplx p = {c[n], 0.0}, dp = {0.0, 0.0};
#pragma unroll
            for (int j = n - 1; j >= 0; --j) {
                dp = cmul(dp, z[k]); dp.re += p.re; dp.im += p.im;
                p = cmul(p, z[k]); p.re += c[j];
            }
            w[k] = {0.0, 0.0};
            if (p.re == 0.0 && p.im == 0.0) continue;
            cplx sum = {0.0, 0.0};
#pragma unroll
            for (int j = 0; j < n; ++j) {
                if (j == k) continue;
                const cplx inv = cinv({z[k].re - z[j].re, z[k].im - z[j].im});
                sum.re += inv.re; sum.im += inv.im;
            }
            const cplx nw = cmul(p, cinv(dp));
            const cplx den = {1.0 - (nw.re * sum.re - nw.im * sum.im), -(nw.re * sum.im + nw.im * sum.re)};
            const cplx s = cmul(nw, cinv(den));
            if (s.re == s.re && s.im == s.im) w[k] = s;
        }
        bool settled = true;                       // every root's relative change below kTol
#pragma unroll
        for (int k = 0; k < n; ++k) {
            z[k].re -= w[k].re; z[k].im -= w[k].im;
            settled = settled && (fabs(w[k].re) + fabs(w[k].im) < kTol * (fabs(z[k].re) + fabs(z[k].im) + 1e-30));
        }
        if (settled) break;
    }

    int ns = 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
        if (!(fabs(z[k].im) < 1e-4 * fmax(1.0, fabs(z[k].re)))) continue;
        const double v = z[k].re;
        if (!(v > 0.0)) continue;
        const double Qv = (v - 2.0 * cb) * v + 1.0;
        const double Nv = (N[2] * v + N[1]) * v + N[0], Dv = D[1] * v + D[0];
        double u;
        if (fabs(Dv) > 1e-9 * b2) {
            u = Nv * fast_rcp(Dv);
        } else {   // D(v) = 0: u from  b2 (1 + u^2 - 2 u cg) = c2 Q(v),  the root that also satisfies the first equation
            const double disc = cg * cg - 1.0 + c2 * Qv * fast_rcp(b2);
            if (!(disc >= 0.0)) continue;
            const double r = sqrt(disc), u1 = cg + r, u2 = cg - r;
            const double e1 = fabs(b2 * (u1 * u1 + v * v - 2.0 * u1 * v * ca) - a2 * Qv);
            const double e2 = fabs(b2 * (u2 * u2 + v * v - 2.0 * u2 * v * ca) - a2 * Qv);
            u = (e1 <= e2) ? u1 : u2;
        }
        if (!(u > 0.0) || !(Qv > 0.0)) continue;
        double s[3];
        {
            const double t = b2 * fast_rcp(Qv);            // s1 = sqrt(b2 / Q(v)); t > 0 here
            s[0] = t * fast_rsqrt(t); s[1] = u * s[0]; s[2] = v * s[0];
        }
        // two Gauss-Newton steps on the three law-of-cosines equations, then the residual test
        double res = 0.0;
        for (int it = 0; it < 3; ++it) {
            const double g0 = s[1] * s[1] + s[2] * s[2] - 2.0 * s[1] * s[2] * ca - a2;
            const double g1 = s[0] * s[0] + s[2] * s[2] - 2.0 * s[0] * s[2] * cb - b2;
            const double g2 = s[0] * s[0] + s[1] * s[1] - 2.0 * s[0] * s[1] * cg - c2;
            res = fmax(fabs(g0), fmax(fabs(g1), fabs(g2)));
            if (it == 2) break;
            const double J01 = 2.0 * (s[1] - s[2] * ca), J02 = 2.0 * (s[2] - s[1] * ca);
            const double J10 = 2.0 * (s[0] - s[2] * cb), J12 = 2.0 * (s[2] - s[0] * cb);
            const double J20 = 2.0 * (s[0] - s[1] * cg), J21 = 2.0 * (s[1] - s[0] * cg);
            // J = [0 J01 J02; J10 0 J12; J20 J21 0]
            const double det = J01 * J12 * J20 + J02 * J10 * J21;
            if (!(det * det > 1e-24 * scale * scale * scale)) continue;     // |det| > 1e-12 scale^1.5
            const double id = fast_rcp(det);
            // adjugate of J times g
            const double d0 = (-J12 * J21) * g0 + (J02 * J21) * g1 + (J01 * J12) * g2;
            const double d1 = (J12 * J20) * g0 + (-J02 * J20) * g1 + (J02 * J10) * g2;
            const double d2 = (J10 * J21) * g0 + (J01 * J20) * g1 + (-J01 * J10) * g2;
            s[0] -= d0 * id; s[1] -= d1 * id; s[2] -= d2 * id;
        }
        if (!(res <= 1e-9 * scale) || !(s[0] > 0.0) || !(s[1] > 0.0) || !(s[2] > 0.0)) continue;
        double Y[3][3], EY[3][3];
        for (int i = 0; i < 3; ++i)
            for (int e = 0; e < 3; ++e) Y[i][e] = s[i] * f[i][e];
        if (!triangle_frame(Y[0], Y[1], Y[2], EY)) continue;
        double* o = Rt[k];
        for (int r = 0; r < 3; ++r)
            for (int cc = 0; cc < 3; ++cc)   // R = EY' * EX  (frames stored as rows)
                o[4 * r + cc] = EY[0][r] * EX[0][cc] + EY[1][r] * EX[1][cc] + EY[2][r] * EX[2][cc];
        for (int r = 0; r < 3; ++r)
            o[4 * r + 3] = Y[0][r] - (o[4 * r] * X[0][0] + o[4 * r + 1] * X[0][1] + o[4 * r + 2] * X[0][2]);
        ++ns;
    }
    return ns;
}

}  // namespace p3p
