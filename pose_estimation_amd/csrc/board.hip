// board.hip — the reference's own per-image arithmetic, batched: projection with the
// 5-coefficient distortion model (validate_pose.py:181-184, :275), reprojection-error
// statistics (validate_pose.py:260-288) and the inverse distortion used ahead of scoring.
// Computed in float64 like the reference (rvec/tvec/K/dist are float64 there, main.py:28-34;
// cv2.projectPoints works in double) so that the int32 truncation of validate_pose.py:184
// lands on the same pixel.
#include <math.h>

#include "common.hpp"

namespace {

struct Cam { double fx, fy, cx, cy, k1, k2, p1, p2, k3; };

__device__ __forceinline__ Cam load_cam(const double* K4, const double* d5) {
    return Cam{K4[0], K4[1], K4[2], K4[3], d5[0], d5[1], d5[2], d5[3], d5[4]};
}

// Rodrigues: R = cos(th) I + (1 - cos(th)) k k' + sin(th) [k]x   (cv2.Rodrigues, main.py:122)
__device__ __forceinline__ void rodrigues(const double* r, double (&R)[9]) {
    const double th = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (th < 1e-12) {
        R[0] = 1; R[1] = -r[2]; R[2] = r[1]; R[3] = r[2]; R[4] = 1; R[5] = -r[0]; R[6] = -r[1]; R[7] = r[0]; R[8] = 1;
        return;
    }
    const double kx = r[0] / th, ky = r[1] / th, kz = r[2] / th;
    double s, c;
    sincos(th, &s, &c);
    const double v = 1 - c;
    R[0] = c + kx * kx * v;      R[1] = kx * ky * v - kz * s; R[2] = kx * kz * v + ky * s;
    R[3] = ky * kx * v + kz * s; R[4] = c + ky * ky * v;      R[5] = ky * kz * v - kx * s;
    R[6] = kz * kx * v - ky * s; R[7] = kz * ky * v + kx * s; R[8] = c + kz * kz * v;
}

__device__ __forceinline__ void project_one(const double (&R)[9], const double* t, const Cam& cam, double X, double Y,
                                            double Z, double& u, double& v) {
    const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0];
    const double yc = R[3] * X + R[4] * Y + R[5] * Z + t[1];
    const double zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double x = xc / zc, y = yc / zc, r2 = x * x + y * y;
    const double g = 1 + cam.k1 * r2 + cam.k2 * r2 * r2 + cam.k3 * r2 * r2 * r2;
    const double xd = x * g + 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x);
    const double yd = y * g + cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y;
    u = cam.fx * xd + cam.cx;
    v = cam.fy * yd + cam.cy;
}

__global__ __launch_bounds__(256) void project_kernel(const float* __restrict__ obj, const double* __restrict__ pose,
                                                     const double* __restrict__ K4, const double* __restrict__ d5,
                                                     const int32_t* __restrict__ n_valid, int N,
                                                     double* __restrict__ uv, int32_t* __restrict__ uv_int) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    if (i >= nv) return;
    const Cam cam = load_cam(K4, d5);
    double R[9];
    rodrigues(pose + (size_t)b * 6, R);
    const float* p = obj + ((size_t)b * N + i) * 3;
    double u, v;
    project_one(R, pose + (size_t)b * 6 + 3, cam, p[0], p[1], p[2], u, v);
    const size_t o = ((size_t)b * N + i) * 2;
    if (uv) { uv[o] = u; uv[o + 1] = v; }
    if (uv_int) { uv_int[o] = (int32_t)u; uv_int[o + 1] = (int32_t)v; }   // truncation toward zero
}

// one block per item: per-point error, then mean / std(ddof 0) / max / min / count.
__global__ __launch_bounds__(256) void reproj_kernel(const float* __restrict__ obj, const float* __restrict__ img,
                                                    const double* __restrict__ pose, const double* __restrict__ K4,
                                                    const double* __restrict__ d5, const int32_t* __restrict__ n_valid,
                                                    int N, double* __restrict__ err, double* __restrict__ stats) {
    __shared__ double red[3][256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const Cam cam = load_cam(K4, d5);
    double R[9];
    rodrigues(pose + (size_t)b * 6, R);
    const double* t = pose + (size_t)b * 6 + 3;

    auto error_of = [&](int i) -> double {
        const float* p = obj + ((size_t)b * N + i) * 3;
        const float* m = img + ((size_t)b * N + i) * 2;
        double u, v;
        project_one(R, t, cam, p[0], p[1], p[2], u, v);
        const double du = u - (double)m[0], dv = v - (double)m[1];
        return sqrt(du * du + dv * dv);
    };
    auto block_reduce = [&](double s, double mx, double mn) {
        red[0][tid] = s; red[1][tid] = mx; red[2][tid] = mn;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) {
            if (tid < w) {
                red[0][tid] += red[0][tid + w];
                red[1][tid] = fmax(red[1][tid], red[1][tid + w]);
                red[2][tid] = fmin(red[2][tid], red[2][tid + w]);
            }
            __syncthreads();
        }
    };

    double s = 0, mx = -INFINITY, mn = INFINITY;
    for (int i = tid; i < nv; i += 256) {
        const double e = error_of(i);
        if (err) err[(size_t)b * N + i] = e;
        s += e; mx = fmax(mx, e); mn = fmin(mn, e);
    }
    block_reduce(s, mx, mn);
    const double mean = nv ? red[0][0] / nv : 0.0;
    const double vmax = red[1][0], vmin = red[2][0];
    __syncthreads();
    double var = 0;
    for (int i = tid; i < nv; i += 256) { const double d = error_of(i) - mean; var += d * d; }
    block_reduce(var, 0, 0);
    if (tid == 0) {
        double* o = stats + (size_t)b * 5;
        o[0] = mean; o[1] = nv ? sqrt(red[0][0] / nv) : 0.0; o[2] = vmax; o[3] = vmin; o[4] = (double)nv;
    }
}

// pixels -> normalised undistorted coordinates: fixed-point iteration of the inverse model.
__global__ __launch_bounds__(256) void undistort_kernel(const float* __restrict__ uv, const double* __restrict__ K4,
                                                       const double* __restrict__ d5, int64_t n, int iters,
                                                       float* __restrict__ xy) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Cam cam = load_cam(K4, d5);
    const double x0 = ((double)uv[2 * i] - cam.cx) / cam.fx, y0 = ((double)uv[2 * i + 1] - cam.cy) / cam.fy;
    double x = x0, y = y0;
    for (int it = 0; it < iters; ++it) {
        const double r2 = x * x + y * y;
        const double icd = 1.0 / (1 + ((cam.k3 * r2 + cam.k2) * r2 + cam.k1) * r2);
        const double dx = 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x);
        const double dy = cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y;
        x = (x0 - dx) * icd;
        y = (y0 - dy) * icd;
    }
    xy[2 * i] = (float)x;
    xy[2 * i + 1] = (float)y;
}


// ---------------------------------------------------------------------------------------------
// Levenberg-Marquardt refinement of a pose on the distorted reprojection error — the iterative
// half of cv2.solvePnP(SOLVEPNP_ITERATIVE) (main.py:101-104).  One 256-thread block per item,
// float64.  Parametrisation: left-multiplicative rotation update R <- exp([dw]x) R, t <- t + dt,
// so dXc/ddw = -[R X]x and dXc/ddt = I; the 2x3 Jacobian of (u, v) w.r.t. Xc is analytic.
// Per iteration: pass 1 accumulates J'J (21 unique), J'r (6) and the cost over the item's points
// (wave shuffles, then LDS across the 4 waves); thread 0 solves the damped 6x6 system (Cholesky);
// pass 2 evaluates the cost at the candidate pose; accept -> lambda/10, reject -> lambda*10.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void exp_so3(const double* w, double (&R)[9]) { rodrigues(w, R); }

__device__ __forceinline__ void mat3_mul(const double (&A)[9], const double (&B)[9], double (&C)[9]) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[3 * r + c] = A[3 * r] * B[c] + A[3 * r + 1] * B[3 + c] + A[3 * r + 2] * B[6 + c];
}

// rotation matrix -> rotation vector (angle in [0, pi])
__device__ void log_so3(const double (&R)[9], double* w) {
    const double tr = R[0] + R[4] + R[8];
    const double c = fmin(fmax((tr - 1.0) * 0.5, -1.0), 1.0);
    const double a0 = R[7] - R[5], a1 = R[2] - R[6], a2 = R[3] - R[1];
    const double th = atan2(0.5 * sqrt(a0 * a0 + a1 * a1 + a2 * a2), c);    // not acos(c): its slope is 1/sin(theta) next to 0 and pi
    if (th < 1e-10) { w[0] = 0.5 * a0; w[1] = 0.5 * a1; w[2] = 0.5 * a2; return; }
    if (3.141592653589793 - th < 1e-6) {
        // sym(R) = cos I + (1 - cos) k k'  ->  k k' = (sym(R) - c I) / (1 - c), exact for every angle
        const double f = 1.0 / (1.0 - c);
        const double d0 = sqrt(fmax((R[0] - c) * f, 0.0)), d1 = sqrt(fmax((R[4] - c) * f, 0.0)),
                     d2 = sqrt(fmax((R[8] - c) * f, 0.0));
        const double s01 = (R[1] + R[3]) * 0.5 * f, s02 = (R[2] + R[6]) * 0.5 * f, s12 = (R[5] + R[7]) * 0.5 * f;
        double k[3];
        if (d0 >= d1 && d0 >= d2) { k[0] = d0; k[1] = s01 / d0; k[2] = s02 / d0; }
        else if (d1 >= d2) { k[1] = d1; k[0] = s01 / d1; k[2] = s12 / d1; }
        else { k[2] = d2; k[0] = s02 / d2; k[1] = s12 / d2; }
        const double sgn = (a0 * k[0] + a1 * k[1] + a2 * k[2]) < 0.0 ? -1.0 : 1.0;
        w[0] = sgn * th * k[0]; w[1] = sgn * th * k[1]; w[2] = sgn * th * k[2];
        return;
    }
    const double f = th / (2.0 * sin(th));
    w[0] = f * a0; w[1] = f * a1; w[2] = f * a2;
}

// residual (ru, rv) of one point and, optionally, its 2x6 Jacobian w.r.t. (dw, dt)
__device__ __forceinline__ void lm_point(const double (&R)[9], const double* t, const Cam& cam, const float (&X3)[3], const float (&m2)[2],
                                         double& ru, double& rv, double (*J)[6]) {
    const double X = X3[0], Y = X3[1], Z = X3[2];
    const double rx = R[0] * X + R[1] * Y + R[2] * Z, ry = R[3] * X + R[4] * Y + R[5] * Z, rz = R[6] * X + R[7] * Y + R[8] * Z;
    const double xc = rx + t[0], yc = ry + t[1], zc = rz + t[2];
    const double iz = 1.0 / zc;
    const double x = xc * iz, y = yc * iz, r2 = x * x + y * y;
    const double g = 1 + cam.k1 * r2 + cam.k2 * r2 * r2 + cam.k3 * r2 * r2 * r2;
    const double xd = x * g + 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x);
    const double yd = y * g + cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y;
    ru = cam.fx * xd + cam.cx - (double)m2[0];
    rv = cam.fy * yd + cam.cy - (double)m2[1];
    if (!J) return;
    const double dg = cam.k1 + 2 * cam.k2 * r2 + 3 * cam.k3 * r2 * r2;          // dg/dr2
    // d(xd, yd)/d(x, y)
    const double xdx = g + 2 * x * x * dg + 2 * cam.p1 * y + 6 * cam.p2 * x;
    const double xdy = 2 * x * y * dg + 2 * cam.p1 * x + 2 * cam.p2 * y;
    const double ydx = 2 * x * y * dg + 2 * cam.p1 * x + 2 * cam.p2 * y;
    const double ydy = g + 2 * y * y * dg + 6 * cam.p1 * y + 2 * cam.p2 * x;
    // d(x, y)/dXc = [iz 0 -x iz; 0 iz -y iz]
    double A[2][3];
    A[0][0] = cam.fx * xdx * iz; A[0][1] = cam.fx * xdy * iz; A[0][2] = -cam.fx * (xdx * x + xdy * y) * iz;
    A[1][0] = cam.fy * ydx * iz; A[1][1] = cam.fy * ydy * iz; A[1][2] = -cam.fy * (ydx * x + ydy * y) * iz;
    // dXc/ddw = -[RX]x = [[0, rz, -ry], [-rz, 0, rx], [ry, -rx, 0]];  dXc/ddt = I
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        J[k][0] = -A[k][1] * rz + A[k][2] * ry;
        J[k][1] = A[k][0] * rz - A[k][2] * rx;
        J[k][2] = -A[k][0] * ry + A[k][1] * rx;
        J[k][3] = A[k][0]; J[k][4] = A[k][1]; J[k][5] = A[k][2];
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// One damped step from the sums s = (J'J upper triangle [21], J'r [6], cost): the candidate pose (Rn, tn).  Returns 0,
// 1 when the step is too small to matter (the candidate is still tried), 2 when the damped system is not positive
// definite (no candidate: the caller raises lambda).  Every loop is unrolled on constant indices: the small matrices
// stay in registers (dynamic indexing put them in scratch memory — an L2 round trip per access).
__device__ __forceinline__ int lm_solve(const double (&s)[28], double lambda, const double (&Rcur)[9], const double (&tcur)[3],
                                        double (&Rn)[9], double (&tn)[3]) {
    // damped system (Marquardt scaling), Cholesky
    double A[6][6], g[6], L[6][6], dlt[6];
    {
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c) { A[a][c] = s[q]; A[c][a] = s[q]; ++q; }
    }
#pragma unroll
    for (int a = 0; a < 6; ++a) { g[a] = -s[21 + a]; A[a][a] += lambda * A[a][a] + 1e-300; }
    bool ok = true;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int c = 0; c <= a; ++c) {
            double v = A[a][c];
#pragma unroll
            for (int k = 0; k < c; ++k) v -= L[a][k] * L[c][k];
            if (a == c) { ok = ok && (v > 0.0); L[a][a] = sqrt(fmax(v, 1e-300)); }
            else L[a][c] = v / L[c][c];
        }
    if (!ok) return 2;
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        double v = g[a];
#pragma unroll
        for (int k = 0; k < a; ++k) v -= L[a][k] * dlt[k];
        dlt[a] = v / L[a][a];
    }
#pragma unroll
    for (int a = 5; a >= 0; --a) {
        double v = dlt[a];
#pragma unroll
        for (int k = a + 1; k < 6; ++k) v -= L[k][a] * dlt[k];
        dlt[a] = v / L[a][a];
    }
    double dR[9];
    exp_so3(dlt, dR);
    mat3_mul(dR, Rcur, Rn);
#pragma unroll
    for (int k = 0; k < 3; ++k) tn[k] = tcur[k] + dlt[3 + k];
    double n2 = 0;
#pragma unroll
    for (int a = 0; a < 6; ++a) n2 += dlt[a] * dlt[a];
    return (n2 < 1e-26) ? 1 : 0;
}

// Point sources of the refinement kernel: the board path hands over interleaved rows (obj [B][N][3], img [B][N][2]),
// the RANSAC path SoA rows [B][N] plus the winning hypothesis' inlier mask.
struct PointsAoS {
    const float* obj; const float* img;
    __device__ __forceinline__ bool get(int b, int N, int i, float (&X3)[3], float (&m2)[2]) const {
        const float* o = obj + ((size_t)b * N + i) * 3;
        const float* m = img + ((size_t)b * N + i) * 2;
        X3[0] = o[0]; X3[1] = o[1]; X3[2] = o[2]; m2[0] = m[0]; m2[1] = m[1];
        return true;
    }
};
struct PointsSoA {
    const float *X, *Y, *Z, *u, *v; const uint64_t* mask; int words;
    __device__ __forceinline__ bool get(int b, int N, int i, float (&X3)[3], float (&m2)[2]) const {
        if (mask && !((mask[(size_t)b * words + (i >> 6)] >> (i & 63)) & 1ull)) return false;
        const size_t o = (size_t)b * N + i;
        X3[0] = X[o]; X3[1] = Y[o]; X3[2] = Z[o]; m2[0] = u[o]; m2[1] = v[o];
        return true;
    }
};

template <class Pts, int NT>
__global__ __launch_bounds__(NT) void refine_pose_kernel(const Pts pts, const int32_t* __restrict__ n_valid, int N,
                                                        const double* __restrict__ K4, const double* __restrict__ d5,
                                                        int max_iters, double* __restrict__ pose, double* __restrict__ rms,
                                                        int32_t* __restrict__ iters_out) {
    constexpr int NW = NT / 64;
    __shared__ double red[NW][28], red_cnt[NW];
    __shared__ double sh_R[9], sh_t[3], sh_Rc[9], sh_tc[3];
    __shared__ int sh_flag;                      // 0 continue, 1 converged/stop
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const Cam cam = load_cam(K4, d5);
    double* P = pose + (size_t)b * 6;

    if (tid == 0) {
        double R0[9];
        rodrigues(P, R0);
        for (int k = 0; k < 9; ++k) sh_R[k] = R0[k];
        for (int k = 0; k < 3; ++k) sh_t[k] = P[3 + k];
        sh_flag = (nv < 3) ? 1 : 0;
    }
    __syncthreads();

    // One pass over the points per LM iteration: the normal equations are accumulated AT THE CANDIDATE together with its
    // cost.  An accepted candidate (the rule) brings the sums of the next iteration with it; after a rejected one the old
    // sums are solved again with a larger lambda — no pass at all.  (Two passes per iteration, normal equations at the
    // current pose and cost at the candidate, computed the same numbers 1.4 times as expensively.)
    __shared__ double sh_s[28], sh_n[28];
    __shared__ double sh_cnt;
    auto sums_at = [&](const double* Rs, const double* ts, double* out, bool count) {
        double R[9], t[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = Rs[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = ts[k];
        double acc[28], cnt = 0.0;
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] = 0.0;
        for (int i = tid; i < nv; i += NT) {
            double ru, rv, J[2][6];
            float X3[3], m2[2];
            if (!pts.get(b, N, i, X3, m2)) continue;
            lm_point(R, t, cam, X3, m2, ru, rv, J);
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int c = a; c < 6; ++c) acc[q++] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] += J[0][a] * ru + J[1][a] * rv;
            acc[27] += ru * ru + rv * rv;
            cnt += 1.0;
        }
#pragma unroll
        for (int k = 0; k < 28; ++k) acc[k] = wave_sum(acc[k]);
        if (count) cnt = wave_sum(cnt);
        __syncthreads();                            // everybody is done with red[] and with *out of the previous use
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 28; ++k) red[wave][k] = acc[k];
            if (count) red_cnt[wave] = cnt;
        }
        __syncthreads();
        if (tid < 28) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += red[w][tid];
            out[tid] = v;
        }
        if (count && tid == 32) {
            double v = 0.0;
            for (int w = 0; w < NW; ++w) v += red_cnt[w];
            sh_cnt = v;
        }
        __syncthreads();
    };
    sums_at(sh_R, sh_t, sh_s, true);

    double lambda = 1e-3;
    int it = 0;
    for (; it < max_iters; ++it) {
        const int stop = sh_flag;
        __syncthreads();                            // thread 0 rewrites the flag below
        if (stop) break;
        if (tid == 0) {
            double s[28], Rcur[9], tcur[3], Rn[9], tn[3];
#pragma unroll
            for (int k = 0; k < 28; ++k) s[k] = sh_s[k];
#pragma unroll
            for (int k = 0; k < 9; ++k) Rcur[k] = sh_R[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) tcur[k] = sh_t[k];
            const int f = lm_solve(s, lambda, Rcur, tcur, Rn, tn);
            if (f != 2) {
#pragma unroll
                for (int k = 0; k < 9; ++k) sh_Rc[k] = Rn[k];
#pragma unroll
                for (int k = 0; k < 3; ++k) sh_tc[k] = tn[k];
            }
            sh_flag = f;
        }
        __syncthreads();
        const int flag = sh_flag;
        if (flag != 2) sums_at(sh_Rc, sh_tc, sh_n, false);    // uniform: every thread sees the same flag
        const double cold = sh_s[27], cnew = sh_n[27];
        __syncthreads();
        if (flag != 2 && cnew < cold) {                        // uniform decision: every thread sees the same numbers
            if (tid == 0) {
                for (int k = 0; k < 9; ++k) sh_R[k] = sh_Rc[k];
                for (int k = 0; k < 3; ++k) sh_t[k] = sh_tc[k];
                if ((cold - cnew) <= 1e-14 * fmax(cold, 1e-300)) sh_flag = 1;
            }
            if (tid < 28) sh_s[tid] = sh_n[tid];
            lambda = fmax(lambda * 0.1, 1e-12);
        } else {
            lambda *= 10.0;
            if (tid == 0) sh_flag = (lambda > 1e12 || flag == 1) ? 1 : 0;
        }
        __syncthreads();
    }
    // the sums at the pose that is returned carry its cost (also with max_iters == 0 and after early stops)
    const double cost = sh_s[27], n_used = sh_cnt;
    if (tid == 0) {
        if (it > 0) {                               // no iteration ran: the caller's pose stays bit-for-bit
            double R[9], w[3];
            for (int k = 0; k < 9; ++k) R[k] = sh_R[k];
            log_so3(R, w);
            P[0] = w[0]; P[1] = w[1]; P[2] = w[2]; P[3] = sh_t[0]; P[4] = sh_t[1]; P[5] = sh_t[2];
        }
        if (rms) rms[b] = n_used > 0.0 ? sqrt(cost / n_used) : 0.0;
        if (iters_out) iters_out[b] = it;
    }
}

// ---- the same iteration with G blocks per item, for large point sets (one block per item leaves all but B of the 256 CUs
// idle: 2.05 ms at 32 items of 65 536 points).  No grid-wide barrier: every LM iteration is one launch.  Launch k finishes
// the iteration whose pass launch k-1 made (sums the G partial sums in a fixed order, accepts or rejects the candidate),
// solves for the next candidate and makes its pass; every block of an item does the few microseconds of serial work
// redundantly (same inputs, same instructions, same numbers) and block 0 records the state for the next launch.  State
// and partial sums are double-buffered in the workspace, so a launch never reads what it writes.  Deterministic; the sums
// are added in another order than in the one-block kernel, so the two agree to rounding, not bit for bit.
struct LmState {
    double R[9], t[3], Rc[9], tc[3], s[28], lambda, cnt;
    int it, stop, pflag, pad;        // pflag: result of lm_solve for the iteration in flight (0, 1, 2); -1: the initial sums
};                                   //        are in flight; -2: finished (nothing in flight, nothing more to do)
constexpr int kLmPart = 32;          // doubles per partial record: 28 sums, the point count, padding
constexpr int kLmNT = 512;

// finishes the iteration in flight from the summed partials n[0..28], then starts the next one unless `last`
__device__ __forceinline__ void lm_advance(LmState& st, const double* n, int max_iters, bool last) {
    if (st.pflag == -2) return;
    if (st.pflag == -1) {
#pragma unroll
        for (int k = 0; k < 28; ++k) st.s[k] = n[k];
        st.cnt = n[28];
    } else {
        const double cold = st.s[27], cnew = n[27];
        if (st.pflag != 2 && cnew < cold) {
#pragma unroll
            for (int k = 0; k < 9; ++k) st.R[k] = st.Rc[k];
#pragma unroll
            for (int k = 0; k < 3; ++k) st.t[k] = st.tc[k];
#pragma unroll
            for (int k = 0; k < 28; ++k) st.s[k] = n[k];
            st.stop = (st.pflag == 1 || (cold - cnew) <= 1e-14 * fmax(cold, 1e-300)) ? 1 : 0;
            st.lambda = fmax(st.lambda * 0.1, 1e-12);
        } else {
            st.lambda *= 10.0;
            st.stop = (st.lambda > 1e12 || st.pflag == 1) ? 1 : 0;
        }
        st.it += 1;
    }
    if (last || st.stop || st.it >= max_iters) { st.pflag = -2; return; }
    st.pflag = lm_solve(st.s, st.lambda, st.R, st.t, st.Rc, st.tc);
}

template <class Pts>
__global__ __launch_bounds__(kLmNT) void refine_pose_step_kernel(const Pts pts, const int32_t* __restrict__ n_valid, int N, int G,
                                                                int chunk, const double* __restrict__ K4,
                                                                const double* __restrict__ d5, int max_iters, int k, int B,
                                                                const double* __restrict__ pose, LmState* __restrict__ state,
                                                                double* __restrict__ part) {
    constexpr int NW = kLmNT / 64;
    __shared__ LmState st;
    __shared__ double nsum[kLmPart];
    __shared__ double red[NW][29];
    const int g = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const LmState* prev = state + (size_t)((k + 1) & 1) * B + b;
    LmState* next = state + (size_t)(k & 1) * B + b;
    if (k > 0) {
        if (prev->pflag == -2) {                    // uniform: nothing in flight; keep the record moving
            if (g == 0 && tid == 0) *next = *prev;
            return;
        }
        if (tid < 29) {                             // the G partial records of the pass in flight, in block order
            const double* pp = part + ((size_t)((k + 1) & 1) * B + b) * G * kLmPart + tid;
            double v = 0.0;
            if (prev->pflag != 2)
                for (int q = 0; q < G; ++q) v += pp[(size_t)q * kLmPart];
            nsum[tid] = v;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (k == 0) {
            const double* P = pose + (size_t)b * 6;
            double R0[9];
            rodrigues(P, R0);
#pragma unroll
            for (int q = 0; q < 9; ++q) { st.R[q] = R0[q]; st.Rc[q] = R0[q]; }
#pragma unroll
            for (int q = 0; q < 3; ++q) { st.t[q] = P[3 + q]; st.tc[q] = P[3 + q]; }
#pragma unroll
            for (int q = 0; q < 28; ++q) st.s[q] = 0.0;
            st.lambda = 1e-3; st.cnt = 0.0; st.it = 0; st.stop = (nv < 3) ? 1 : 0; st.pflag = -1; st.pad = 0;
        } else {
            LmState w = *prev;
            lm_advance(w, nsum, max_iters, false);
            st = w;
        }
        if (g == 0) *next = st;
    }
    __syncthreads();
    if (st.pflag == -2 || st.pflag == 2) return;   // uniform: finished, or no candidate to try
    const Cam cam = load_cam(K4, d5);
    double R[9], t[3];
#pragma unroll
    for (int q = 0; q < 9; ++q) R[q] = st.Rc[q];
#pragma unroll
    for (int q = 0; q < 3; ++q) t[q] = st.tc[q];
    double acc[29];
#pragma unroll
    for (int q = 0; q < 29; ++q) acc[q] = 0.0;
    const int i1 = min(nv, (g + 1) * chunk);
    for (int i = g * chunk + tid; i < i1; i += kLmNT) {
        double ru, rv, J[2][6];
        float X3[3], m2[2];
        if (!pts.get(b, N, i, X3, m2)) continue;
        lm_point(R, t, cam, X3, m2, ru, rv, J);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int c = a; c < 6; ++c) acc[q++] += J[0][a] * J[0][c] + J[1][a] * J[1][c];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += J[0][a] * ru + J[1][a] * rv;
        acc[27] += ru * ru + rv * rv;
        acc[28] += 1.0;
    }
#pragma unroll
    for (int q = 0; q < 29; ++q) acc[q] = wave_sum(acc[q]);
    if (lane == 0)
#pragma unroll
        for (int q = 0; q < 29; ++q) red[wave][q] = acc[q];
    __syncthreads();
    if (tid < 29) {
        double v = 0.0;
        for (int w = 0; w < NW; ++w) v += red[w][tid];
        part[(((size_t)(k & 1) * B + b) * G + g) * kLmPart + tid] = v;
    }
}

// after the last step launch (index k_last): finishes the iteration in flight and writes the results
__global__ __launch_bounds__(64) void refine_pose_finish_kernel(int G, int max_iters, int k_last, int B,
                                                               const LmState* __restrict__ state, const double* __restrict__ part,
                                                               double* __restrict__ pose, double* __restrict__ rms,
                                                               int32_t* __restrict__ iters_out) {
    __shared__ double nsum[kLmPart];
    const int b = blockIdx.x, tid = threadIdx.x;
    const LmState* prev = state + (size_t)(k_last & 1) * B + b;
    if (tid < 29) {
        const double* pp = part + ((size_t)(k_last & 1) * B + b) * G * kLmPart + tid;
        double v = 0.0;
        if (prev->pflag != 2 && prev->pflag != -2)
            for (int q = 0; q < G; ++q) v += pp[(size_t)q * kLmPart];
        nsum[tid] = v;
    }
    __syncthreads();
    if (tid != 0) return;
    LmState w = *prev;
    lm_advance(w, nsum, max_iters, true);
    if (w.it > 0) {                                 // no iteration ran: the caller's pose stays bit-for-bit
        double* P = pose + (size_t)b * 6;
        double rv[3];
        log_so3(w.R, rv);
        P[0] = rv[0]; P[1] = rv[1]; P[2] = rv[2]; P[3] = w.t[0]; P[4] = w.t[1]; P[5] = w.t[2];
    }
    if (rms) rms[b] = w.cnt > 0.0 ? sqrt(w.s[27] / w.cnt) : 0.0;
    if (iters_out) iters_out[b] = w.it;
}

struct LmPlan { int G, chunk; size_t off_part, total; };
LmPlan plan_lm(int B, int N) {
    LmPlan p;
    int G = 256 / (B > 0 ? B : 1);                  // one block per CU
    if (G > 16) G = 16;
    if (G > N / 4096) G = N / 4096;                 // at least 4096 points per block
    if (G < 1) G = 1;
    p.G = G;
    p.chunk = (ph_ceil_div(N > 0 ? N : 1, G) + 63) / 64 * 64;
    p.off_part = (sizeof(LmState) * 2 * (size_t)(B > 0 ? B : 0) + 255) / 256 * 256;
    p.total = G > 1 ? p.off_part + sizeof(double) * 2 * (size_t)B * G * kLmPart : 0;
    return p;
}

// Winning 3x4 hypothesis -> (rvec, tvec): undo K when given, project the 3x3 block onto SO(3) (one Newton step of the
// polar decomposition is enough for a matrix that is a rotation up to f32 rounding), Rodrigues vector.
__global__ __launch_bounds__(64) void pose_from_projection_kernel(const float* __restrict__ P, const int32_t* __restrict__ sel,
                                                                 int H, int B, const double* __restrict__ K4,
                                                                 double* __restrict__ pose) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int h = sel ? min(max(sel[b], 0), H - 1) : 0;
    const float* p = P + ((size_t)b * H + h) * 12;
    double M[3][4];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) M[r][c] = (double)p[4 * r + c];
    if (K4) {
        const double fx = K4[0], fy = K4[1], cx = K4[2], cy = K4[3];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            M[0][c] = (M[0][c] - cx * M[2][c]) / fx;
            M[1][c] = (M[1][c] - cy * M[2][c]) / fy;
        }
    }
    double R[9] = {M[0][0], M[0][1], M[0][2], M[1][0], M[1][1], M[1][2], M[2][0], M[2][1], M[2][2]};
    // R <- R (3 I - R'R) / 2   (Newton-Schulz step towards the nearest orthogonal matrix)
    double G[9], Rn[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            G[3 * i + j] = (i == j ? 3.0 : 0.0) - (R[i] * R[j] + R[3 + i] * R[3 + j] + R[6 + i] * R[6 + j]);
    mat3_mul(R, G, Rn);
#pragma unroll
    for (int k = 0; k < 9; ++k) Rn[k] *= 0.5;
    double w[3];
    log_so3(Rn, w);
    double* o = pose + (size_t)b * 6;
    o[0] = w[0]; o[1] = w[1]; o[2] = w[2]; o[3] = M[0][3]; o[4] = M[1][3]; o[5] = M[2][3];
}

// get_camera_position_and_orientation (main.py:112-143), batched: per pose 22 doubles =
// position (3) | forward (3) | up (3) | R_cam_to_world row-major (9) | quaternion w,x,y,z (4).
// ---------------------------------------------------------------------------------------------
// Initial pose of a coplanar (z = 0) point set from the plane-to-image homography — the first half of
// cv2.solvePnP(SOLVEPNP_ITERATIVE) on board points (main.py:101-104), batched: one 256-thread block per image.
// Normalised DLT: image points undistorted (20 fixed-point steps), both point sets centred and scaled to mean distance
// sqrt(2); the 2n x 9 system enters as its 9 x 9 normal matrix A'A (45 block-reduced sums; the normalisation keeps its
// condition number near 1e2..1e4, and the Levenberg-Marquardt stage that follows removes what the squaring costs), whose
// eigenvector of the smallest eigenvalue (cyclic Jacobi, one thread, matrices in LDS) is the homography; then
// H = Ti^-1 Hn To, scale 2 / (|h1| + |h2|), sign from the board lying in front of the camera, and the nearest rotation to
// [r1 r2 r1 x r2] by Newton's polar iteration X <- (X + X^-T) / 2 (the U V' of the SVD the host twin takes).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void initial_pose_kernel(const float* __restrict__ obj, const float* __restrict__ img,
                                                          const int32_t* __restrict__ n_valid, int N,
                                                          const double* __restrict__ K4, const double* __restrict__ d5,
                                                          double* __restrict__ pose, int32_t* __restrict__ ok_out) {
    __shared__ double red[4][45];
    __shared__ double M[81], V[81];
    __shared__ double sh[8];                               // mo.x mo.y mi.x mi.y so si
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const Cam cam = load_cam(K4, d5);
    double* P = pose + (size_t)b * 6;
    if (nv < 4) {                                          // uniform
        if (tid == 0) { for (int k = 0; k < 6; ++k) P[k] = NAN; if (ok_out) ok_out[b] = 0; }
        return;
    }
    auto point = [&](int i, double& X, double& Y, double& x, double& y) {
        const float* o = obj + ((size_t)b * N + i) * 3;
        const float* m = img + ((size_t)b * N + i) * 2;
        X = o[0]; Y = o[1];
        const double x0 = ((double)m[0] - cam.cx) / cam.fx, y0 = ((double)m[1] - cam.cy) / cam.fy;
        x = x0; y = y0;
        for (int it = 0; it < 20; ++it) {
            const double r2 = x * x + y * y;
            const double icd = 1.0 / (1 + ((cam.k3 * r2 + cam.k2) * r2 + cam.k1) * r2);
            const double dx = 2 * cam.p1 * x * y + cam.p2 * (r2 + 2 * x * x);
            const double dy = cam.p1 * (r2 + 2 * y * y) + 2 * cam.p2 * x * y;
            x = (x0 - dx) * icd; y = (y0 - dy) * icd;
        }
    };
    auto block_sum = [&](double (&a)[45], int n) {         // sums of a[0..n) over the block -> red[0][0..n)
#pragma unroll
        for (int k = 0; k < 45; ++k) if (k < n) a[k] = wave_sum(a[k]);
        __syncthreads();                                   // red is free again
        if (lane == 0)
#pragma unroll
            for (int k = 0; k < 45; ++k) if (k < n) red[wave][k] = a[k];
        __syncthreads();
        if (tid < n) red[0][tid] = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
    };
    double a[45];
    // pass 1: centroids
#pragma unroll
    for (int k = 0; k < 45; ++k) a[k] = 0.0;
    for (int i = tid; i < nv; i += 256) { double X, Y, x, y; point(i, X, Y, x, y); a[0] += X; a[1] += Y; a[2] += x; a[3] += y; }
    block_sum(a, 4);
    if (tid < 4) sh[tid] = red[0][tid] / nv;
    __syncthreads();
    const double mox = sh[0], moy = sh[1], mix = sh[2], miy = sh[3];
    // pass 2: mean distances to the centroids
#pragma unroll
    for (int k = 0; k < 45; ++k) a[k] = 0.0;
    for (int i = tid; i < nv; i += 256) {
        double X, Y, x, y; point(i, X, Y, x, y);
        a[0] += sqrt((X - mox) * (X - mox) + (Y - moy) * (Y - moy));
        a[1] += sqrt((x - mix) * (x - mix) + (y - miy) * (y - miy));
    }
    block_sum(a, 2);
    if (tid < 2) sh[4 + tid] = 1.4142135623730951 / fmax(red[0][tid] / nv, 1e-12);
    __syncthreads();
    const double so = sh[4], si = sh[5];
    // pass 3: A'A of the rows [Pn 1 0 0 0 -x Pn -x] and [0 0 0 Pn 1 -y Pn -y]
#pragma unroll
    for (int k = 0; k < 45; ++k) a[k] = 0.0;
    for (int i = tid; i < nv; i += 256) {
        double X, Y, x, y; point(i, X, Y, x, y);
        const double px = (X - mox) * so, py = (Y - moy) * so, xn = (x - mix) * si, yn = (y - miy) * si;
        const double ra[9] = {px, py, 1.0, 0.0, 0.0, 0.0, -xn * px, -xn * py, -xn};
        const double rb[9] = {0.0, 0.0, 0.0, px, py, 1.0, -yn * px, -yn * py, -yn};
        int q = 0;
#pragma unroll
        for (int r = 0; r < 9; ++r)
#pragma unroll
            for (int c = r; c < 9; ++c) a[q++] += ra[r] * ra[c] + rb[r] * rb[c];
    }
    block_sum(a, 45);
    if (tid != 0) return;
    {
        int q = 0;
        for (int r = 0; r < 9; ++r)
            for (int c = r; c < 9; ++c) { M[9 * r + c] = red[0][q]; M[9 * c + r] = red[0][q]; ++q; }
        for (int k = 0; k < 81; ++k) V[k] = (k % 10 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 12; ++sweep) {             // cyclic Jacobi on the symmetric 9 x 9
        double off = 0.0;
        for (int r = 0; r < 9; ++r) for (int c = r + 1; c < 9; ++c) off += M[9 * r + c] * M[9 * r + c];
        if (off < 1e-40) break;
        for (int p = 0; p < 8; ++p)
            for (int q = p + 1; q < 9; ++q) {
                const double apq = M[9 * p + q];
                if (apq == 0.0) continue;
                const double theta = (M[9 * q + q] - M[9 * p + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 9; ++k) {              // columns p, q
                    const double mkp = M[9 * k + p], mkq = M[9 * k + q];
                    M[9 * k + p] = c * mkp - s * mkq; M[9 * k + q] = s * mkp + c * mkq;
                }
                for (int k = 0; k < 9; ++k) {              // rows p, q
                    const double mpk = M[9 * p + k], mqk = M[9 * q + k];
                    M[9 * p + k] = c * mpk - s * mqk; M[9 * q + k] = s * mpk + c * mqk;
                }
                for (int k = 0; k < 9; ++k) {
                    const double vkp = V[9 * k + p], vkq = V[9 * k + q];
                    V[9 * k + p] = c * vkp - s * vkq; V[9 * k + q] = s * vkp + c * vkq;
                }
            }
    }
    int kmin = 0;
    for (int k = 1; k < 9; ++k) if (M[10 * k] < M[10 * kmin]) kmin = k;
    double Hn[9];
    for (int k = 0; k < 9; ++k) Hn[k] = V[9 * k + kmin];
    // H = Ti^-1 Hn To,  Ti^-1 = [1/si 0 mi.x; 0 1/si mi.y; 0 0 1],  To = [so 0 -so mo.x; 0 so -so mo.y; 0 0 1]
    double G[9], H[9];
    const double isi = 1.0 / si;
    for (int c = 0; c < 3; ++c) {
        G[c] = isi * Hn[c] + mix * Hn[6 + c];
        G[3 + c] = isi * Hn[3 + c] + miy * Hn[6 + c];
        G[6 + c] = Hn[6 + c];
    }
    for (int r = 0; r < 3; ++r) {
        H[3 * r] = G[3 * r] * so; H[3 * r + 1] = G[3 * r + 1] * so;
        H[3 * r + 2] = -so * mox * G[3 * r] - so * moy * G[3 * r + 1] + G[3 * r + 2];
    }
    const double n1 = sqrt(H[0] * H[0] + H[3] * H[3] + H[6] * H[6]), n2 = sqrt(H[1] * H[1] + H[4] * H[4] + H[7] * H[7]);
    double lam = 2.0 / (n1 + n2);
    if (H[8] * lam < 0.0) lam = -lam;                      // board in front of the camera
    const double r1[3] = {lam * H[0], lam * H[3], lam * H[6]}, r2[3] = {lam * H[1], lam * H[4], lam * H[7]};
    const double t[3] = {lam * H[2], lam * H[5], lam * H[8]};
    const double r3[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    double X[9] = {r1[0], r2[0], r3[0], r1[1], r2[1], r3[1], r1[2], r2[2], r3[2]};
    for (int it = 0; it < 30; ++it) {                      // polar factor: X <- (X + X^-T) / 2
        const double c00 = X[4] * X[8] - X[5] * X[7], c01 = X[5] * X[6] - X[3] * X[8], c02 = X[3] * X[7] - X[4] * X[6];
        const double c10 = X[2] * X[7] - X[1] * X[8], c11 = X[0] * X[8] - X[2] * X[6], c12 = X[1] * X[6] - X[0] * X[7];
        const double c20 = X[1] * X[5] - X[2] * X[4], c21 = X[2] * X[3] - X[0] * X[5], c22 = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * c00 + X[1] * c01 + X[2] * c02;
        const double id = 1.0 / det;
        const double Y[9] = {c00 * id, c01 * id, c02 * id, c10 * id, c11 * id, c12 * id, c20 * id, c21 * id, c22 * id};   // X^-T
        double change = 0.0;
        for (int k = 0; k < 9; ++k) { const double nx = 0.5 * (X[k] + Y[k]); change += fabs(nx - X[k]); X[k] = nx; }
        if (change < 1e-15) break;
    }
    const double R[9] = {X[0], X[1], X[2], X[3], X[4], X[5], X[6], X[7], X[8]};
    double w[3];
    log_so3(R, w);
    bool fin = true;
    for (int k = 0; k < 3; ++k) { P[k] = w[k]; P[3 + k] = t[k]; fin = fin && (w[k] - w[k] == 0.0) && (t[k] - t[k] == 0.0); }
    if (ok_out) ok_out[b] = fin ? 1 : 0;
}

// Quaternion: the algorithm of scipy Rotation.from_matrix(...).as_quat() for a proper rotation
// (largest of the three diagonal entries and the trace picks the branch; no sign canonicalisation),
// re-ordered to [w, x, y, z] as main.py:138-141 does.
__global__ __launch_bounds__(64) void camera_from_pose_kernel(const double* __restrict__ pose, int B, double* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const double* p = pose + (size_t)b * 6;
    double R[9];
    rodrigues(p, R);                                   // board -> camera
    double* o = out + (size_t)b * 22;
    const double t0 = p[3], t1 = p[4], t2 = p[5];
    double C[9];                                       // camera -> world = R'
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = R[3 * j + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        o[i] = -(C[3 * i] * t0 + C[3 * i + 1] * t1 + C[3 * i + 2] * t2);    // -R' t        (main.py:126)
        o[3 + i] = C[3 * i + 2];                                           // R' [0 0 1]'  (main.py:132)
        o[6 + i] = -C[3 * i + 1];                                          // R' [0 -1 0]' (main.py:135)
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) o[9 + k] = C[k];
    const double d0 = C[0], d1 = C[4], d2 = C[8], tr = d0 + d1 + d2;
    double q[4];                                       // x, y, z, w
    if (tr > d0 && tr > d1 && tr > d2) {              // numpy argmax over (d0, d1, d2, tr): first maximum wins
        q[0] = C[7] - C[5]; q[1] = C[2] - C[6]; q[2] = C[3] - C[1]; q[3] = 1.0 + tr;
    } else {
        const int i = (d0 >= d1 && d0 >= d2) ? 0 : (d1 >= d2 ? 1 : 2);
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        q[i] = 1.0 - tr + 2.0 * C[4 * i];
        q[j] = C[3 * j + i] + C[3 * i + j];
        q[k] = C[3 * k + i] + C[3 * i + k];
        q[3] = C[3 * k + j] - C[3 * j + k];
    }
    const double inv = 1.0 / sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    o[18] = q[3] * inv; o[19] = q[0] * inv; o[20] = q[1] * inv; o[21] = q[2] * inv;
}

}  // namespace

extern "C" int ph_project_points(const float* obj, const double* pose, const double* K4, const double* dist5,
                                 const int32_t* n_valid, int B, int N, double* uv, int32_t* uv_int, void* stream) {
    if (B < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return PH_OK;
    if (!obj || !pose || !K4 || !dist5 || (!uv && !uv_int)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(project_kernel, dim3(ph_ceil_div(N, 256), B), dim3(256), 0, (hipStream_t)stream, obj, pose, K4,
                       dist5, n_valid, N, uv, uv_int);
    return ph_launch_status();
}

extern "C" int ph_reprojection_error(const float* obj, const float* img, const double* pose, const double* K4,
                                     const double* dist5, const int32_t* n_valid, int B, int N, double* err,
                                     double* stats, void* stream) {
    if (B < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!pose || !K4 || !dist5 || !stats || (N > 0 && (!obj || !img))) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(reproj_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, obj, img, pose, K4, dist5, n_valid, N,
                       err, stats);
    return ph_launch_status();
}

extern "C" int ph_undistort_points(const float* uv, const double* K4, const double* dist5, int64_t n, int iters,
                                   float* xy_norm, void* stream) {
    if (n < 0 || iters < 0) return PH_ERR_INVALID_ARG;
    if (n == 0) return PH_OK;
    if (!uv || !K4 || !dist5 || !xy_norm) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(undistort_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, uv, K4,
                       dist5, n, iters, xy_norm);
    return ph_launch_status();
}

extern "C" int ph_refine_pose_lm(const float* obj, const float* img, const int32_t* n_valid, int B, int N,
                                 const double* K4, const double* dist5, int max_iters, double* pose, double* rms,
                                 int32_t* iters, void* stream) {
    if (B < 0 || N < 0 || max_iters < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!pose || !K4 || !dist5 || (N > 0 && (!obj || !img))) return PH_ERR_INVALID_ARG;
    const PointsAoS pts{obj, img};
    if (N <= 2048)
        hipLaunchKernelGGL((refine_pose_kernel<PointsAoS, 256>), dim3(B), dim3(256), 0, (hipStream_t)stream, pts, n_valid, N,
                           K4, dist5, max_iters, pose, rms, iters);
    else
        hipLaunchKernelGGL((refine_pose_kernel<PointsAoS, 512>), dim3(B), dim3(512), 0, (hipStream_t)stream, pts, n_valid,
                           N, K4, dist5, max_iters, pose, rms, iters);
    return ph_launch_status();
}

extern "C" int ph_refine_pose_lm_soa(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                                     const uint64_t* mask, const int32_t* n_valid, int B, int N, const double* K4,
                                     const double* dist5, int max_iters, double* pose, double* rms, int32_t* iters,
                                     void* stream) {
    if (B < 0 || N < 0 || max_iters < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!pose || !K4 || !dist5 || (N > 0 && (!X || !Y || !Z || !u || !v))) return PH_ERR_INVALID_ARG;
    const PointsSoA pts{X, Y, Z, u, v, mask, (N + 63) / 64};
    if (N <= 2048)
        hipLaunchKernelGGL((refine_pose_kernel<PointsSoA, 256>), dim3(B), dim3(256), 0, (hipStream_t)stream, pts, n_valid, N,
                           K4, dist5, max_iters, pose, rms, iters);
    else
        hipLaunchKernelGGL((refine_pose_kernel<PointsSoA, 512>), dim3(B), dim3(512), 0, (hipStream_t)stream, pts, n_valid,
                           N, K4, dist5, max_iters, pose, rms, iters);
    return ph_launch_status();
}

extern "C" size_t ph_refine_pose_lm_soa_ws_bytes(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return plan_lm(B, N).total;
}

extern "C" int ph_refine_pose_lm_soa_ws(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                                        const uint64_t* mask, const int32_t* n_valid, int B, int N, const double* K4,
                                        const double* dist5, int max_iters, double* pose, double* rms, int32_t* iters,
                                        void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || N < 0 || max_iters < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    const LmPlan p = plan_lm(B, N);
    if (p.G <= 1 || B > 65535)                      // small point sets: the one-block kernel, no workspace
        return ph_refine_pose_lm_soa(X, Y, Z, u, v, mask, n_valid, B, N, K4, dist5, max_iters, pose, rms, iters, stream);
    if (!pose || !K4 || !dist5 || !X || !Y || !Z || !u || !v) return PH_ERR_INVALID_ARG;
    if (!ws || ws_bytes < p.total || !ph_aligned(ws, 16)) return PH_ERR_WORKSPACE;
    const PointsSoA pts{X, Y, Z, u, v, mask, (N + 63) / 64};
    LmState* state = (LmState*)ws;
    double* part = (double*)((char*)ws + p.off_part);
    for (int k = 0; k <= max_iters; ++k)
        hipLaunchKernelGGL((refine_pose_step_kernel<PointsSoA>), dim3(p.G, B), dim3(kLmNT), 0, (hipStream_t)stream, pts, n_valid, N,
                           p.G, p.chunk, K4, dist5, max_iters, k, B, pose, state, part);
    hipLaunchKernelGGL(refine_pose_finish_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, p.G, max_iters, max_iters, B, state,
                       part, pose, rms, iters);
    return ph_launch_status();
}

extern "C" int ph_pose_from_projection(const float* P, const int32_t* sel, int H, int B, const double* K4, double* pose,
                                       void* stream) {
    if (B < 0 || H < 1) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!P || !pose) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pose_from_projection_kernel, dim3(ph_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, P, sel, H, B,
                       K4, pose);
    return ph_launch_status();
}

extern "C" int ph_initial_board_pose(const float* obj, const float* img, const int32_t* n_valid, int B, int N,
                                     const double* K4, const double* dist5, double* pose, int32_t* ok, void* stream) {
    if (B < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0) return PH_OK;
    if (!pose || !K4 || !dist5 || (N > 0 && (!obj || !img))) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(initial_pose_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, obj, img, n_valid, N, K4, dist5, pose, ok);
    return ph_launch_status();
}

extern "C" int ph_camera_from_pose(const double* pose, int B, double* camera, void* stream) {
    if (B < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!pose || !camera) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(camera_from_pose_kernel, dim3(ph_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, pose, B, camera);
    return ph_launch_status();
}
