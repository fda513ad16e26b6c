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
