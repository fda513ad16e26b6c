// p3p.hip — P3P hypothesis generation on the device (SURVEY.md §8f item 1, §8d cfg 5).  No reference code
// (main.py:101-104 calls cv2.solvePnP on all points); the solver is p3p_core.hpp.  One thread per minimal
// sample, float64 in registers: quartic by Grunert's substitution, four roots by Aberth-Ehrlich, every real
// positive root -> depths -> Gauss-Newton polish -> (R, t).  Sample s owns hypothesis slots 4s .. 4s+3.
#include <math.h>

#include "common.hpp"
#include "p3p_core.hpp"

namespace {

__global__ __launch_bounds__(128) void p3p_kernel(const float* __restrict__ X, const float* __restrict__ Y,
                                                  const float* __restrict__ Z, const float* __restrict__ u,
                                                  const float* __restrict__ v, const int32_t* __restrict__ samples, int S,
                                                  int N, const double* __restrict__ K4, float* __restrict__ P,
                                                  int32_t* __restrict__ n_sol) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 128 + threadIdx.x;
    if (s >= S) return;
    const int32_t* si = samples + ((size_t)b * S + s) * 3;
    double Xw[3][3], f[3][3];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int i = si[r];
        ok &= (i >= 0 && i < N);
        const size_t o = (size_t)b * N + (ok ? i : 0);
        Xw[r][0] = X[o]; Xw[r][1] = Y[o]; Xw[r][2] = Z[o];
        const double x = u[o], y = v[o];
        const double inv = ph5::fast_rsqrt(x * x + y * y + 1.0);
        f[r][0] = x * inv; f[r][1] = y * inv; f[r][2] = inv;
    }
    double Rt[4][12];
    int ns = 0;
    if (ok) ns = p3p::solve(Xw, f, Rt);
    float* o = P + ((size_t)b * S + s) * 48;
    if (ns == 0) {
#pragma unroll
        for (int e = 0; e < 48; e += 4) *reinterpret_cast<float4*>(o + e) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        double fx = 1.0, fy = 1.0, cx = 0.0, cy = 0.0;
        if (K4) { fx = K4[0]; fy = K4[1]; cx = K4[2]; cy = K4[3]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float w[12];
#pragma unroll
            for (int c = 0; c < 4; ++c) {        // rows of K [R | t]
                w[c] = (float)(fx * Rt[k][c] + cx * Rt[k][8 + c]);
                w[4 + c] = (float)(fy * Rt[k][4 + c] + cy * Rt[k][8 + c]);
                w[8 + c] = (float)Rt[k][8 + c];
            }
#pragma unroll
            for (int e = 0; e < 12; e += 4) *reinterpret_cast<float4*>(o + 12 * k + e) = make_float4(w[e], w[e + 1], w[e + 2], w[e + 3]);
        }
    }
    if (n_sol) n_sol[(size_t)b * S + s] = ns;
}

}  // namespace

extern "C" int ph_solve_p3p(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                            const int32_t* samples, int B, int S, int N, const double* K4, float* P,
                            int32_t* n_solutions, void* stream) {
    if (B < 0 || S < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || S == 0) return PH_OK;
    if (!X || !Y || !Z || !u || !v || !samples || !P || !ph_aligned(P, 16)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(p3p_kernel, dim3(ph_ceil_div(S, 128), B), dim3(128), 0, (hipStream_t)stream, X, Y, Z, u, v, samples, S,
                       N, K4, P, n_solutions);
    return ph_launch_status();
}
