// fivepoint.hip — RANSAC hypothesis generation on the device: counter-based minimal sampling
// (Philox4x32-10) and Nistér's 5-point essential-matrix solver.  SURVEY.md §8f item 1: there is
// no reference code (main.py only calls cv2.solvePnP); the algebra is the published one
// (D. Nistér, "An efficient solution to the five-point relative pose problem", PAMI 2004).
//
// One thread per minimal sample, float64 in private memory: null space of the 5x9 epipolar system
// (Gauss-Jordan, full pivoting, re-orthonormalised) -> the 10 cubic constraints det(E) = 0 and
// 2EE'E - tr(EE')E = 0 as a 10x20 coefficient matrix -> Gauss-Jordan -> 3x3 polynomial matrix in z
// -> degree-10 determinant -> all roots by Aberth-Ehrlich iteration -> for every real root the E it
// defines, unit Frobenius norm, stored as f32 in one of the sample's 10 hypothesis slots (unused
// slots are all-zero matrices, which score zero inliers).
#include <math.h>

#include <mutex>

#include "common.hpp"
#include "fivepoint_core.hpp"

namespace {

// ---------------------------------------------------------------- Philox4x32-10 (Salmon et al. 2011)
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0, hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
__device__ __forceinline__ uint32_t philox_word0(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c[0];
}

// sample s of item b: m distinct indices in [0, n_valid[b]); draw t = Philox(counter (s,t,0,0), key (seed,b))
// word 0 mapped by multiply-shift; a repeated index is skipped.  n_valid < m -> -1.
__global__ __launch_bounds__(256) void sample_kernel(uint32_t seed, const uint32_t* __restrict__ item_key,
                                                    const int32_t* __restrict__ n_valid, int N, int S, int m,
                                                    int32_t* __restrict__ out) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= S) return;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    int32_t* o = out + ((size_t)b * S + s) * m;
    if (nv < m) { for (int k = 0; k < m; ++k) o[k] = -1; return; }
    const uint32_t key = item_key ? item_key[b] : (uint32_t)b;
    int got[8];
    int n = 0;
    for (uint32_t t = 0; n < m; ++t) {
        const uint32_t w = philox_word0((uint32_t)s, t, seed, key);
        const int idx = (int)(((uint64_t)w * (uint64_t)nv) >> 32);
        bool dup = false;
        for (int k = 0; k < n; ++k) dup |= (got[k] == idx);
        if (!dup) got[n++] = idx;
    }
    for (int k = 0; k < m; ++k) o[k] = got[k];
}

// Stage 1: one thread per minimal sample builds the degree-10 polynomial (sections 1-8 of the core).
// Blocks are narrow on purpose: the work is a long dependent chain per sample over ~3 KB of dynamically
// indexed working arrays, kept in LDS (one Work5 per thread), and with only B*S samples the launch is
// latency-bound — narrow blocks spread the samples over all CUs.  32 threads per block while that gives at most
// one block per CU; beyond, up to 52 (52 x 2984 B = 155 KB, the most one CU's LDS holds), so that e.g.
// 32 pairs x 410 samples still run as ONE round of 256 blocks instead of two.
constexpr int kPolyThreads = 32;
constexpr int kPolyThreadsMax = 52;
__global__ __launch_bounds__(64) void fivepoint_poly_kernel(const float* __restrict__ x1, const float* __restrict__ y1,
                                                                     const float* __restrict__ x2, const float* __restrict__ y2,
                                                                     const int32_t* __restrict__ samples, int S, int N,
                                                                     ph5::Poly5* __restrict__ polys) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ph5::Work5* work = reinterpret_cast<ph5::Work5*>(smem);
    const int b = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    ph5::Poly5* out = polys + (size_t)b * S + s;
    const int32_t* si = samples + ((size_t)b * S + s) * 5;
    double pts[5][4];
    bool ok = true;
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        const int i = si[r];
        ok &= (i >= 0 && i < N);
        const size_t o = (size_t)b * N + (ok ? i : 0);
        pts[r][0] = x1[o]; pts[r][1] = y1[o]; pts[r][2] = x2[o]; pts[r][3] = y2[o];
    }
    ph5::Poly5 P;
    if (!ok || !ph5::build_polynomial(pts, P, work[threadIdx.x])) P.n = 0;
    *out = P;
}

// Stage 2: one lane per (sample, root): 6 samples x 10 roots per wave (lanes 60-63 idle).  The ten lanes
// of a sample run the Aberth-Ehrlich iteration together (simultaneous update, roots exchanged by
// shuffles), then every lane turns its own root into an essential matrix and writes slot k.
// Three waves per SIMD (168 VGPR, 7 spilled, against 192): the 2 187 one-wave blocks of 32 pairs x 410 samples then run as
// ONE round (256 CUs x 12 waves) instead of a full round and a tail, and a wave fits beside two waves of the L2 filter.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void fivepoint_roots_kernel(const ph5::Poly5* __restrict__ polys, long n_samples,
                                                            float* __restrict__ E_out, int32_t* __restrict__ n_sol) {
    const int lane = threadIdx.x;
    const int g = lane / 10, k = lane - 10 * g;
    const long s = (long)blockIdx.x * 6 + g;
    const bool active = (g < 6) && (s < n_samples);
    const ph5::Poly5* Pp = polys + (active ? s : 0);
    ph5::Poly5 P;
    for (int q = 0; q < 11; ++q) P.c[q] = Pp->c[q];
    P.n = Pp->n;
    const int n = P.n;
    const bool live = active && k < n;
    ph5::cplx z = live ? ph5::aberth_start(P, k) : ph5::cplx{0.0, 0.0};
    for (int it = 0; it < ph5::kAberthMaxIter; ++it) {
        ph5::cplx sum = {0.0, 0.0};
        for (int d = 1; d < 10; ++d) {
            int kk = k + d; kk = kk >= 10 ? kk - 10 : kk;
            const int src = min(10 * g + kk, 63);
            const double zr = __shfl(z.re, src), zi = __shfl(z.im, src);
            if (kk < n) {
                const ph5::cplx dd = {z.re - zr, z.im - zi};
                const ph5::cplx inv = ph5::cinv(dd);
                sum.re += inv.re; sum.im += inv.im;
            }
        }
        double change = 0.0;
        if (live) {
            const ph5::cplx w = ph5::aberth_step(P, z, sum);
            z.re -= w.re; z.im -= w.im;
            change = (fabs(w.re) + fabs(w.im)) / (fabs(z.re) + fabs(z.im) + 1e-30);
        }
        if (__all(change < ph5::aberth_tol(it))) break;            // wave-uniform exit
    }
    float e9[9];
    bool real = false;
    if (live) {
        for (int q = 0; q < 3; ++q) {
            for (int r = 0; r < 4; ++r) { P.Bx[q][r] = Pp->Bx[q][r]; P.By[q][r] = Pp->By[q][r]; }
            for (int r = 0; r < 5; ++r) P.B1[q][r] = Pp->B1[q][r];
        }
        for (int q = 0; q < 4; ++q)
            for (int r = 0; r < 9; ++r) P.basis[q][r] = Pp->basis[q][r];
        real = ph5::root_to_E(P, z, e9);
    }
    if (active) {
        float* eo = E_out + (size_t)s * 90 + 9 * k;
        for (int e = 0; e < 9; ++e) eo[e] = real ? e9[e] : 0.0f;
    }
    if (n_sol) {
        const unsigned long long votes = __ballot(real);
        if (active && k == 0) n_sol[s] = (int32_t)__popcll((votes >> (10 * g)) & 0x3ffull);
    }
}

}  // namespace

extern "C" int ph_sample_minimal_keyed(uint32_t seed, const uint32_t* item_key, const int32_t* n_valid, int B, int N, int S,
                                       int m, int32_t* samples, void* stream) {
    if (B < 0 || N < 0 || S < 0 || m < 1 || m > 8) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || S == 0) return PH_OK;
    if (!samples) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(sample_kernel, dim3(ph_ceil_div(S, 256), B), dim3(256), 0, (hipStream_t)stream, seed, item_key,
                       n_valid, N, S, m, samples);
    return ph_launch_status();
}

extern "C" int ph_sample_minimal(uint32_t seed, const int32_t* n_valid, int B, int N, int S, int m, int32_t* samples,
                                 void* stream) {
    return ph_sample_minimal_keyed(seed, nullptr, n_valid, B, N, S, m, samples, stream);
}

extern "C" size_t ph_solve_essential_5pt_ws_bytes(int B, int S) {
    if (B <= 0 || S <= 0) return 0;
    return sizeof(ph5::Poly5) * (size_t)B * S;
}

namespace {

int solve_5pt_launch(const float* x1, const float* y1, const float* x2, const float* y2, const int32_t* samples, int B, int S,
                     int N, float* E, int32_t* n_solutions, ph5::Poly5* polys, hipStream_t st) {
    int threads = kPolyThreads;
    while (threads < kPolyThreadsMax && (long)B * ph_ceil_div(S, threads) > 256) threads += 4;
    const size_t lds = sizeof(ph5::Work5) * threads;           // up to 155 KB: above the 64 KiB default, opt in
    if (hipFuncSetAttribute((const void*)fivepoint_poly_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(sizeof(ph5::Work5) * kPolyThreadsMax)) != hipSuccess)
        return PH_ERR_LAUNCH;
    hipLaunchKernelGGL(fivepoint_poly_kernel, dim3(ph_ceil_div(S, threads), B), dim3(threads), lds, st, x1, y1, x2, y2,
                       samples, S, N, polys);
    const long n_samples = (long)B * S;
    hipLaunchKernelGGL(fivepoint_roots_kernel, dim3((unsigned)((n_samples + 5) / 6)), dim3(64), 0, st, polys, n_samples, E,
                       n_solutions);
    return ph_launch_status();
}

// ---- the no-allocation variant's scratch: the polynomial records of PH_NOALLOC_MAX_SAMPLES minimal samples, in the
// library's own device memory (every GPU the library runs on holds its copy of the array, at its own address, until
// the process ends).  In host code the array's NAME is only a stand-in for the symbol: the device address is asked
// for once per device with hipGetSymbolAddress and only that pointer is ever handed to a kernel.
static __device__ __align__(256) unsigned char scratch[(size_t)PH_NOALLOC_MAX_SAMPLES * sizeof(ph5::Poly5)];
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];            // per device, filled on the entry point's first call on that device
std::mutex g_scratch_mu;

int noalloc_scratch(ph5::Poly5** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return PH_ERR_LAUNCH;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return PH_ERR_LAUNCH;
        g_scratch_ptr[dev] = p;
    }
    *out = (ph5::Poly5*)g_scratch_ptr[dev];
    return PH_OK;
}

}  // namespace

extern "C" int ph_solve_essential_5pt(const float* x1, const float* y1, const float* x2, const float* y2,
                                      const int32_t* samples, int B, int S, int N, float* E, int32_t* n_solutions,
                                      void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || S < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || S == 0) return PH_OK;
    if (!x1 || !y1 || !x2 || !y2 || !samples || !E) return PH_ERR_INVALID_ARG;
    if (!ws || ws_bytes < ph_solve_essential_5pt_ws_bytes(B, S) || !ph_aligned(ws, 8)) return PH_ERR_WORKSPACE;
    return solve_5pt_launch(x1, y1, x2, y2, samples, B, S, N, E, n_solutions, (ph5::Poly5*)ws, (hipStream_t)stream);
}

extern "C" int ph_solve_essential_5pt_noalloc(const float* x1, const float* y1, const float* x2, const float* y2,
                                              const int32_t* samples, int B, int S, int N, float* E, int32_t* n_solutions,
                                              void* stream) {
    if (B < 0 || S < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if ((long)B * S > (long)PH_NOALLOC_MAX_SAMPLES) return PH_ERR_TOO_LARGE;     // beyond the static scratch: refuse, never spill
    if (B == 0 || S == 0) return PH_OK;
    if (!x1 || !y1 || !x2 || !y2 || !samples || !E) return PH_ERR_INVALID_ARG;
    ph5::Poly5* polys = nullptr;
    const int rc = noalloc_scratch(&polys);
    if (rc != PH_OK) return rc;
    // nothing in the scratch has to start at zero: the polynomial stage writes the record of every one of the B*S
    // samples (failed ones as n = 0) before the roots stage reads it, on the same stream
    return solve_5pt_launch(x1, y1, x2, y2, samples, B, S, N, E, n_solutions, polys, (hipStream_t)stream);
}
