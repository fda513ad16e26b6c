// fivepoint.hip — RANSAC hypothesis generation on the device: counter-based minimal sampling
// (Philox4x32-10) and Nistér's 5-point essential-matrix solver.  SURVEY.md §8f item 1: there is
// no reference code (main.py only calls cv2.solvePnP); the algebra is the published one
// (D. Nistér, "An efficient solution to the five-point relative pose problem", PAMI 2004).
//
// Per minimal sample, float64 (ten lanes per sample in both stages, see below): null space of the 5x9 epipolar system
// (Householder QR) -> the 10 cubic constraints det(E) = 0 and 2EE'E - tr(EE')E = 0 as a 10x20 coefficient matrix
// -> Gauss-Jordan -> 3x3 polynomial matrix in z -> degree-10 determinant -> all roots by Aberth-Ehrlich iteration
// (polynomial and derivative in f64, the repulsion term between the iterates in f32) -> Newton polish on the real axis
// -> for every real root the E it defines, unit Frobenius norm, stored as f32 in one of the sample's 10 hypothesis
// slots (unused slots are all-zero matrices, which score zero inliers).
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

// Stage 1: TEN LANES per minimal sample build the degree-10 polynomial (sections 1-8 of the core), six samples per
// one-wave block (lanes 60-63 idle) — the lane mapping of the roots stage.  The first version of this stage put one
// thread on a sample and its 2.9 KB of dynamically indexed working arrays in LDS: 52 samples per CU, one wave per CU,
// a dependent chain of ~3 000 LDS round trips — 411 us for 32 x 920 samples whose arithmetic is worth 4 us.  Here every
// array lives in registers under static indices, or in 1.1 KB of LDS per sample where lanes must exchange data:
//   P1  null space of the 5x9 epipolar system by Householder QR of its transpose (no pivoting, straight-line code;
//       every lane runs it redundantly — in a wave that costs what one lane costs); lanes 0-3 then push one unit
//       vector each through the reflections: an orthonormal basis X, Y, Z, W               -> LDS Ep, global basis
//   P3  lanes 0-5: the six entries of EE' (degree 2); lanes 6-8: the three 2x2 minors of det E    -> LDS D[0..8]
//   P4  lane l: coefficient l of trace(EE')                                                  -> LDS D[9]
//   P5  lane l: constraint row l (0: det E = 0; 1 + 3r + c: entry (r, c) of 2EE'E - tr(EE')E = 0), four
//       (degree 2) x (degree 1) products whose operands it reads from LDS                        -> registers A[20]
//   P6  Gauss-Jordan on the 10 x 20 system, row per lane: per pivot column a group arg-max (partial pivoting over
//       the rows not used yet; no swaps — a lane remembers which pivot it became) and a broadcast of the pivot
//       row, both by ds_bpermute
//   P7  rows e..j (pivots 4..9) -> LDS; lanes 0-2: rows k, l, m of B(z)                      -> LDS Bq, global
//   P9  det B(z) (every lane, lane 0 stores) and the degree.
// An orthonormal null-space basis other than the Gauss-Jordan + Gram-Schmidt one of the sequential core changes the
// polynomial's coefficients, not the essential matrices it yields.
namespace coop {

constexpr int kGroup = 10, kGroupsPerWave = 6;
struct alignas(16) SampleLds {
    union {
        double Ep[9][4];          // E entries as linear forms in (x, y, z, 1): Ep[e][k] = basis[k][e]   (P2 .. P5)
        double Bq[3][14];         // rows of B(z): Bx (4) | By (4) | B1 (5) | pad                       (P8 .. P9)
    };
    double D[10][10];             // degree-2 polynomials (P3 .. P5), then the reduced rows by pivot index (P7 .. P8)
};
static_assert(sizeof(SampleLds) == 8 * (42 + 100), "layout");

// operand tables, indexed by the lane within its group
// P3: out = sum_t sgn * Ep[a] (x) Ep[b]  (degree 1 x degree 1)
__device__ const int8_t kP3a[9][3] = {{0, 1, 2}, {0, 1, 2}, {0, 1, 2}, {3, 4, 5}, {3, 4, 5}, {6, 7, 8}, {1, 2, 0}, {2, 0, 0}, {0, 1, 0}};
__device__ const int8_t kP3b[9][3] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {3, 4, 5}, {6, 7, 8}, {6, 7, 8}, {5, 4, 0}, {3, 5, 0}, {4, 3, 0}};
__device__ const int8_t kP3s[9][3] = {{1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, 1, 1}, {1, -1, 0}, {1, -1, 0}, {1, -1, 0}};
// P5: row = sum_t sgn * D[d] (x) Ep[e]  (degree 2 x degree 1); D: 0..5 = EE' entries 00 01 02 11 12 22, 6..8 = minors, 9 = trace
__device__ const int8_t kP5d[10][4] = {{6, 7, 8, 0}, {0, 1, 2, 9}, {0, 1, 2, 9}, {0, 1, 2, 9}, {1, 3, 4, 9}, {1, 3, 4, 9},
                                       {1, 3, 4, 9}, {2, 4, 5, 9}, {2, 4, 5, 9}, {2, 4, 5, 9}};
__device__ const int8_t kP5e[10][4] = {{6, 7, 8, 0}, {0, 3, 6, 0}, {1, 4, 7, 1}, {2, 5, 8, 2}, {0, 3, 6, 3}, {1, 4, 7, 4},
                                       {2, 5, 8, 5}, {0, 3, 6, 6}, {1, 4, 7, 7}, {2, 5, 8, 8}};
__device__ const int8_t kP5s[10][4] = {{1, 1, 1, 0}, {2, 2, 2, -1}, {2, 2, 2, -1}, {2, 2, 2, -1}, {2, 2, 2, -1}, {2, 2, 2, -1},
                                       {2, 2, 2, -1}, {2, 2, 2, -1}, {2, 2, 2, -1}, {2, 2, 2, -1}};

__device__ __forceinline__ void wave_sync() {              // LDS operations of one wave complete in order: compiler fence only
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace coop

__global__ __launch_bounds__(64) void fivepoint_poly_kernel(const float* __restrict__ x1, const float* __restrict__ y1,
                                                            const float* __restrict__ x2, const float* __restrict__ y2,
                                                            const int32_t* __restrict__ samples, long n_samples, int S, int N,
                                                            ph5::Poly5* __restrict__ polys) {
    using namespace coop;
    __shared__ SampleLds lds[kGroupsPerWave];
    // Three groups of ten lanes in either half of the wave (lanes 30, 31, 62, 63 shadow the group before them, l >= 10):
    // a ds_bpermute goes through the crossbar in two passes of 32 lanes, and a group that straddled lane 32 made its
    // exchanges collide with another group's source lanes (SQ_LDS_BANK_CONFLICT: two cycles per shuffle, round 2's layout
    // of six groups side by side; benchmarks/micro/bpermute_conflicts.hip).
    const int lane = threadIdx.x;
    const int half = lane >> 5, r32 = lane & 31, gg = min(r32 / kGroup, 2);
    const int g = 3 * half + gg, l = r32 - kGroup * gg;
    const int base = 32 * half + kGroup * gg;
    const long s = (long)blockIdx.x * kGroupsPerWave + g;
    const bool active = (r32 < 3 * kGroup) && (s < n_samples);
    const long sc = min(s, n_samples - 1);
    const long b = sc / S;
    SampleLds& L = lds[g];
    ph5::Poly5* out = polys + sc;

    // ---- P0: the five correspondences (every lane of the group loads them)
    bool ok = true;
    double M[9][5];                                      // transpose of the epipolar system: column r = row q_r
    {
        const int32_t* si = samples + sc * 5;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            const int i = si[r];
            ok &= (i >= 0 && i < N);
            const size_t o = (size_t)b * N + (ok ? i : 0);
            const double a = x1[o], c = y1[o], d = x2[o], e = y2[o];
            M[0][r] = d * a; M[1][r] = d * c; M[2][r] = d; M[3][r] = e * a; M[4][r] = e * c; M[5][r] = e;
            M[6][r] = a; M[7][r] = c; M[8][r] = 1.0;
        }
    }
    // ---- P1: Householder QR of M (9 x 5); reflector k is kept in place: v_k = M[k..8][k], H_k = I - beta_k v_k v_k'
    double beta[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double n2 = 0.0;
#pragma unroll
        for (int i = k; i < 9; ++i) n2 = fma(M[i][k], M[i][k], n2);
        const double nrm = n2 > 0.0 ? n2 * ph5::fast_rsqrt(n2) : 0.0;     // H_k is orthogonal whatever the last bits of |x| are
        ok &= nrm > 1e-12;                               // a column that is (numerically) dependent: degenerate sample
        const double alpha = M[k][k] > 0.0 ? -nrm : nrm;
        const double v0 = M[k][k] - alpha;
        M[k][k] = v0;
        const double vtv = fma(v0, v0, n2 - (v0 + alpha) * (v0 + alpha));     // |v|^2 = |x|^2 - x0^2 + v0^2
        beta[k] = vtv > 0.0 ? 2.0 * ph5::fast_rcp(vtv) : 0.0;
#pragma unroll
        for (int j = k + 1; j < 5; ++j) {
            double d = 0.0;
#pragma unroll
            for (int i = k; i < 9; ++i) d = fma(M[i][k], M[i][j], d);
            d *= beta[k];
#pragma unroll
            for (int i = k; i < 9; ++i) M[i][j] = fma(-d, M[i][k], M[i][j]);
        }
    }
    // ---- P2: null vector l (lanes 0-3) = H_0 H_1 ... H_4 e_{5+l}
    {
        double nv[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) nv[i] = (i == 5 + l) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            double d = 0.0;
#pragma unroll
            for (int i = k; i < 9; ++i) d = fma(M[i][k], nv[i], d);
            d *= beta[k];
#pragma unroll
            for (int i = k; i < 9; ++i) nv[i] = fma(-d, M[i][k], nv[i]);
        }
        if (l < 4) {
#pragma unroll
            for (int e = 0; e < 9; ++e) L.Ep[e][l] = nv[e];
            if (active)
#pragma unroll
                for (int e = 0; e < 9; ++e) out->basis[l][e] = nv[e];
        }
    }
    wave_sync();
    // ---- P3: lanes 0-8: three (degree 1) x (degree 1) products
    if (l < 9) {
        double acc[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) acc[k] = 0.0;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double sg = (double)kP3s[l][t];
            const double2* ap = reinterpret_cast<const double2*>(L.Ep[kP3a[l][t]]);
            const double2* bp = reinterpret_cast<const double2*>(L.Ep[kP3b[l][t]]);
            const double2 a01 = ap[0], a23 = ap[1], b01 = bp[0], b23 = bp[1];
            const double a[4] = {a01.x * sg, a01.y * sg, a23.x * sg, a23.y * sg}, bb[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[ph5::kMul11[i][j]] = fma(a[i], bb[j], acc[ph5::kMul11[i][j]]);
        }
#pragma unroll
        for (int k = 0; k < 10; ++k) L.D[l][k] = acc[k];
    }
    wave_sync();
    // ---- P4: trace(EE'), coefficient l
    if (l < 10) L.D[9][l] = L.D[0][l] + L.D[3][l] + L.D[5][l];
    wave_sync();
    // ---- P5: constraint row l, columns in Nister's order
    double A[20];
    {
        const int lr = min(l, 9);
        double tmp[20];
#pragma unroll
        for (int k = 0; k < 20; ++k) tmp[k] = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double sg = (double)kP5s[lr][t];
            const double* dp = L.D[kP5d[lr][t]];
            const double2* ep = reinterpret_cast<const double2*>(L.Ep[kP5e[lr][t]]);
            const double2 e01 = ep[0], e23 = ep[1];
            const double e4[4] = {e01.x * sg, e01.y * sg, e23.x * sg, e23.y * sg};
            double d10[10];
#pragma unroll
            for (int i = 0; i < 10; i += 2) { const double2 v = *reinterpret_cast<const double2*>(dp + i); d10[i] = v.x; d10[i + 1] = v.y; }
#pragma unroll
            for (int i = 0; i < 10; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) tmp[ph5::kMul21[i][j]] = fma(d10[i], e4[j], tmp[ph5::kMul21[i][j]]);
        }
#pragma unroll
        for (int c = 0; c < 20; ++c) A[c] = tmp[ph5::kNister[c]];
    }
    wave_sync();                                         // D and Ep are dead from here on (their storage is reused)
    // ---- P6: Gauss-Jordan, row per lane
    int piv = (l < 10) ? -1 : 99;                        // pivot index this lane's row became (-1: not yet); idle lanes never compete
#pragma unroll
    for (int c = 0; c < 10; ++c) {
        // arg-max of |A[c]| over the rows not used yet, on f32 copies (which of two nearly equal candidates becomes the
        // pivot changes the result's rounding, nothing else): a key of the value's upper 28 bits over 15 - lane (ties ->
        // the lowest lane), one 32-bit shuffle and one integer max per row instead of a 64-bit shuffle, a compare and three
        // selects.  A row that is used up has key 0, below every candidate's.
        const uint32_t key = piv < 0 ? ((__float_as_uint((float)fabs(A[c])) & ~15u) | (uint32_t)(15 - l)) : 0u;
        uint32_t bk = 0u;
#pragma unroll
        for (int k = 0; k < kGroup; ++k) bk = max(bk, (uint32_t)__shfl((int)key, base + k));
        const int p = 15 - (int)(bk & 15u);
        ok &= __uint_as_float(bk & ~15u) > 1e-14f;       // false for a NaN too
        const bool me = (l == p);
        double f = A[c];
        if (me) {
            const double inv = ph5::fast_rcp(A[c]);
#pragma unroll
            for (int j = c + 1; j < 20; ++j) A[j] *= inv;
            piv = c;
            f = 0.0;                                     // the pivot row itself stays as it is
        }
#pragma unroll
        for (int j = c + 1; j < 20; ++j) A[j] = fma(-f, __shfl(A[j], base + p), A[j]);
    }
    // ---- P7: the right halves of rows e..j (pivots 4..9) -> LDS, by pivot index
    if (piv >= 4 && piv < 10) {
#pragma unroll
        for (int j = 0; j < 10; ++j) L.D[piv][j] = A[10 + j];
    }
    wave_sync();
    // ---- P8: k = e - z f, l = g - z h, m = i - z j (ascending powers of z)
    if (l < 3) {
        const double* p = L.D[4 + 2 * l];
        const double* r = L.D[5 + 2 * l];
        double q13[13];
        q13[3] = -r[0]; q13[2] = p[0] - r[1]; q13[1] = p[1] - r[2]; q13[0] = p[2];
        q13[7] = -r[3]; q13[6] = p[3] - r[4]; q13[5] = p[4] - r[5]; q13[4] = p[5];
        q13[12] = -r[6]; q13[11] = p[6] - r[7]; q13[10] = p[7] - r[8]; q13[9] = p[8] - r[9]; q13[8] = p[9];
#pragma unroll
        for (int k = 0; k < 13; ++k) L.Bq[l][k] = q13[k];
        if (active) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { out->Bx[l][k] = q13[k]; out->By[l][k] = q13[4 + k]; }
#pragma unroll
            for (int k = 0; k < 5; ++k) out->B1[l][k] = q13[8 + k];
        }
    }
    wave_sync();
    // ---- P9: det B(z), degree 10 (every lane; lane 0 of the group stores)
    {
        double Bx[3][4], By[3][4], B1[3][5], c[11];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { Bx[q][k] = L.Bq[q][k]; By[q][k] = L.Bq[q][4 + k]; }
#pragma unroll
            for (int k = 0; k < 5; ++k) B1[q][k] = L.Bq[q][8 + k];
        }
#pragma unroll
        for (int k = 0; k < 11; ++k) c[k] = 0.0;
        double t7[8], t6[7];
        auto pmul = [](const double* a, int na, const double* bb, int nb, double sgn, double* o) {
#pragma unroll
            for (int i = 0; i < na; ++i)
#pragma unroll
                for (int j = 0; j < nb; ++j) o[i + j] = fma(sgn * a[i], bb[j], o[i + j]);
        };
#pragma unroll
        for (int k = 0; k < 8; ++k) t7[k] = 0.0;
        pmul(By[1], 4, B1[2], 5, 1.0, t7); pmul(B1[1], 5, By[2], 4, -1.0, t7);
        pmul(Bx[0], 4, t7, 8, 1.0, c);
#pragma unroll
        for (int k = 0; k < 8; ++k) t7[k] = 0.0;
        pmul(Bx[1], 4, B1[2], 5, 1.0, t7); pmul(B1[1], 5, Bx[2], 4, -1.0, t7);
        pmul(By[0], 4, t7, 8, -1.0, c);
#pragma unroll
        for (int k = 0; k < 7; ++k) t6[k] = 0.0;
        pmul(Bx[1], 4, By[2], 4, 1.0, t6); pmul(By[1], 4, Bx[2], 4, -1.0, t6);
        pmul(B1[0], 5, t6, 7, 1.0, c);
        double cmax = 0.0;
#pragma unroll
        for (int k = 0; k <= 10; ++k) cmax = fmax(cmax, fabs(c[k]));
        ok = ok && (cmax > 0.0) && (cmax < HUGE_VAL);
        int n = 10;
#pragma unroll
        for (int k = 10; k >= 1; --k)                     // a tiny leading coefficient is a legitimate huge root: keep it
            if (n == k && fabs(c[k]) <= 1e-30 * cmax) n = k - 1;
        if (!ok || n < 1) n = 0;
        if (active && l == 0) {
#pragma unroll
            for (int k = 0; k < 11; ++k) out->c[k] = c[k];
            out->n = n;
        }
    }
}

// Start values of the root iteration (fivepoint_core.hpp's aberth_start10: points on a circle of the geometric-mean
// root radius) by the f32 hardware transcendentals: v_log / v_exp / v_sin / v_cos instead of f64 pow, sin and cos
// (several hundred instructions).  Start values only have to be distinct and of the right magnitude.
__device__ __forceinline__ ph5::cplx aberth_start_fast(const double (&c)[11], int n, int k) {
    const double x = fmin(fmax(fabs(c[0] / ph5::lead_coeff(c, n)), 1e-30), 1e30);   // the radius is clamped to [1e-3, 1e3] anyway
    const float inv_n = __builtin_amdgcn_rcpf((float)n);
    float rad = __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * inv_n);
    rad = fminf(fmaxf(rad, 1e-3f), 1e3f);
    const float rev = (float)k * inv_n + 0.4f / 6.2831853f;                           // angle in revolutions
    return {(double)(rad * __builtin_amdgcn_cosf(rev)), (double)(rad * __builtin_amdgcn_sinf(rev))};
}

// Stage 2: one lane per (sample, root): 6 samples x 10 roots per wave (lanes 60-63 idle).  The ten lanes
// of a sample run the Aberth-Ehrlich iteration together (simultaneous update, roots exchanged by
// shuffles), then every lane turns its own root into an essential matrix and writes slot k.
// 99 VGPR since the repulsion term moved to f32 (168 before): four waves per SIMD.  The occupancy hint stays (3, 8): with
// (4, 4) the compiler drops the kernel's 96 bytes of scratch and the kernel takes 85 instead of 77 us (same box).
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void fivepoint_roots_kernel(const ph5::Poly5* __restrict__ polys, long n_samples,
                                                            float* __restrict__ E_out, int32_t* __restrict__ n_sol) {
    // lanes: three groups of ten in either half of the wave, as in the polynomial stage (no group straddles lane 32)
    const int lane = threadIdx.x;
    const int half = lane >> 5, r32 = lane & 31, gg = min(r32 / 10, 2);
    const int g = 3 * half + gg, k = r32 - 10 * gg;          // k >= 10: idle lanes 30, 31, 62, 63
    const int base = 32 * half + 10 * gg;
    const long s = (long)blockIdx.x * 6 + g;
    const bool active = (k < 10) && (s < n_samples);
    const ph5::Poly5* Pp = polys + (active ? s : 0);
    const int n = Pp->n;
    double c[11];                                          // the polynomial in registers, zero above its degree
#pragma unroll
    for (int q = 0; q < 11; ++q) c[q] = Pp->c[q];                                       // eleven plain loads, then selects: a
#pragma unroll                                                                     // conditional load is a branch each
    for (int q = 0; q < 11; ++q) c[q] = (q <= n) ? c[q] : 0.0;
    const bool live = active && k < n;
    ph5::cplx z = live ? aberth_start_fast(c, n, k) : ph5::cplx{0.0, 0.0};
    double tol = ph5::kAberthTol;
    for (int it = 0; it < ph5::kAberthMaxIter; ++it) {
        // The repulsion term sum_j 1/(z_k - z_j) in FLOAT32 (the roots are exchanged as f32 copies, one quarter-rate
        // reciprocal per term): it only steers the iterates apart — the fixed points of the update are the zeros of p
        // whatever the term is, p and p' stay in f64, and every root is polished on the real axis afterwards — and it
        // was three quarters of an iteration in f64 (nine f64 reciprocals with two Newton steps each, 36 shuffles).
        // The copies are clamped to +-1e18 (squares stay finite); a difference whose square is below the normal range
        // contributes nothing.
        const float zrf = fminf(fmaxf((float)z.re, -1e18f), 1e18f), zif = fminf(fmaxf((float)z.im, -1e18f), 1e18f);
        float sre = 0.0f, sim = 0.0f;
#pragma unroll
        for (int d = 1; d < 10; ++d) {
            int kk = k + d; kk = kk >= 10 ? kk - 10 : kk;
            const int src = base + kk;
            const float dr = zrf - __shfl(zrf, src), di = zif - __shfl(zif, src);
            const float n2 = fmaf(dr, dr, di * di);
            const float r = (kk < n && n2 > 1e-37f) ? __builtin_amdgcn_rcpf(n2) : 0.0f;
            sre = fmaf(dr, r, sre); sim = fmaf(-di, r, sim);
        }
        const ph5::cplx sum = {(double)sre, (double)sim};
        if (it >= 13 && (it - 13) % 3 == 0) tol *= 10.0;           // = ph5::aberth_tol(it)
        bool settled = true;
        if (live) {
            const ph5::cplx w = ph5::aberth_step10(c, z, sum);
            z.re -= w.re; z.im -= w.im;
            settled = fabs(w.re) + fabs(w.im) < tol * (fabs(z.re) + fabs(z.im) + 1e-30);    // relative change of this root
        }
        if (__all(settled)) break;                                 // wave-uniform exit
    }
    float e9[9];
    bool real = false;
    if (live) real = ph5::root_to_E10(c, *Pp, z, e9);
    if (active) {
        float* eo = E_out + (size_t)s * 90 + 9 * k;
        for (int e = 0; e < 9; ++e) eo[e] = real ? e9[e] : 0.0f;
    }
    if (n_sol) {
        const unsigned long long votes = __ballot(real);
        if (active && k == 0) n_sol[s] = (int32_t)__popcll((votes >> base) & 0x3ffull);
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
    const long n_samples = (long)B * S;
    hipLaunchKernelGGL(fivepoint_poly_kernel, dim3((unsigned)((n_samples + 5) / 6)), dim3(64), 0, st, x1, y1, x2, y2, samples,
                       n_samples, S, N, polys);
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
