// triangulate.hip — batched linear (DLT) triangulation and recoverPose-style cheirality vote.
// SURVEY.md §8 row a-11 (no reference code; contract in include/posehot.h).
//
// One thread per (item, candidate, point).  The 4x4 system A (rows x*P[2]-P[0], y*P[2]-P[1] for
// both views) is diagonalised in registers by a one-sided (Hestenes) Jacobi SVD — it works on A
// itself, so the f32 result does not pay the squared condition number of A'A.  The right singular
// vector of the smallest singular value is de-homogenised and stored; the depth signs in both
// views feed a ballot-counted vote used by the host's essential-matrix decomposition.
// HBM traffic per point: 16 B in (+ 96 B of P per item, L2-resident) and 12 B out.
#include <math.h>

#include "common.hpp"

namespace {

constexpr int kSweeps = 3;   // measured against a float64 SVD on the CPU, baselines 0.5 / 0.05 / 0.005 m: 3, 4, 5 and 6 sweeps give the same error to 3 digits (4 until late in round 2: a quarter of the kernel for nothing)

__device__ __forceinline__ void rotate_cols(float (&A)[4][4], float (&V)[4][4], int p, int q) {
    float alpha = 0.f, beta = 0.f, gamma = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        alpha = ph_fma(A[r][p], A[r][p], alpha);
        beta = ph_fma(A[r][q], A[r][q], beta);
        gamma = ph_fma(A[r][p], A[r][q], gamma);
    }
    // rotation angle that zeroes the (p,q) inner product; t = tan(theta), |t| <= 1
    // v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 themselves (1 ulp): the correctly rounded forms expand to a division by scale / fmas /
    // fixup and a range-scaled square root — 30 of a sweep's ~300 instructions were v_div_scale alone — and a Jacobi rotation
    // by an angle that is off in its last bit still orthogonalises: the next sweep takes what is left
    const float zeta = (beta - alpha) * __builtin_amdgcn_rcpf(2.0f * gamma);
    const float az = fabsf(zeta);
    float t = __builtin_amdgcn_rcpf(az + __builtin_amdgcn_sqrtf(ph_fma(zeta, zeta, 1.0f)));
    t = (zeta < 0.0f) ? -t : t;
    t = (gamma == 0.0f || !(az < INFINITY)) ? 0.0f : t;
    const float c = __builtin_amdgcn_rsqf(ph_fma(t, t, 1.0f));
    const float s = c * t;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float ap = A[r][p], aq = A[r][q];
        A[r][p] = ph_fma(c, ap, -s * aq);
        A[r][q] = ph_fma(s, ap, c * aq);
        const float vp = V[r][p], vq = V[r][q];
        V[r][p] = ph_fma(c, vp, -s * vq);
        V[r][q] = ph_fma(s, vp, c * vq);
    }
}

__global__ __launch_bounds__(256) void triangulate_kernel(const float* __restrict__ P1, const float* __restrict__ P2,
                                                         const float* __restrict__ x1, const float* __restrict__ y1,
                                                         const float* __restrict__ x2, const float* __restrict__ y2,
                                                         const int32_t* __restrict__ n_valid, const uint64_t* __restrict__ mask,
                                                         int C, int N, float* __restrict__ XYZ, int32_t* __restrict__ front) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const bool ok = i < nv;
    if (blockIdx.x * 256 + (int)(threadIdx.x & ~63u) >= nv) return;     // a wave beyond the item's valid points (wave-uniform):
                                                                        // the arrays are sized for the capacity, not the count
    if (!XYZ && mask && ((blockIdx.x * 256 + (int)(threadIdx.x & ~63u)) >> 6) < (N + 63) / 64 &&
        mask[(size_t)b * ((N + 63) / 64) + ((blockIdx.x * 256 + (int)(threadIdx.x & ~63u)) >> 6)] == 0ull)
        return;                                                         // vote only, and none of the wave's 64 points is selected
    const float* pa = P1 + ((size_t)b * C + c) * 12;
    const float* pb = P2 + ((size_t)b * C + c) * 12;
    float Pa[12], Pb[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) { Pa[e] = pa[e]; Pb[e] = pb[e]; }
    // Twin candidates.  The four (R, t) candidates of an essential matrix come as (Ra, t), (Ra, -t), (Rb, t), (Rb, -t) with
    // P1 = [I | 0]: negating the fourth column of both projection matrices mirrors the DLT system (A' = A diag(1,1,1,-1)),
    // every Jacobi rotation, the null vector's last component and with it X and both depths — exactly, sign flips being
    // exact in IEEE arithmetic.  So an even candidate whose successor is such a twin computes once and reports for both
    // (X' = -X, the twin's vote = both depths NEGATIVE), and the twin's own blocks leave at once.
    auto twin_of = [&](int ce, int co) {                   // is candidate co the fourth-column mirror of candidate ce?
        const float* qa = P1 + ((size_t)b * C + ce) * 12; const float* ra = P1 + ((size_t)b * C + co) * 12;
        const float* qb = P2 + ((size_t)b * C + ce) * 12; const float* rb = P2 + ((size_t)b * C + co) * 12;
        bool same = true;
#pragma unroll
        for (int e = 0; e < 12; ++e) {
            const bool col4 = (e & 3) == 3;
            same = same && (col4 ? (ra[e] == -qa[e] && rb[e] == -qb[e]) : (ra[e] == qa[e] && rb[e] == qb[e]));
        }
        return same;
    };
    if ((c & 1) && twin_of(c - 1, c)) return;              // wave-uniform (the matrices are per item and candidate)
    const bool has_twin = !(c & 1) && (c + 1 < C) && twin_of(c, c + 1);

    const size_t o = (size_t)b * N + (ok ? i : 0);
    const float u1 = x1[o], v1 = y1[o], u2 = x2[o], v2 = y2[o];

    float A[4][4], V[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        A[0][k] = ph_fma(u1, Pa[8 + k], -Pa[k]);
        A[1][k] = ph_fma(v1, Pa[8 + k], -Pa[4 + k]);
        A[2][k] = ph_fma(u2, Pb[8 + k], -Pb[k]);
        A[3][k] = ph_fma(v2, Pb[8 + k], -Pb[4 + k]);
#pragma unroll
        for (int r = 0; r < 4; ++r) V[r][k] = (r == k) ? 1.0f : 0.0f;
    }
#pragma unroll 1
    for (int sweep = 0; sweep < kSweeps; ++sweep) {
        rotate_cols(A, V, 0, 1); rotate_cols(A, V, 2, 3);
        rotate_cols(A, V, 0, 2); rotate_cols(A, V, 1, 3);
        rotate_cols(A, V, 0, 3); rotate_cols(A, V, 1, 2);
    }
    float best = INFINITY, hx = 0.f, hy = 0.f, hz = 0.f, hw = 1.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float n2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) n2 = ph_fma(A[r][k], A[r][k], n2);
        const bool lt = n2 < best;
        best = lt ? n2 : best;
        hx = lt ? V[0][k] : hx; hy = lt ? V[1][k] : hy; hz = lt ? V[2][k] : hz; hw = lt ? V[3][k] : hw;
    }
    const float rw = 1.0f / hw;                            // one correctly rounded reciprocal for the point itself
    const float X = hx * rw, Y = hy * rw, Z = hz * rw;
    if (ok && XYZ) {
        float* dst = XYZ + (((size_t)b * C + c) * N + i) * 3;
        dst[0] = X; dst[1] = Y; dst[2] = Z;
        if (has_twin) { dst += (size_t)N * 3; dst[0] = -X; dst[1] = -Y; dst[2] = -Z; }
    }
    if (front) {
        const float z1 = ph_fma(Pa[8], X, ph_fma(Pa[9], Y, ph_fma(Pa[10], Z, Pa[11])));
        const float z2 = ph_fma(Pb[8], X, ph_fma(Pb[9], Y, ph_fma(Pb[10], Z, Pb[11])));
        bool sel = ok;
        if (mask && ok) sel = (mask[(size_t)b * ((N + 63) / 64) + (i >> 6)] >> (i & 63)) & 1ull;
        const unsigned long long votes = __ballot(sel && z1 > 0.0f && z2 > 0.0f);
        if ((threadIdx.x & 63) == 0 && votes) atomicAdd(front + (size_t)b * C + c, (int32_t)__popcll(votes));
        if (has_twin) {
            const unsigned long long tv = __ballot(sel && z1 < 0.0f && z2 < 0.0f);
            if ((threadIdx.x & 63) == 0 && tv) atomicAdd(front + (size_t)b * C + c + 1, (int32_t)__popcll(tv));
        }
    }
}

// Stable compaction of kept matches: a 256-thread block per 1024 queries of a pair (grid (chunks, B): a block per PAIR
// walked 8 192 queries in four dependent rounds — 24 us for 64 pairs, on 64 of the 256 CUs).  A block first counts the
// kept matches in front of its chunk (coalesced sweep over idx / keep: at most 64 KB from the L2), then packs its own.
__global__ __launch_bounds__(256) void compact_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ keep,
                                                     const float2* __restrict__ kp_q, const float2* __restrict__ kp_t,
                                                     const int32_t* __restrict__ view_q, const int32_t* __restrict__ view_t,
                                                     int nq, int nt, float* __restrict__ x1, float* __restrict__ y1,
                                                     float* __restrict__ x2, float* __restrict__ y2,
                                                     int32_t* __restrict__ match_q, int32_t* __restrict__ match_t,
                                                     int32_t* __restrict__ n_match) {
    constexpr int U = 4;                                   // chunk = 256 * U queries, element i0 + 256 u + tid on thread tid
    __shared__ int wave_tot[U][4];
    __shared__ int red[4];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int vq = view_q ? view_q[b] : b, vt = view_t ? view_t[b] : b;
    const float2* KQ = kp_q + (size_t)vq * nq;
    const float2* KT = kp_t + (size_t)vt * nt;
    const int32_t* IDX = idx + (size_t)b * nq;
    const int32_t* KEEP = keep + (size_t)b * nq;
    const int i0 = c * (256 * U);
    // this chunk's candidates first (their loads travel with the sweep's)
    int j[U], kp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = i0 + 256 * u + tid;
        j[u] = PH_NO_INDEX; kp[u] = 0;
        if (i < nq) { j[u] = IDX[i]; kp[u] = KEEP[i]; }
    }
    int before = 0;
    for (int i = tid; i < i0; i += 256 * 4) {              // i0 is a multiple of 1024: four loads per array in flight, no tail
        int jj[4], kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { jj[u] = IDX[i + 256 * u]; kk[u] = KEEP[i + 256 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) before += (kk[u] != 0 && jj[u] >= 0 && jj[u] < nt) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    bool k[U];
    int rank[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        k[u] = kp[u] != 0 && j[u] >= 0 && j[u] < nt;
        const unsigned long long bal = __ballot(k[u]);
        rank[u] = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[u][wave] = __popcll(bal);
    }
    if (lane == 0) red[wave] = before;
    __syncthreads();
    float2 a[U], t[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
        if (k[u]) { a[u] = KQ[i0 + 256 * u + tid]; t[u] = KT[j[u]]; }
    int off = red[0] + red[1] + red[2] + red[3];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        int mine = off;
        for (int w = 0; w < wave; ++w) mine += wave_tot[u][w];
        if (k[u]) {
            const size_t o = (size_t)b * nq + mine + rank[u];
            x1[o] = a[u].x; y1[o] = a[u].y; x2[o] = t[u].x; y2[o] = t[u].y;
            if (match_q) match_q[o] = i0 + 256 * u + tid;
            if (match_t) match_t[o] = j[u];
        }
        off += wave_tot[u][0] + wave_tot[u][1] + wave_tot[u][2] + wave_tot[u][3];
    }
    if (c == (int)gridDim.x - 1 && tid == 0) n_match[b] = off;      // the last chunk's block has seen every kept match
}

// models[b][sel[b]][0..stride) -> out[b][0..stride): the winning hypothesis of every item in one small launch
__global__ __launch_bounds__(256) void gather_models_kernel(const float* __restrict__ models, const int32_t* __restrict__ sel,
                                                           int B, int H, int stride, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * stride) return;
    const int b = i / stride, e = i - b * stride;
    const int h = sel ? min(max(sel[b], 0), H - 1) : 0;
    out[i] = models[((size_t)b * H + h) * stride + e];
}

// Stable compaction of the non-empty models of every item (the solvers leave a slot all-zero when its root is not a
// solution: 53 % of the 5-point slots, 55 % of the P3P slots): live models first, in slot order, the tail zero-filled,
// plus the map compact position -> slot.  One 1024-thread block per item, rounds of 1024 * U slots (one round for the
// headline's 9 200 slots): (1) thread t loads slots t, t + 1024, ... of the round — all U * STRIDE loads in flight together —
// and KEEPS them in registers; the waves' ballots go to a table of 64-bit words; (2) wave 0 turns the table into exclusive
// prefix counts; (3) every live slot is written from the registers to base + prefix of its word + rank inside the word.
// STRIDE = floats per model when known at compile time (9, 12), 0 = the run-time value (one slot per thread and round, read
// twice).  One memory round trip and two barriers per round: 64 items x 9 200 slots x 9 floats take 27.5 us (30 with 1 024
// slots per round and every model read for the ballot and again for the copy) — what is left is one CU per item moving
// 660 KB in 36-byte pieces; ph_compact_models_counted below cuts an item over many blocks.
template <int STRIDE>
__global__ __launch_bounds__(1024) void compact_models_kernel(const float* __restrict__ models, int H, int stride_rt,
                                                             float* __restrict__ out, int32_t* __restrict__ slot,
                                                             int32_t* __restrict__ n_live) {
    const int stride = STRIDE > 0 ? STRIDE : stride_rt;
    constexpr int NE = STRIDE > 0 ? STRIDE : 1;            // floats held per slot
    constexpr int U = STRIDE == 0 ? 1 : (STRIDE <= 9 ? 9 : 8);      // slots per thread and round: <= 96 registers of models
    constexpr int kRound = 1024 * U;
    __shared__ unsigned long long mask[kRound / 64];
    __shared__ int pref[kRound / 64];
    __shared__ int total_sh;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* M = models + (size_t)b * H * stride;
    float* O = out + (size_t)b * H * stride;
    int32_t* S = slot + (size_t)b * H;
    int base = 0;                                          // live models before this round (the same in every thread)
    for (int h0 = 0; h0 < H; h0 += kRound) {
        const int span = min(H - h0, kRound), nw = (span + 63) / 64;
        float m[U][NE];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = u * 1024 + tid;
            if ((r >> 6) >= nw) break;                     // wave-uniform: whole waves vote
            uint32_t any = 0;
            if (r < span) {
                if (STRIDE > 0) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) { m[u][e] = M[(size_t)(h0 + r) * stride + e]; any |= __float_as_uint(m[u][e]) << 1; }   // +-0 -> 0
                } else {
                    for (int e = 0; e < stride; ++e) any |= __float_as_uint(M[(size_t)(h0 + r) * stride + e]) << 1;
                }
            }
            const unsigned long long votes = __ballot(any != 0u);
            if (lane == 0) mask[r >> 6] = votes;
        }
        __syncthreads();
        if (wave == 0) {
            const int per = (nw + 63) / 64, w0 = lane * per;
            int mine = 0;
            for (int i = 0; i < per; ++i) mine += (w0 + i < nw) ? __popcll(mask[w0 + i]) : 0;
            int incl = mine;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); incl += (lane >= o) ? v : 0; }
            int run = incl - mine;
            for (int i = 0; i < per; ++i)
                if (w0 + i < nw) { pref[w0 + i] = run; run += __popcll(mask[w0 + i]); }
            if (lane == 63) total_sh = incl;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = u * 1024 + tid;
            if ((r >> 6) >= nw) break;
            const unsigned long long w = mask[r >> 6];
            if ((w >> lane) & 1ull) {
                const int pos = base + pref[r >> 6] + __popcll(w & ((1ull << lane) - 1ull));
                if (STRIDE > 0) {
#pragma unroll
                    for (int e = 0; e < NE; ++e) O[(size_t)pos * stride + e] = m[u][e];
                } else {
                    for (int e = 0; e < stride; ++e) O[(size_t)pos * stride + e] = M[(size_t)(h0 + r) * stride + e];
                }
                S[pos] = h0 + r;
            }
        }
        base += total_sh;
        __syncthreads();                                   // the tables are rewritten by the next round
    }
    for (int pos = base + tid; pos < H; pos += 1024) {
        for (int e = 0; e < stride; ++e) O[(size_t)pos * stride + e] = 0.0f;
        S[pos] = -1;
    }
    if (tid == 0) n_live[b] = base;
}

// The same compaction when the number of live models of every GROUP of `group` consecutive slots is known (the solvers
// report it: n_solutions per minimal sample, 10 slots for the 5-point solver, 4 for P3P): the position of a chunk's first
// live model is a sum over the counts in front of it — a few hundred integers — instead of a pass over the models in front
// of it, so an item is cut into chunks of kChunkGroups * group slots with a block each: (chunks, B) blocks instead of B.
// 64 items x 9 200 slots x 9 floats: 14 us against the one-block-per-item kernel's 27 (whose 64 blocks each move 660 KB through
// one CU's memory path in 36-byte pieces).  Which slots of a chunk are live is still read from the models themselves; the
// counts only place the chunk (a count that disagrees with its group's models leaves a gap or an overlap: precondition).
constexpr int kCountedThreads = 256;
template <int STRIDE, int GROUP>
__global__ __launch_bounds__(kCountedThreads) void compact_models_counted_kernel(const float* __restrict__ models,
                                                                                 const int32_t* __restrict__ counts, int H,
                                                                                 float* __restrict__ out, int32_t* __restrict__ slot,
                                                                                 int32_t* __restrict__ n_live) {
    constexpr int kChunk = 1280;                           // slots per block: a multiple of 64, of 10 and of 4
    static_assert(kChunk % 64 == 0 && kChunk % GROUP == 0 && kChunk % kCountedThreads == 0, "chunk");
    constexpr int U = kChunk / kCountedThreads, kWords = kChunk / 64;
    __shared__ unsigned long long mask[kWords];
    __shared__ int pref[kWords];
    __shared__ int red[2][kCountedThreads / 64];
    const int b = blockIdx.y, c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h0 = c * kChunk, span = min(H - h0, kChunk), nw = (span + 63) / 64;
    const float* M = models + (size_t)b * H * STRIDE;
    float* O = out + (size_t)b * H * STRIDE;
    int32_t* S = slot + (size_t)b * H;
    // the chunk's models: all loads in flight together, kept in registers
    float m[U][STRIDE];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int r = u * kCountedThreads + tid;
        if ((r >> 6) >= nw) break;                         // wave-uniform: whole waves vote
        uint32_t any = 0;
        if (r < span) {
#pragma unroll
            for (int e = 0; e < STRIDE; ++e) { m[u][e] = M[(size_t)(h0 + r) * STRIDE + e]; any |= __float_as_uint(m[u][e]) << 1; }   // +-0 -> 0
        }
        const unsigned long long votes = __ballot(any != 0u);
        if (lane == 0) mask[r >> 6] = votes;
    }
    // live models in front of the chunk, and of the whole item
    const int G = H / GROUP, g0 = h0 / GROUP;
    int before = 0, all = 0;
    for (int g = tid; g < G; g += kCountedThreads) {
        const int n = counts[(size_t)b * G + g];
        all += n;
        before += g < g0 ? n : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { before += __shfl_xor(before, o); all += __shfl_xor(all, o); }
    if (lane == 0) { red[0][wave] = before; red[1][wave] = all; }
    __syncthreads();
    before = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    all = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (tid == 0) {
        int run = 0;
        for (int w = 0; w < nw; ++w) { pref[w] = run; run += __popcll(mask[w]); }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int r = u * kCountedThreads + tid;
        if ((r >> 6) >= nw) break;
        const unsigned long long w = mask[r >> 6];
        if ((w >> lane) & 1ull) {
            const int pos = before + pref[r >> 6] + __popcll(w & ((1ull << lane) - 1ull));
            if (pos < H) {                                 // counts that overstate cannot push a write out of the item
#pragma unroll
                for (int e = 0; e < STRIDE; ++e) O[(size_t)pos * STRIDE + e] = m[u][e];
                S[pos] = h0 + r;
            }
        }
    }
    // the zero tail, dealt over the item's blocks
    all = min(max(all, 0), H);
    for (int pos = all + c * kCountedThreads + tid; pos < H; pos += gridDim.x * kCountedThreads) {
#pragma unroll
        for (int e = 0; e < STRIDE; ++e) O[(size_t)pos * STRIDE + e] = 0.0f;
        S[pos] = -1;
    }
    if (c == 0 && tid == 0) n_live[b] = all;
}

// out[b] = table[b][sel[b]] (-1 when sel[b] is outside [0, H)): compact position of a winner -> its slot
__global__ __launch_bounds__(64) void lookup_index_kernel(const int32_t* __restrict__ table, const int32_t* __restrict__ sel, int B,
                                                         int H, int32_t* __restrict__ out) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int s = sel[b];
    out[b] = (s >= 0 && s < H) ? table[(size_t)b * H + s] : -1;
}

// Cheirality winner and pose-table row of every item: win = argmax_c front[b][c] (first maximum), table row =
// R (9, row-major) | t (3) of P2[b][win] = [R | t], inliers, matches, winning hypothesis, status.
__global__ __launch_bounds__(64) void pose_table_kernel(const float* __restrict__ P2, const int32_t* __restrict__ front, int C,
                                                       const int32_t* __restrict__ best_count,
                                                       const int32_t* __restrict__ n_match,
                                                       const int32_t* __restrict__ best_idx, int B, int min_inliers,
                                                       float* __restrict__ table, int32_t* __restrict__ win) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    int w = 0, best = front[(size_t)b * C];
    for (int c = 1; c < C; ++c) {
        const int f = front[(size_t)b * C + c];
        if (f > best) { best = f; w = c; }
    }
    const float* P = P2 + ((size_t)b * C + w) * 12;
    float* o = table + (size_t)b * 16;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        o[3 * r] = P[4 * r]; o[3 * r + 1] = P[4 * r + 1]; o[3 * r + 2] = P[4 * r + 2];
        o[9 + r] = P[4 * r + 3];
    }
    o[12] = (float)best_count[b];
    o[13] = (float)n_match[b];
    o[14] = (float)best_idx[b];
    o[15] = best_count[b] >= min_inliers ? 1.0f : 0.0f;
    if (win) win[b] = w;
}

// The four (R, t) candidates of an essential matrix E ~ [t]x R in closed form (B. K. P. Horn, "Recovering baseline and
// orientation from essential matrix", 1990), float64, one thread per item:
//   t t' = tr(EE')/2 I - EE'   -> t from its row of largest diagonal (sign: that diagonal entry positive)
//   R    = (Cof(E) -+ [t]x E) / |t|^2,   Cof(E) rows = cross products of the other two rows of E
// Candidates in the order (Ra, t), (Ra, -t), (Rb, t), (Rb, -t), t unit; P2 = [R | t], P1 = [I | 0].  An E that is not
// an essential matrix to working precision gives R that is not orthonormal to the same degree (the 5-point solver's
// E are essential to f32 rounding); an all-zero E gives R = I, t = 0.
__global__ __launch_bounds__(64) void decompose_essential_kernel(const float* __restrict__ models, const int32_t* __restrict__ sel,
                                                                int B, int H, float* __restrict__ P1, float* __restrict__ P2) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const int h = sel ? min(max(sel[b], 0), H - 1) : 0;
    const float* src = models + ((size_t)b * H + h) * 9;
    double E[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) E[i][j] = (double)src[3 * i + j];
    double G[3][3];                                   // EE'
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[i][j] = E[i][0] * E[j][0] + E[i][1] * E[j][1] + E[i][2] * E[j][2];
    const double half_tr = 0.5 * (G[0][0] + G[1][1] + G[2][2]);
    double T[3][3];                                   // t t'
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[i][j] = (i == j ? half_tr : 0.0) - G[i][j];
    int r = 0;
    if (T[1][1] > T[r][r]) r = 1;
    if (T[2][2] > T[r][r]) r = 2;
    double R[2][3][3], t[3];
    if (T[r][r] > 0.0) {
        const double inv = 1.0 / sqrt(T[r][r]);
        double tb[3] = {T[r][0] * inv, T[r][1] * inv, T[r][2] * inv};       // |tb|^2 = tr(EE')/2 up to rounding
        const double n2 = tb[0] * tb[0] + tb[1] * tb[1] + tb[2] * tb[2];
        const double in2 = 1.0 / n2, in1 = 1.0 / sqrt(n2);
        for (int i = 0; i < 3; ++i) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
            for (int j = 0; j < 3; ++j) {
                const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
                const double cof = E[i1][j1] * E[i2][j2] - E[i1][j2] * E[i2][j1];
                const double txe = tb[i1] * E[i2][j] - tb[i2] * E[i1][j];        // ([t]x E)[i][j]
                R[0][i][j] = (cof - txe) * in2;
                R[1][i][j] = (cof + txe) * in2;
            }
            t[i] = tb[i] * in1;
        }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) R[0][i][j] = R[1][i][j] = (i == j) ? 1.0 : 0.0;
            t[i] = 0.0;
        }
    }
    for (int c = 0; c < 4; ++c) {
        float* p2 = P2 + ((size_t)b * 4 + c) * 12;
        const double sg = (c & 1) ? -1.0 : 1.0;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) p2[4 * i + j] = (float)R[c >> 1][i][j];
            p2[4 * i + 3] = (float)(sg * t[i]);
        }
        if (P1) {
            float* p1 = P1 + ((size_t)b * 4 + c) * 12;
            for (int e = 0; e < 12; ++e) p1[e] = (e == 0 || e == 5 || e == 10) ? 1.0f : 0.0f;
        }
    }
}

}  // namespace

extern "C" int ph_decompose_essential(const float* E, const int32_t* sel, int B, int H, float* P1, float* P2, void* stream) {
    if (B < 0 || H < 1) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!E || !P2) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(decompose_essential_kernel, dim3(ph_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, E, sel, B, H, P1, P2);
    return ph_launch_status();
}

extern "C" int ph_triangulate(const float* P1, const float* P2, const float* x1, const float* y1, const float* x2,
                              const float* y2, const int32_t* n_valid, const uint64_t* mask, int B, int C, int N,
                              float* XYZ, int32_t* front, void* stream) {
    if (B < 0 || C < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535 || C > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || C == 0) return PH_OK;
    hipStream_t st = (hipStream_t)stream;
    if (front) ph_zero_words(front, (size_t)B * C, st);
    if (N == 0) return PH_OK;
    if (!P1 || !P2 || !x1 || !y1 || !x2 || !y2 || (!XYZ && !front)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(triangulate_kernel, dim3(ph_ceil_div(N, 256), C, B), dim3(256), 0, st, P1, P2, x1, y1, x2, y2,
                       n_valid, mask, C, N, XYZ, front);
    return ph_launch_status();
}

extern "C" int ph_compact_matches(const int32_t* idx, const int32_t* keep, const float* kp_q, const float* kp_t,
                                  const int32_t* view_q, const int32_t* view_t, int B, int nq, int nt, float* x1,
                                  float* y1, float* x2, float* y2, int32_t* match_q, int32_t* match_t, int32_t* n_match,
                                  void* stream) {
    if (B < 0 || nq < 0 || nt < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0) return PH_OK;
    if (!n_match) return PH_ERR_INVALID_ARG;
    if (nq > 0 && (!idx || !keep || !kp_q || !x1 || !y1 || !x2 || !y2 || (nt > 0 && !kp_t))) return PH_ERR_INVALID_ARG;
    if (!ph_aligned(kp_q, 8) || !ph_aligned(kp_t, 8)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(compact_kernel, dim3(nq > 0 ? ph_ceil_div(nq, 1024) : 1, B), dim3(256), 0, (hipStream_t)stream, idx, keep, (const float2*)kp_q,
                       (const float2*)kp_t, view_q, view_t, nq, nt, x1, y1, x2, y2, match_q, match_t, n_match);
    return ph_launch_status();
}

extern "C" int ph_gather_models(const float* models, const int32_t* sel, int B, int H, int stride, float* out, void* stream) {
    if (B < 0 || H < 1 || stride < 1) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!models || !out) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(gather_models_kernel, dim3(ph_ceil_div(B * stride, 256)), dim3(256), 0, (hipStream_t)stream, models, sel,
                       B, H, stride, out);
    return ph_launch_status();
}

extern "C" int ph_compact_models(const float* models, int B, int H, int stride, float* out, int32_t* slot, int32_t* n_live,
                                 void* stream) {
    if (B < 0 || H < 0 || stride < 1) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0) return PH_OK;
    if (!n_live || (H > 0 && (!models || !out || !slot)) || models == out) return PH_ERR_INVALID_ARG;
    if (stride == 9)
        hipLaunchKernelGGL(compact_models_kernel<9>, dim3(B), dim3(1024), 0, (hipStream_t)stream, models, H, stride, out, slot, n_live);
    else if (stride == 12)
        hipLaunchKernelGGL(compact_models_kernel<12>, dim3(B), dim3(1024), 0, (hipStream_t)stream, models, H, stride, out, slot, n_live);
    else
        hipLaunchKernelGGL(compact_models_kernel<0>, dim3(B), dim3(1024), 0, (hipStream_t)stream, models, H, stride, out, slot, n_live);
    return ph_launch_status();
}

extern "C" int ph_compact_models_counted(const float* models, const int32_t* counts, int group, int B, int H, int stride,
                                         float* out, int32_t* slot, int32_t* n_live, void* stream) {
    if (B < 0 || H < 0 || stride < 1 || group < 1) return PH_ERR_INVALID_ARG;
    if (H % group != 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0) return PH_OK;
    if (!n_live || (H > 0 && (!models || !out || !slot || !counts)) || models == out) return PH_ERR_INVALID_ARG;
    const dim3 grid(ph_ceil_div(H > 0 ? H : 1, 1280), B);
    if (stride == 9 && group == 10)
        hipLaunchKernelGGL((compact_models_counted_kernel<9, 10>), grid, dim3(kCountedThreads), 0, (hipStream_t)stream, models,
                           counts, H, out, slot, n_live);
    else if (stride == 12 && group == 4)
        hipLaunchKernelGGL((compact_models_counted_kernel<12, 4>), grid, dim3(kCountedThreads), 0, (hipStream_t)stream, models,
                           counts, H, out, slot, n_live);
    else                                                   // other shapes: the counts are not needed
        return ph_compact_models(models, B, H, stride, out, slot, n_live, stream);
    return ph_launch_status();
}

extern "C" int ph_lookup_index(const int32_t* table, const int32_t* sel, int B, int H, int32_t* out, void* stream) {
    if (B < 0 || H < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!sel || !out || (H > 0 && !table)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(lookup_index_kernel, dim3(ph_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, table, sel, B, H, out);
    return ph_launch_status();
}

extern "C" int ph_pose_table(const float* P2, const int32_t* front, int C, const int32_t* best_count, const int32_t* n_match,
                             const int32_t* best_idx, int B, int min_inliers, float* table, int32_t* win, void* stream) {
    if (B < 0 || C < 1) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!P2 || !front || !best_count || !n_match || !best_idx || !table) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pose_table_kernel, dim3(ph_ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, P2, front, C, best_count,
                       n_match, best_idx, B, min_inliers, table, win);
    return ph_launch_status();
}
