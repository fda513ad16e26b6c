// score.hip — RANSAC hypothesis scoring: Sampson error (essential matrix) and reprojection
// error (PnP), inlier counts by wave ballot.  SURVEY.md §8 row a-10.  The PnP residual is the
// reference's reprojection error (validate_pose.py:260-288) evaluated on undistorted points;
// the essential-matrix side has no reference code (contract: include/posehot.h).
//
// Work cut: points live on the lanes (coalesced SoA loads, PPL points per lane held in
// registers), the hypothesis is wave-uniform.  A wave owns HPW hypotheses and walks the point
// range of its block in chunks of 64*PPL; per (hypothesis, chunk) it does PPL residual tests,
// each followed by one 64-bit ballot whose popcount is added to a scalar counter.
//   grid = (ceil(H / (4*HPW)), point_splits, B), block = 256 threads = 4 waves.
// With one point split the counter is stored directly; with several (few items, many points)
// the launch first zeroes count[] and the waves add their partial counts atomically (integer
// adds: the result does not depend on arrival order).
#include <math.h>

#include "common.hpp"

namespace {

constexpr int kHPW = 8;   // hypotheses per wave
constexpr int kPPL = 8;   // points per lane per chunk

// ---- the two residual tests; op order = DESIGN.md "Arithmetic contracts" -----------------
__device__ __forceinline__ bool sampson_inlier(const float (&E)[9], float x1, float y1, float x2, float y2, float thr_sq) {
    const float e0 = ph_fma(E[0], x1, ph_fma(E[1], y1, E[2]));
    const float e1 = ph_fma(E[3], x1, ph_fma(E[4], y1, E[5]));
    const float e2 = ph_fma(E[6], x1, ph_fma(E[7], y1, E[8]));
    const float t0 = ph_fma(E[0], x2, ph_fma(E[3], y2, E[6]));
    const float t1 = ph_fma(E[1], x2, ph_fma(E[4], y2, E[7]));
    const float num = ph_fma(x2, e0, ph_fma(y2, e1, e2));
    const float den = ph_fma(e0, e0, ph_fma(e1, e1, ph_fma(t0, t0, t1 * t1)));
    return (num * num) < (thr_sq * den);
}

__device__ __forceinline__ bool pnp_inlier(const float (&P)[12], float X, float Y, float Z, float u, float v, float thr_sq) {
    const float pu = ph_fma(P[0], X, ph_fma(P[1], Y, ph_fma(P[2], Z, P[3])));
    const float pv = ph_fma(P[4], X, ph_fma(P[5], Y, ph_fma(P[6], Z, P[7])));
    const float w = ph_fma(P[8], X, ph_fma(P[9], Y, ph_fma(P[10], Z, P[11])));
    const float du = ph_fma(-u, w, pu);
    const float dv = ph_fma(-v, w, pv);
    const float e2 = ph_fma(du, du, dv * dv);
    // (w > 0) && (e2 < thr_sq * (w * w)) in ONE compare (compares issue at half rate: two of them were 4 of a test's 19 issue
    // slots): w * |w| is w * w bit for bit when w > 0 and <= 0 (or NaN) otherwise, and e2 >= 0 (or NaN) is below no such bound.
    // Needs thr_sq > 0, which the launch checks.
    return e2 < thr_sq * (w * fabsf(w));
}

// kSeedDiv: the seed launch of the bail-out scoring takes the first 1 / kSeedDiv of the hypothesis block.  Minimal-sample essential
// matrices: few models come near the best count (17 of 4 330 within 10 %), a sixteenth of them is what it takes to hold one
// (benchmarks/dbg/score_time.py: 64 / 128 / 256 / 576 seed models -> 0.587 / 0.546 / 0.531 / 0.541 of the exhaustive work);
// P3P poses on 70 % inliers: a third of the models are good ones, a 64th of the block is seed enough (cfg 5: 7.28 against 7.49 ms).
// Seed sizes that make the seed launch exactly one round of resident blocks (512 or 448 models at 64 pairs) change nothing: one
// stream 1.796 / 1.793 against 1.801 ms per step, three streams 1.667 / 1.649 against 1.642.
struct EssentialModel {
    static constexpr int kParams = 9, kArrays = 4, kSeedDiv = 16;
    __device__ static __forceinline__ bool test(const float (&m)[9], const float (&p)[4], float thr_sq) {
        return sampson_inlier(m, p[0], p[1], p[2], p[3], thr_sq);
    }
};
struct PnpModel {
    static constexpr int kParams = 12, kArrays = 5, kSeedDiv = 64;
    __device__ static __forceinline__ bool test(const float (&m)[12], const float (&p)[5], float thr_sq) {
        return pnp_inlier(m, p[0], p[1], p[2], p[3], p[4], thr_sq);
    }
};

template <int NA> struct PointArrays { const float* a[NA]; };

// Bail = true (only with one point split): a hypothesis is abandoned as soon as its count so far plus the points not
// yet seen cannot reach the best count known for the item.  That knowledge travels by gossip: the item owns
// PH_BAILOUT_SLOTS/64 counters, 256 bytes apart (one hot word per item serialises the whole launch in the L2: measured
// 6x slower than no bail-out at all); every wave walks round them, one per point chunk — the read is issued a chunk
// ahead — raising the counter to its own best running count (a lower bound of a final count) and taking the larger
// value along.  An abandoned hypothesis reports the count it had — a lower bound below the maximum — so argmax, ties
// included, is that of the exhaustive counts; stale knowledge only delays a bail-out.
constexpr int kRep = PH_BAILOUT_SLOTS / 64, kRepStride = 64;
template <class Model, bool Bail>
__global__ __launch_bounds__(256) void score_kernel(const float* __restrict__ models, PointArrays<Model::kArrays> pts,
                                                   const int32_t* __restrict__ n_valid, int H, int N, int psplits,
                                                   int pts_per_split, float thr_sq, int32_t* __restrict__ count,
                                                   int32_t* __restrict__ best_lb, int h_begin, int h_end, int tell_all) {
    constexpr int NP = Model::kParams, NA = Model::kArrays;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.z;
    const int h0 = h_begin + (blockIdx.x * 4 + wave) * kHPW;
    if (h0 >= h_end) return;                               // wave-uniform; no barriers below
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    const int p_begin = blockIdx.y * pts_per_split;
    const int p_end = min(nv, p_begin + pts_per_split);

    // the wave's hypotheses stay in registers for the whole kernel (tail waves re-score H-1)
    // An all-zero model (the solvers' empty solution slot) passes neither test for any point (0 < 0, w > 0): its
    // count is 0 by the contract's own arithmetic, so its residual tests are skipped (wave-uniform branch).
    float m[kHPW][NP];
    bool live[kHPW];
#pragma unroll
    for (int k = 0; k < kHPW; ++k) {
        const float* mb = models + ((size_t)b * H + min(h0 + k, H - 1)) * NP;
        uint32_t any = 0;
#pragma unroll
        for (int e = 0; e < NP; ++e) any |= __float_as_uint(mb[e]) << 1;      // +-0 -> 0
        live[k] = __builtin_amdgcn_readfirstlane(any) != 0u;
#pragma unroll
        for (int e = 0; e < NP; ++e) {
            m[k][e] = mb[e];
            // pin the (wave-uniform) value in a VGPR: as SGPR operands the 72-96 model words overflow the
            // scalar file and every fma with two of them needs a v_mov (one constant-bus read per VALU op)
            asm volatile("" : "+v"(m[k][e]));
        }
    }
    int cnt[kHPW];
    bool wave_live = false;                                // a wave of empty slots (the tail of a compacted block) leaves at once
#pragma unroll
    for (int k = 0; k < kHPW; ++k) { cnt[k] = 0; wave_live |= live[k]; }

    __amdgpu_buffer_rsrc_t rsrc[NA];                       // the item's rows of the point arrays, valid points only
#pragma unroll
    for (int a = 0; a < NA; ++a)
        rsrc[a] = __builtin_amdgcn_make_buffer_rsrc((void*)(pts.a[a] + (size_t)b * N), 0, max(p_end, 0) * 4, 0x00020000);
    int known = 0;                                         // best count of the item this wave knows of
    int32_t* lbp = Bail ? best_lb + (size_t)b * PH_BAILOUT_SLOTS : nullptr;
    int rot = (blockIdx.x * 4 + wave) & (kRep - 1);
    int pending = 0;
    if (Bail) pending = __hip_atomic_load(lbp + rot * kRepStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int c0 = p_begin; c0 < p_end && wave_live; c0 += 64 * kPPL) {
        if (Bail) {
            const int seen = __builtin_amdgcn_readfirstlane(pending);
            if (known > seen && lane == 0) atomicMax(lbp + rot * kRepStride, known);
            known = max(known, seen);
            rot = (rot + 1) & (kRep - 1);
            pending = __hip_atomic_load(lbp + rot * kRepStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int remaining = p_end - c0;
            bool any_live = false;
#pragma unroll
            for (int k = 0; k < kHPW; ++k) {
                live[k] = live[k] && (cnt[k] + remaining >= known);
                any_live |= live[k];
            }
            if (!any_live) break;
        }
        // Raw buffer loads: one 32-bit lane offset per chunk, the point's position inside the chunk in the instruction's
        // immediate, reads beyond the item's valid points return 0 without touching memory (then set to NaN below) — no
        // 64-bit vector address arithmetic, no clamping (it was ~100 of a chunk's ~1 300 vector instructions).
        float pt[kPPL][NA];
        const unsigned voff = (unsigned)(c0 + lane) * 4u;
#pragma unroll
        for (int p = 0; p < kPPL; ++p)
#pragma unroll
            for (int a = 0; a < NA; ++a)
                pt[p][a] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc[a], voff + 256u * p, 0, 0));
        if (c0 + 64 * kPPL > p_end) {                       // the item's last chunk (wave-uniform)
#pragma unroll
            for (int p = 0; p < kPPL; ++p) pt[p][0] = (c0 + p * 64 + lane < p_end) ? pt[p][0] : NAN;   // NaN never passes '<'
        }
#pragma unroll
        for (int k = 0; k < kHPW; ++k) {
            if (!live[k]) continue;
#pragma unroll
            for (int p = 0; p < kPPL; ++p) cnt[k] += __popcll(__ballot(Model::test(m[k], pt[p], thr_sq)));
        }
        if (Bail) {
            int lb = cnt[0];
#pragma unroll
            for (int k = 1; k < kHPW; ++k) lb = max(lb, cnt[k]);
            known = max(known, lb);
        }
    }
    if (Bail && known > 0) {
        // the seed launch tells every counter what it found (its waves are few); the main launch one more counter
        if (tell_all) { if (lane < kRep) atomicMax(lbp + lane * kRepStride, known); }
        else if (lane == 0) atomicMax(lbp + rot * kRepStride, known);
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kHPW; ++k) {
            if (h0 + k < h_end) {
                int32_t* dst = count + (size_t)b * H + h0 + k;
                if (psplits == 1) *dst = cnt[k]; else atomicAdd(dst, cnt[k]);
            }
        }
    }
}

// one model per item -> 64-bit inlier words; one wave per 64 points.
template <class Model>
__global__ __launch_bounds__(256) void mask_kernel(const float* __restrict__ models, const int32_t* __restrict__ sel, int H,
                                                  PointArrays<Model::kArrays> pts, const int32_t* __restrict__ n_valid,
                                                  int N, int words, float thr_sq, uint64_t* __restrict__ mask) {
    constexpr int NP = Model::kParams, NA = Model::kArrays;
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.y;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= words) return;
    const int nv = n_valid ? min(max(n_valid[b], 0), N) : N;
    int s = sel ? sel[b] : 0;
    s = min(max(s, 0), H - 1);
    const float* mb = models + ((size_t)b * H + s) * NP;
    float m[NP];
#pragma unroll
    for (int e = 0; e < NP; ++e) m[e] = mb[e];
    const int i = w * 64 + lane;
    float p[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) p[a] = (i < nv) ? pts.a[a][(size_t)b * N + i] : NAN;
    const uint64_t bits = __ballot(Model::test(m, p, thr_sq));
    if (lane == 0) mask[(size_t)b * words + w] = bits;
}

// argmax over hypotheses, ties -> lowest index: max over key = (count << 32) | ~h.  One 1024-thread block per item, ten
// loads in flight per thread: the headline's 9 200 counts are ONE round trip (256 threads x 8 loads were five: 12 -> 6 us).
constexpr int kSelThreads = 1024, kSelLoads = 10;
__global__ __launch_bounds__(kSelThreads) void select_best_kernel(const int32_t* __restrict__ count, int H,
                                                                 int32_t* __restrict__ best_idx, int32_t* __restrict__ best_count,
                                                                 const int32_t* __restrict__ table) {
    __shared__ unsigned long long red[kSelThreads / 64];
    const int b = blockIdx.x;
    unsigned long long best = 0;
    for (int h0 = 0; h0 < H; h0 += kSelThreads * kSelLoads) {
        int c[kSelLoads];
#pragma unroll
        for (int u = 0; u < kSelLoads; ++u) { const int hyp = h0 + kSelThreads * u + (int)threadIdx.x; c[u] = (hyp < H) ? count[(size_t)b * H + hyp] : -1; }
#pragma unroll
        for (int u = 0; u < kSelLoads; ++u) {
            const int hyp = h0 + kSelThreads * u + (int)threadIdx.x;
            if (hyp < H) {
                const unsigned long long key = ((unsigned long long)(uint32_t)max(c[u], 0) << 32) | (uint32_t)(0xffffffffu - (uint32_t)hyp);
                best = key > best ? key : best;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kSelThreads / 64; ++k) best = red[k] > best ? red[k] : best;
        if (H > 0) {
            const int32_t h = (int32_t)(0xffffffffu - (uint32_t)(best & 0xffffffffu));
            best_idx[b] = table ? table[(size_t)b * H + h] : h;       // table: the slot a compact position came from
            best_count[b] = (int32_t)(best >> 32);
        } else {
            best_idx[b] = PH_NO_INDEX;
            best_count[b] = 0;
        }
    }
}

struct ScorePlan { int psplits, pts_per_split; };
ScorePlan plan_score(int B, int H, int N) {
    ScorePlan p;
    const long blocks = (long)B * ph_ceil_div(H, 4 * kHPW);
    const long target = 1024;                                  // 4 blocks (16 waves) per CU
    int s = (int)((target + blocks - 1) / blocks);
    const int chunk = 64 * kPPL;
    const int max_s = ph_ceil_div(N, 2 * chunk);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    p.pts_per_split = ph_ceil_div(ph_ceil_div(N, s), chunk) * chunk;
    p.psplits = ph_ceil_div(N, p.pts_per_split);
    if (p.psplits < 1) p.psplits = 1;
    return p;
}

template <class Model>
int score_launch(const float* models, PointArrays<Model::kArrays> pts, const int32_t* n_valid, int B, int H, int N,
                 float thr_sq, int32_t* count, int32_t* best_lb, void* stream) {
    if (B < 0 || H < 0 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || H == 0) return PH_OK;
    if (!models || !count) return PH_ERR_INVALID_ARG;
    for (int a = 0; a < Model::kArrays; ++a) if (N > 0 && !pts.a[a]) return PH_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const ScorePlan p = plan_score(B, H, N > 0 ? N : 1);
    const bool none = N == 0 || !(thr_sq > 0.0f);              // no point, or a bound no residual is below: every count is 0
    if (p.psplits > 1 || none) {
        ph_zero_words(count, (size_t)B * H, st);
        if (none) return PH_OK;
    }
    const dim3 grid(ph_ceil_div(H, 4 * kHPW), p.psplits, B);
    if (best_lb && p.psplits == 1) {
        ph_zero_words(best_lb, (size_t)B * PH_BAILOUT_SLOTS, st);
        // Seed launch: the first sixteenth of the hypotheses is scored first (with bail-out among themselves) and leaves
        // its best count in every counter, so the main launch knows a real final count from its first chunk on — all
        // waves of one launch start together, and running counts alone let a hopeless hypothesis go only past half
        // of its points.
        int h_seed = 0;
        if (H >= 1024) h_seed = ph_ceil_div(H / Model::kSeedDiv, 4 * kHPW) * (4 * kHPW);
        if (h_seed > 0)
            hipLaunchKernelGGL((score_kernel<Model, true>), dim3(h_seed / (4 * kHPW), 1, B), dim3(256), 0, st, models, pts,
                               n_valid, H, N, 1, p.pts_per_split, thr_sq, count, best_lb, 0, h_seed, 1);
        hipLaunchKernelGGL((score_kernel<Model, true>), dim3(ph_ceil_div(H - h_seed, 4 * kHPW), 1, B), dim3(256), 0, st, models,
                           pts, n_valid, H, N, 1, p.pts_per_split, thr_sq, count, best_lb, h_seed, H, 0);
    } else                    // several point splits add partial counts: no bound to bail out on, exhaustive counts
        hipLaunchKernelGGL((score_kernel<Model, false>), grid, dim3(256), 0, st, models, pts, n_valid, H, N, p.psplits,
                           p.pts_per_split, thr_sq, count, best_lb, 0, H, 0);
    return ph_launch_status();
}

template <class Model>
int mask_launch(const float* models, const int32_t* sel, int H, PointArrays<Model::kArrays> pts, const int32_t* n_valid,
                int B, int N, float thr_sq, uint64_t* mask, void* stream) {
    if (B < 0 || H < 1 || N < 0) return PH_ERR_INVALID_ARG;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || N == 0) return PH_OK;
    if (!models || !mask) return PH_ERR_INVALID_ARG;
    for (int a = 0; a < Model::kArrays; ++a) if (!pts.a[a]) return PH_ERR_INVALID_ARG;
    const int words = ph_ceil_div(N, 64);
    if (!(thr_sq > 0.0f)) {                                    // as in score_launch: no inlier under such a bound
        ph_zero_words(mask, (size_t)B * words * 2, (hipStream_t)stream);
        return ph_launch_status();
    }
    hipLaunchKernelGGL(mask_kernel<Model>, dim3(ph_ceil_div(words, 4), B), dim3(256), 0, (hipStream_t)stream, models, sel,
                       H, pts, n_valid, N, words, thr_sq, mask);
    return ph_launch_status();
}

}  // namespace

extern "C" int ph_score_essential(const float* E, const float* x1, const float* y1, const float* x2, const float* y2,
                                  const int32_t* n_valid, int B, int H, int N, float thr_sq, int32_t* count,
                                  void* stream) {
    return score_launch<EssentialModel>(E, PointArrays<4>{{x1, y1, x2, y2}}, n_valid, B, H, N, thr_sq, count, nullptr, stream);
}

extern "C" int ph_score_essential_bailout(const float* E, const float* x1, const float* y1, const float* x2, const float* y2,
                                          const int32_t* n_valid, int B, int H, int N, float thr_sq, int32_t* count,
                                          int32_t* best_lb, void* stream) {
    if (B > 0 && !best_lb) return PH_ERR_INVALID_ARG;
    return score_launch<EssentialModel>(E, PointArrays<4>{{x1, y1, x2, y2}}, n_valid, B, H, N, thr_sq, count, best_lb, stream);
}

extern "C" int ph_score_pnp(const float* P, const float* X, const float* Y, const float* Z, const float* u,
                            const float* v, const int32_t* n_valid, int B, int H, int N, float thr_sq, int32_t* count,
                            void* stream) {
    return score_launch<PnpModel>(P, PointArrays<5>{{X, Y, Z, u, v}}, n_valid, B, H, N, thr_sq, count, nullptr, stream);
}

extern "C" int ph_score_pnp_bailout(const float* P, const float* X, const float* Y, const float* Z, const float* u,
                                    const float* v, const int32_t* n_valid, int B, int H, int N, float thr_sq,
                                    int32_t* count, int32_t* best_lb, void* stream) {
    if (B > 0 && !best_lb) return PH_ERR_INVALID_ARG;
    return score_launch<PnpModel>(P, PointArrays<5>{{X, Y, Z, u, v}}, n_valid, B, H, N, thr_sq, count, best_lb, stream);
}

extern "C" int ph_select_best(const int32_t* count, int B, int H, int32_t* best_idx, int32_t* best_count, void* stream) {
    if (B < 0 || H < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!best_idx || !best_count || (H > 0 && !count)) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(select_best_kernel, dim3(B), dim3(kSelThreads), 0, (hipStream_t)stream, count, H, best_idx, best_count, nullptr);
    return ph_launch_status();
}

extern "C" int ph_select_best_mapped(const int32_t* count, const int32_t* table, int B, int H, int32_t* best_idx,
                                     int32_t* best_count, void* stream) {
    if (B < 0 || H < 0) return PH_ERR_INVALID_ARG;
    if (B == 0) return PH_OK;
    if (!best_idx || !best_count || (H > 0 && (!count || !table))) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(select_best_kernel, dim3(B), dim3(kSelThreads), 0, (hipStream_t)stream, count, H, best_idx, best_count, table);
    return ph_launch_status();
}

extern "C" int ph_inlier_mask_essential(const float* E, const int32_t* sel, int H, const float* x1, const float* y1,
                                        const float* x2, const float* y2, const int32_t* n_valid, int B, int N,
                                        float thr_sq, uint64_t* mask, void* stream) {
    return mask_launch<EssentialModel>(E, sel, H, PointArrays<4>{{x1, y1, x2, y2}}, n_valid, B, N, thr_sq, mask, stream);
}

extern "C" int ph_inlier_mask_pnp(const float* P, const int32_t* sel, int H, const float* X, const float* Y,
                                  const float* Z, const float* u, const float* v, const int32_t* n_valid, int B, int N,
                                  float thr_sq, uint64_t* mask, void* stream) {
    return mask_launch<PnpModel>(P, sel, H, PointArrays<5>{{X, Y, Z, u, v}}, n_valid, B, N, thr_sq, mask, stream);
}
