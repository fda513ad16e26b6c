// match_l2_filter.hip — the squared-L2 matcher as "fp16 candidate filter + exact f32 re-rank".
// Same contract and bit-identical outputs as the f32 MFMA kernel of match_l2.hip; the
// matrix work runs on v_mfma_f32_32x32x16_f16 (16x the f32 MFMA rate, ONE product per algorithmic product) and
// only a handful of candidates per query are re-evaluated with the contract's f32 fma chain.
//
// Filter.  Per pair, the train rows and the query rows are scaled by powers of two st, sq that bring the largest
// row norm of either side into [2^14, 2^15) (exact in f32; every element then lies inside fp16's normal range down
// to 2^-29 of the largest norm), rounded to fp16 (nearest), and the score  s = q.t - |t|^2/2  is approximated by
//     sA = sq*st * (-|t|^2/2) + sum_k fp16(sq*q_k) * fp16(st*t_k)        (fp16 products are exact in f32; f32 accumulate)
// so sA ~ sq*st*s.  Error, in unscaled units: the two roundings contribute at most (2^-10 + 2^-22) sum|q_k||t_k|,
// elements below fp16's normal range (flushed or not) at most 2^-24 Tmax (|q| + Qmax), the f32 accumulation of the
// MFMAs and the rounding of the contract's own chain less than 2^-15 (|q||t| + |t|^2/2).  Hence
//     |sA/(sq*st) - s| <= e_i := (2^-10 + 2^-13) (|q_i| Tmax + Tmax^2/2) + 2^-23 Tmax (|q_i| + Qmax),
// Tmax, Qmax = largest row norm of the pair's train / query view.  (On SIFT-like unit-norm data the observed error is
// a tenth of e_i; with three bf16 split products instead — the previous version of this file — the band was 2^-13
// and the candidate count 9 instead of 11 per query, at three times the MFMA work.)
// If a2 is ANY lower bound of the second-largest sA of the query (e.g. the second-largest seen so far by one
// lane), both of the exact top-2 rows have sA >= a2 - 2 e_i: such rows are appended to a candidate list.  Every
// (query, train split, lane half) owns a private list segment and counts in a register, so an append is one
// fire-and-forget store: no atomics, nothing to wait for.  A short priming pass over the first tiles sets the bound
// before anything is appended, and those tiles are seen again last, under the final bound, so a query collects ~8
// candidates out of 8192 rows.  A list entry is (row, sA).
// The scales need the row norms' maxima.  A caller that hands in ph_row_sqnorm's norms gets them from a reduction over
// those (l2_filter_prep_kernel); with norm_q = norm_t = NULL the call computes the norms itself inside the passes that
// read the two tables anyway, on provisional scales that are checked afterwards ("norms computed inside the call" below).
//
// Re-rank.  l2_rerank_kernel first prunes the list with the FINAL bound (second-largest listed sA - 2 e_i: 2-3 rows
// survive), then evaluates the contract's exact chain (k order 8c+{0,4,1,5,2,6,3,7}, start
// -|t|^2/2) for the listed rows, takes (max s, lowest index on ties) and the runner-up, and writes idx/d1sq/d2sq
// exactly as the f32 kernel does.  A list that overflowed its capacity (adversarially ordered input), or a pair whose
// norms are so extreme that the scales leave f32's comfortable range, makes the query fall back to the exact chain
// over ALL rows inside the same kernel: slower, never wrong.
#include <math.h>

#include <type_traits>

#include "common.hpp"
#include "match_l2_filter.hpp"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x2 __attribute__((ext_vector_type(2)));

constexpr int kTR = 32;       // train rows per MFMA tile
constexpr int kSub = 2;       // MFMA tiles per staged tile (one barrier per staged tile)
constexpr int kSR = kTR * kSub;   // train rows per staged tile
constexpr int kQT = 2;        // 32-query column tiles per wave
constexpr int kDim = 128;
constexpr int kCPR = kDim / 8;  // 16-byte fp16 chunks per row
constexpr int kNS = kDim / 16;  // MFMA k-steps per row
constexpr int kCapH = 32;     // list entries per (query, split, lane half)
constexpr int kBlocksPerCU = 2;   // blocks per CU the split count of a SMALL batch aims at (3, the residency, times the same: 8 pairs
                                  // 6 % faster, 16 pairs 7 % slower)
constexpr int kDepth = 2;         // staged tiles copied ahead of the one being worked on (3: same time; 4: slower)
constexpr int kPrimeRows = 512;   // rows of the priming pass (per split); 128 and 256 time the same, 1024 is 5 % slower

// ---- per-pair scales and the error band (one definition for the three kernels) -------------------------------------
struct PairScale {
    float sq, st, s;      // powers of two; s = sq * st
    bool bad;             // norms too extreme for the scaled arithmetic: every query takes the exact path
};
__device__ __forceinline__ PairScale pair_scale(float Tmax, float Qmax) {
    int et = 0, eq = 0;
    if (Tmax > 0.0f) frexpf(Tmax, &et); else et = 15;
    if (Qmax > 0.0f) frexpf(Qmax, &eq); else eq = 15;
    const int kt = 15 - et, kq = 15 - eq;              // M < 2^e  ->  M * 2^(15-e) in [2^14, 2^15)
    PairScale p;
    p.bad = !(Tmax < INFINITY) || !(Qmax < INFINITY) || abs(kt) > 60 || abs(kq) > 60;
    p.st = p.bad ? 1.0f : ldexpf(1.0f, kt);
    p.sq = p.bad ? 1.0f : ldexpf(1.0f, kq);
    p.s = p.sq * p.st;
    return p;
}
// 2 e_i in scaled units
__device__ __forceinline__ float band2_scaled(float qnorm_sq, float Tmax, float Qmax, float s) {
    const float qn = sqrtf(qnorm_sq);
    const float e = (1.0f / 1024.0f + 1.0f / 8192.0f) * (qn * Tmax + 0.5f * Tmax * Tmax) + (1.0f / 8388608.0f) * Tmax * (qn + Qmax);
    return 2.0f * e * s;
}

// 8 consecutive f32, scaled by a power of two -> 8 fp16 (round to nearest even)
__device__ __forceinline__ uint4 to_f16x8(const float4 a, const float4 b, float scale) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t h[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const _Float16 lo = (_Float16)(x[2 * p] * scale), hi = (_Float16)(x[2 * p + 1] * scale);
        h[p] = (uint32_t)__builtin_bit_cast(unsigned short, lo) | ((uint32_t)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    return make_uint4(h[0], h[1], h[2], h[3]);
}

// max as plain C++, never inline asm: the operands are fresh MFMA results, and the compiler pads the MFMA -> VALU
// read hazard only for instructions it can see into (an asm v_max reading the accumulator right behind the last MFMA
// reads stale registers).  This file is built with -fno-honor-nans, so fmaxf is a single v_max_f32.
__device__ __forceinline__ float vmax(float a, float b) { return fmaxf(a, b); }

// The value of x in the lane of the same number of the lower (lo) and of the upper (hi) half of the wave, in every lane:
// one v_permlane32_swap (VALU) instead of a ds_bpermute round trip through the LDS pipe, the busiest unit of the filter.
// Callers use lo and hi symmetrically (max, min, median), so which one is "mine" does not matter.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r.x); hi = __uint_as_float(r.y);
}

__device__ __forceinline__ f32x16 mfma_f16(const uint4 a, const uint4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// Tmax, Qmax per pair; with zero_cnt also zeroes the per-query slot counters of the launch (only needed when no filter
// kernel follows: the filter writes the counter of every (live query, slot) itself, empty splits included)
__global__ __launch_bounds__(256) void l2_filter_prep_kernel(const float* __restrict__ tnorm, const float* __restrict__ qnorm,
                                                            const int32_t* __restrict__ view_t, const int32_t* __restrict__ view_q,
                                                            const int32_t* __restrict__ n_t, const int32_t* __restrict__ n_q,
                                                            int nq, int nt, int slots, int B, int zero_cnt,
                                                            float* __restrict__ tqmax, int32_t* __restrict__ cnt) {
    __shared__ float red[2][4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int vt = view_t ? view_t[b] : b, vq = view_q ? view_q[b] : b;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const int nqv = n_q ? min(max(n_q[b], 0), nq) : nq;
    const float* TN = tnorm + (size_t)vt * nt;
    const float* QN = qnorm + (size_t)vq * nq;
    // eight independent loads in flight per thread and array (one load per trip made the kernel a chain of 32 memory round trips)
    float m = 0.0f, mq = 0.0f;
    for (int j0 = 0; j0 < ntv; j0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + 256 * u + tid; v[u] = (j < ntv) ? TN[j] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, v[u]);
    }
    for (int j0 = 0; j0 < nqv; j0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int j = j0 + 256 * u + tid; v[u] = (j < nqv) ? QN[j] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 8; ++u) mq = fmaxf(mq, v[u]);
    }
    for (int o = 32; o > 0; o >>= 1) { m = fmaxf(m, __shfl_xor(m, o)); mq = fmaxf(mq, __shfl_xor(mq, o)); }
    if ((tid & 63) == 0) { red[0][tid >> 6] = m; red[1][tid >> 6] = mq; }
    if (zero_cnt)
        for (size_t i = tid; i < (size_t)nq * slots; i += 256) cnt[(size_t)b * nq * slots + i] = 0;
    __syncthreads();
    if (tid < 2) tqmax[tid * B + b] = sqrtf(fmaxf(fmaxf(red[tid][0], red[tid][1]), fmaxf(red[tid][2], red[tid][3])));
}

// Finite? — on the bits: this file is built with -fno-honor-nans, under which a floating-point comparison may be compiled
// into its negated opposite and say anything about a NaN.
__device__ __forceinline__ bool finite_bits(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

// ---- norms computed inside the call (norm_q == norm_t == NULL): one pass over each table -------------------------------------
// The row norms are only needed by this call's own kernels, and both tables are read by them anyway: the train table by
// the kernel that writes the tile images, the query table by the filter.  What stands in the way is the scale — it
// follows from the LARGEST row norm of a view, which is only known after the pass.  So the pass uses a PROVISIONAL power
// of two, taken from an estimate of the norm of row 0 of the view, and checks afterwards: it stands if the largest
// norm, scaled, lies in [2^13, 60000] (fp16's largest number is 65504; the band of the header needs 2^12.5 at the low
// end, see band2_scaled's callers) — for descriptors normalised to any common length, always.  Otherwise the train
// view's image is written again with the scale of its true maximum (l2_presplit_kernel<true>, which returns at once for
// every pair whose provisional scale stood), and a query block scales its rows by its OWN maximum and multiplies the
// image's presets, which carry the provisional query scale, by the power of two between the two (sixteen extra
// multiplications per MFMA tile in such blocks only).  Scores of different queries are never compared with each other,
// so the query scale may differ from block to block; the re-rank reads it per block.
// Provisional scale of a row: the same value in every lane of the (full) wave that calls it.
__device__ __forceinline__ float prov_scale(const float* __restrict__ row, int lane) {
    const float2 x = ((const float2*)row)[lane];              // kDim = 128 = 64 lanes x 2 elements
    float sum = ph_fma(x.x, x.x, x.y * x.y);
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    if (!finite_bits(sum) || __float_as_uint(sum) == 0u) return 1.0f;
    int e = 0;
    frexpf(sqrtf(sum), &e);
    return abs(15 - e) > 60 ? 1.0f : ldexpf(1.0f, 15 - e);
}
__device__ __forceinline__ bool scale_stands(float scaled_max) { return scaled_max >= 8192.0f && scaled_max <= 60000.0f; }
struct TrainScale { float st, Tmax; bool bad, fixup; };
__device__ __forceinline__ TrainScale train_scale(uint32_t tmax_sq_bits, float st0) {
    TrainScale r;
    r.Tmax = sqrtf(__uint_as_float(tmax_sq_bits));           // largest FINITE squared row norm of the valid train rows
    const PairScale p = pair_scale(r.Tmax, 1.0f);
    r.fixup = !scale_stands(st0 * r.Tmax);
    r.st = r.fixup ? p.st : st0;
    r.bad = p.bad;
    return r;
}
struct QueryScale { float sq; bool bad; };
__device__ __forceinline__ QueryScale query_block_scale(float qmax_sq, float sq0) {
    const float qm = sqrtf(qmax_sq);
    QueryScale r = {sq0, false};
    if (scale_stands(sq0 * qm) || __float_as_uint(qmax_sq) == 0u) return r;
    const PairScale p = pair_scale(1.0f, qm);
    r.sq = p.sq; r.bad = p.bad;
    return r;
}
constexpr int kQB = 4 * 32 * kQT;   // queries per filter block (kWaves * 32 * kQT)

// Train tile image, written once per (pair, staged tile) and streamed by every query block of the pair straight into
// LDS (global_load_lds: no registers, no VALU, no LDS write instructions in the filter loop).  It is the LDS image:
//   uint4 [0, kSub*512)          scaled fp16 rows, MFMA tile after MFMA tile; chunk p of row r at r*16 + (p ^ (r & 15))
//   then kSub*32 floats          sq*st*(-|t|^2/2) of the 32 rows of every MFMA tile (read ONCE per MFMA tile into 16
//                                registers that are the C operand of the first MFMA of every column tile's chain: the
//                                LDS pipe is the busiest unit of this kernel — at the MFMA peak a CU has time for one
//                                ds_read_b128 per MFMA, and a copy of the presets per column tile was 0.5 of that);
//                                -inf for rows at or beyond the pair's valid count
// A train row whose squared norm is not finite (a NaN or an overflowing element: outside the contract, posehot.h) is written
// as a padded row — zeros and a -inf preset: it can never be listed, whatever the compiler makes of comparisons with NaN.
constexpr int kImgRows = kSub * kTR * kCPR;               // uint4
constexpr int kImgTail = kSub * kTR / 4;                  // uint4
constexpr int kImg = kImgRows + kImgTail;                 // uint4 per staged tile image
// FN = false: norms given by the caller ([V][nt], indexed by view), scales from tqmax.
// FN = true : the fix-up pass of the fused-norms path: norms from the workspace ([B][nt], indexed by pair), and only the
//             pairs whose provisional train scale did not stand are written again.
template <bool FN>
__global__ __launch_bounds__(256) void l2_presplit_kernel(const float* __restrict__ dt, const float* __restrict__ tnorm,
                                                         const int32_t* __restrict__ view_t, const int32_t* __restrict__ n_t,
                                                         int nt, int tiles, int B, const float* __restrict__ tqmax,
                                                         const uint32_t* __restrict__ tmax_bits, const float* __restrict__ scale0,
                                                         uint4* __restrict__ img) {
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles, tid = threadIdx.x;
    const int vt = view_t ? view_t[b] : b;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    float st, sboth;
    if (FN) {
        const TrainScale ts = train_scale(tmax_bits[b], scale0[2 * b]);
        if (!ts.fixup) return;
        st = ts.st; sboth = ts.st * scale0[2 * b + 1];
    } else {
        const PairScale sc = pair_scale(tqmax[b], tqmax[B + b]);
        st = sc.st; sboth = sc.s;
    }
    const float* TN = tnorm + (size_t)(FN ? b : vt) * nt;
    const float4* T = (const float4*)(dt + (size_t)vt * nt * kDim);
    uint4* out = img + (size_t)blockIdx.x * kImg;
    const int t0 = tile * kSR;
#pragma unroll
    for (int u = 0; u < kImgRows / 256; ++u) {
        const int id = tid + 256 * u, row = id / kCPR, p = id % kCPR;      // row within the staged tile
        const bool ok = t0 + row < ntv && finite_bits(TN[min(t0 + row, nt - 1)]);
        const float4* src = T + (size_t)(t0 + row) * (kDim / 4) + 2 * p;
        const float4 a = ok ? src[0] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 c = ok ? src[1] : make_float4(0.f, 0.f, 0.f, 0.f);
        out[row * kCPR + (p ^ (row & 15))] = to_f16x8(a, c, st);
    }
    if (tid < kSub * kTR) {
        const int sub = tid / kTR, row = sub * kTR + (tid & (kTR - 1));
        const float tn = TN[min(t0 + row, nt - 1)];
        ((float*)(out + kImgRows))[tid] = (t0 + row < ntv && finite_bits(tn)) ? -0.5f * tn * sboth : -INFINITY;
    }
}

// Fused-norms path, train side: ONE pass over the train view writes its tile images (provisional scale), its squared row
// norms (the contract's natural-order fma chain: bit for bit ph_row_sqnorm's) and the maximum of the finite ones (atomic
// max on the bits: squared norms are >= +0).  A WAVE owns a staged tile of 64 rows and works through it in four slabs of
// sixteen whole rows: coalesced loads (32 lanes per 512-byte row), every lane converts the float4 it has just loaded and
// stores its 8 bytes of the image (32 lanes = one 256-byte image row: whole cache lines), the slab goes through an 8 KB
// LDS block of the wave's own, and lanes 0-15 sum one row each in natural order.  No block barrier.
// (Tried first: one block per tile with a 32 KB LDS copy and one lane per row behind a barrier, 112 us per 64 x 8192
// rows; the chain as a systolic pass through the sixteen lanes that hold a row's chunks, 257 us; slabs of 64 rows x 128
// bytes as in ph_row_sqnorm, every lane busy in the chain but 64-byte pieces of image rows per store, 102 us.)
__global__ __launch_bounds__(256) void l2_presplit_norm_kernel(const float* __restrict__ dt, const float* __restrict__ dq,
                                                              const int32_t* __restrict__ view_t, const int32_t* __restrict__ view_q,
                                                              const int32_t* __restrict__ n_t, int nt, int nq, int tiles, int n_wtiles,
                                                              float* __restrict__ tnorm_ws, uint32_t* __restrict__ tmax_bits,
                                                              float* __restrict__ scale0, uint4* __restrict__ img) {
    static_assert(kSR == 64 && kDim == 128 && kCPR == 16, "a wave per staged tile, slabs of sixteen rows");
    __shared__ float4 blk[4][16 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = blockIdx.x * 4 + wave;                    // (pair, staged tile) of this wave
    if (wt >= n_wtiles) return;                              // wave-uniform; no block barrier below
    const int b = wt / tiles, tile = wt % tiles;
    const int vt = view_t ? view_t[b] : b, vq = view_q ? view_q[b] : b;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const float4* T = (const float4*)(dt + (size_t)vt * nt * kDim);
    uint4* out4 = img + (size_t)wt * kImg;
    uint2* out = (uint2*)out4;
    const int t0 = tile * kSR;
    float4* my = blk[wave];
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");       // the LDS block is this wave's own; its LDS operations complete in order
        __builtin_amdgcn_wave_barrier();
    };
    const int ci = lane & 31, rh = lane >> 5;                // float4 of the row, which of the two rows of a load
    float4 v[8];
    auto fetch = [&](int slab) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 16 * slab + 2 * i + rh;
            v[i] = (t0 + r < ntv) ? T[(size_t)(t0 + r) * 32 + ci] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    fetch(0);
    // (after the first slab's loads were issued: row 0 of either view is one more memory round trip)
    const float st0 = prov_scale(dt + (size_t)vt * nt * kDim, lane);
    const float sq0 = prov_scale(dq + (size_t)vq * nq * kDim, lane);
    if (tile == 0 && lane == 0) { scale0[2 * b] = st0; scale0[2 * b + 1] = sq0; }
    uint32_t mx = 0u;
#pragma unroll
    for (int slab = 0; slab < 4; ++slab) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int rl = 2 * i + rh, r = 16 * slab + rl;
            my[rl * 32 + (ci ^ rl)] = v[i];
            // this lane's half of image chunk ci / 2 of row r: four fp16
            const _Float16 h0 = (_Float16)(v[i].x * st0), h1 = (_Float16)(v[i].y * st0), h2 = (_Float16)(v[i].z * st0), h3 = (_Float16)(v[i].w * st0);
            uint2 w;
            w.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
            w.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
            out[(r * kCPR + ((ci >> 1) ^ (r & 15))) * 2 + (ci & 1)] = w;
        }
        if (slab + 1 < 4) fetch(slab + 1);
        wave_sync();
        if (lane < 16) {                                     // one row per lane, natural order
            float acc = 0.0f;
#pragma unroll 16
            for (int k = 0; k < 32; ++k) {
                const float4 x = my[lane * 32 + (k ^ lane)];
                acc = ph_fma(x.x, x.x, acc); acc = ph_fma(x.y, x.y, acc);
                acc = ph_fma(x.z, x.z, acc); acc = ph_fma(x.w, x.w, acc);
            }
            const int r = 16 * slab + lane;
            const bool valid = t0 + r < ntv, ok = valid && finite_bits(acc);
            ((float*)(out4 + kImgRows))[r] = ok ? -0.5f * acc * (sq0 * st0) : -INFINITY;
            if (valid) tnorm_ws[(size_t)b * nt + t0 + r] = acc;
            if (ok) mx = max(mx, __float_as_uint(acc));
            if (valid && !ok) {                              // outside the contract, rare: the row becomes a padded row
                __atomic_thread_fence(__ATOMIC_RELEASE);     // behind the image stores of this wave's other lanes
#pragma unroll
                for (int q = 0; q < kCPR; ++q) out4[r * kCPR + q] = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        wave_sync();
    }
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, o));
    if (lane == 0 && mx != 0u) atomicMax(tmax_bits + b, mx);           // one atomic per tile
}
// the "no neighbour" outputs of a call without train rows (fused-norms path: nothing else to do)
__global__ __launch_bounds__(256) void l2_none_kernel(size_t n, int32_t* __restrict__ idx, float* __restrict__ d1sq,
                                                     float* __restrict__ d2sq, int32_t* __restrict__ keep) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    idx[i] = PH_NO_INDEX;
    if (d1sq) d1sq[i] = INFINITY;
    if (d2sq) d2sq[i] = INFINITY;
    if (keep) keep[i] = 0;
}

// One LDS-DMA instruction: every active lane copies 16 bytes from its own global address to LDS byte address
// lds_dst + 16 * lane (lds_dst wave-uniform, passed in M0).  Written as asm on purpose: the compiler drains a DMA it
// knows about (vmcnt(0)) in front of the next LDS read of any address, which would serialise copy and MFMA chain; the
// completion is waited for by hand at the one place that needs it.  An asm load is absent from the compiler's own
// vmcnt bookkeeping, which only makes its waits for its own (older or younger) loads more conservative.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// the same with a scalar base address and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// FN = true: the fused-norms path — the query rows' squared norms are computed HERE (natural-order chain, handed between
// the two lane halves that hold a row's odd and even 8-element chunks) and written to qnorm_ws [B][nq]; the scales are the
// provisional ones of l2_presplit_norm_kernel unless they did not stand (see above); the block's query scale goes to
// qscale [B][nqb] for the re-rank.
template <int WAVES, bool FN>
__global__ __launch_bounds__(WAVES * 64, 3) void l2_filter_kernel(
    const float* __restrict__ dq, const uint4* __restrict__ img, int tiles,
    const int32_t* __restrict__ view_q, const int32_t* __restrict__ view_t,
    const int32_t* __restrict__ n_q, const int32_t* __restrict__ n_t,
    int nq, int nt, int rows_per_split,
    const float* __restrict__ qnorm, const float* __restrict__ tqmax,
    int32_t* __restrict__ cnt, int2* __restrict__ list, int slots, int nqb, int B,
    float* __restrict__ qnorm_ws, const uint32_t* __restrict__ tmax_bits, const float* __restrict__ scale0,
    float* __restrict__ qscale) {
    constexpr int NT = WAVES * 64;
    static_assert(NT / 64 * 32 * kQT == kQB, "queries per block");
    static_assert(WAVES == 4 && kImgRows % 256 == 0 && kImgTail % 4 == 0 && kImgTail <= 256 && kSub * kTR <= 256, "staging deal");
    __shared__ uint4 smem[(kDepth + 1) * kImg];   // ring of kDepth + 1 staged tiles; ONE array: a second LDS object beside an LDS-DMA
                                                  // target makes the compiler drain the DMA before every LDS read

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    // Block -> (pair, split, query block).  Consecutive block ids go round-robin over the 8 XCDs, each with its own L2:
    // all blocks of a pair are given ids of one residue mod 8, so a pair's train view is streamed through ONE L2.
    const int per_pair = nqb * (slots / 2);
    int b, rr;
    if (blockIdx.x < (unsigned)((B & ~7) * per_pair)) {          // whole groups of 8 pairs
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        b = xcd + 8 * (i / per_pair); rr = i % per_pair;
    } else {
        b = blockIdx.x / per_pair; rr = blockIdx.x % per_pair;
    }
    const int split = rr / nqb, xb = rr % nqb;
    const int vq = view_q ? view_q[b] : b;
    const int nqv = n_q ? min(max(n_q[b], 0), nq) : nq;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const float* Q = dq + (size_t)vq * nq * kDim;
    float Tm, Qm, sq_use, s_use;                     // largest norms (Qm: what the band's sub-range term needs), query scale, sq * st
    float sq0 = 1.0f, st_use = 1.0f;
    if (FN) {
        const TrainScale ts = train_scale(tmax_bits[b], scale0[2 * b]);
        sq0 = scale0[2 * b + 1];
        Tm = ts.Tmax; st_use = ts.st; sq_use = sq0; Qm = 0.0f; s_use = 0.0f;       // Qm, s_use: once the block's scale is settled
    } else {
        Tm = tqmax[b]; Qm = tqmax[B + b];
        const PairScale sc = pair_scale(Tm, Qm);
        sq_use = sc.sq; st_use = sc.st; s_use = sc.s;
    }
    const uint4* IMG = img + (size_t)b * tiles * kImg;
    // the pair's candidate lists as a raw buffer (at most 4 GB per pair: nq * slots * 256 bytes)
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
        list + (size_t)b * nq * slots * kCapH, 0, (int)((size_t)nq * slots * kCapH * sizeof(int2)), 0x00020000);

    const int qw0 = xb * (NT / 64) * (32 * kQT) + wave * (32 * kQT);   // first query of this wave

    // B fragments: query row (qw0 + ct*32 + li), elements 16s + 8h .. +7, scaled, as fp16
    uint4 bh[kQT][kNS];
    float thr[kQT], eps2[kQT], a1[kQT], a2[kQT], qn2[kQT];
    size_t qoff[kQT];
    unsigned lpos[kQT], lend[kQT];                   // byte offsets in the pair's lists: where this lane's private segment takes its
                                                     // next entry (it keeps counting beyond the end: overflow), and the segment's end
    auto convert = [&](int ct, float scale) {
        const float4* qrow = (const float4*)(Q + (size_t)min(qw0 + ct * 32 + li, nq - 1) * kDim);
#pragma unroll
        for (int s = 0; s < kNS; ++s) bh[ct][s] = to_f16x8(qrow[4 * s + 2 * h], qrow[4 * s + 2 * h + 1], scale);
    };
    float resc = 1.0f;                               // FN: (this block's query scale) / (the provisional one in the image's presets)
    bool blk_bad = false;                            // FN: the block's query norms are beyond the scaled arithmetic: exact path
    if (FN) {
        float blk_max = 0.0f;
#pragma unroll
        for (int ct = 0; ct < kQT; ++ct) {
            const int qi = qw0 + ct * 32 + li;
            const float4* qrow = (const float4*)(Q + (size_t)min(qi, nq - 1) * kDim);
            // |q|^2 in natural element order: chunk 2s sits in the lower lane half, chunk 2s + 1 in the upper one
            float acc = 0.0f;
#pragma unroll
            for (int s = 0; s < kNS; ++s) {
                const float4 x = qrow[4 * s + 2 * h], y = qrow[4 * s + 2 * h + 1];
                bh[ct][s] = to_f16x8(x, y, sq0);
                float lo, hi;
#pragma unroll
                for (int half = 0; half < 2; ++half) {           // half 0: the lower lanes' chain step is the valid one, then the upper lanes'
                    float r = acc;
                    r = ph_fma(x.x, x.x, r); r = ph_fma(x.y, x.y, r); r = ph_fma(x.z, x.z, r); r = ph_fma(x.w, x.w, r);
                    r = ph_fma(y.x, y.x, r); r = ph_fma(y.y, y.y, r); r = ph_fma(y.z, y.z, r); r = ph_fma(y.w, y.w, r);
                    both_halves(r, lo, hi);
                    acc = half == 0 ? lo : hi;
                }
            }
            qn2[ct] = acc;
            if (split == 0 && h == 0 && qi < nq) qnorm_ws[(size_t)b * nq + qi] = acc;
            if (qi < nqv && finite_bits(acc)) blk_max = fmaxf(blk_max, acc);
        }
        for (int o = 32; o > 0; o >>= 1) blk_max = fmaxf(blk_max, __shfl_xor(blk_max, o));
        float* red = (float*)smem;                   // the staging ring is not in use yet
        if (lane == 0) red[wave] = blk_max;
        __syncthreads();
        blk_max = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();                             // everybody has read before the first copy lands
        const QueryScale qs = query_block_scale(blk_max, sq0);
        blk_bad = qs.bad;
        sq_use = qs.sq;
        if (sq_use != sq0) {                         // block-uniform, rare: rows again at the block's own scale
            resc = sq_use / sq0;
#pragma unroll
            for (int ct = 0; ct < kQT; ++ct) convert(ct, sq_use);
        }
        s_use = sq_use * st_use;
        Qm = 16384.0f / sq_use;                      // the band's sub-range term needs a bound of 2^12.5 / sq (header)
        if (split == 0 && tid == 0) qscale[(size_t)b * nqb + xb] = sq_use;
    } else {
        const float* QN = qnorm + (size_t)vq * nq;
#pragma unroll
        for (int ct = 0; ct < kQT; ++ct) {
            convert(ct, sq_use);
            qn2[ct] = QN[min(qw0 + ct * 32 + li, nq - 1)];
        }
    }
#pragma unroll
    for (int ct = 0; ct < kQT; ++ct) {
        const int qi = qw0 + ct * 32 + li;
        const bool live = qi < nqv && !blk_bad;
        qoff[ct] = ((size_t)b * nq + min(qi, nq - 1)) * slots + (2 * split + h);
        lpos[ct] = (unsigned)(((size_t)min(qi, nq - 1) * slots + (2 * split + h)) * kCapH * sizeof(int2));
        lend[ct] = lpos[ct] + (unsigned)(kCapH * sizeof(int2));
        eps2[ct] = band2_scaled(qn2[ct], Tm, Qm, s_use);
        thr[ct] = live ? -3.0e38f : INFINITY;           // finite floor: the -inf score of a padded row never reaches it;
                                                         // a dead lane never has a candidate
        a1[ct] = -INFINITY; a2[ct] = -INFINITY;
    }
    if (blk_bad) {                                   // every live query of the block: "overflowed" -> the re-rank's exact scan
#pragma unroll
        for (int ct = 0; ct < kQT; ++ct)
            if (qw0 + ct * 32 + li < nqv) cnt[qoff[ct]] = kCapH + 1;
        return;
    }
    const uint32_t resc_bits = __builtin_amdgcn_readfirstlane(__float_as_uint(resc));      // compared as an integer: a scalar compare
    const float resc_u = __uint_as_float(resc_bits);

    const int t_begin = split * rows_per_split;          // a multiple of kSR
    const int t_end = min(ntv, t_begin + rows_per_split);
    if (t_begin >= t_end) {                              // a split beyond the pair's valid train rows: empty segments
#pragma unroll
        for (int ct = 0; ct < kQT; ++ct)
            if (qw0 + ct * 32 + li < nqv) cnt[qoff[ct]] = 0;
        return;
    }
    const int n_tiles = (t_end - t_begin + kSR - 1) / kSR;
    const int n_prime = min(kPrimeRows / kSR, max(1, n_tiles / 8));

    // staged tile -> LDS buffer: wave w copies KB slices w, w+4, ... and its share of the tail, asynchronously.  Scalar base
    // + one per-lane offset register that never changes (tid * 16): a copy costs scalar adds and the DMA instruction itself,
    // no vector instruction (vector address arithmetic in front of every copy sat between the barrier and the first MFMA
    // of every wave: 0.15 of the kernel's time, measured by leaving the copies out).
    const unsigned wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned voff = (unsigned)tid * 16u;
    auto stage = [&](int tile, int buf) {
        const char* src = (const char*)(IMG + (size_t)tile * kImg);
        const unsigned dst = lds_addr(smem) + (unsigned)buf * (kImg * 16u) + wave_s * 1024u;
#pragma unroll
        for (int u = 0; u < kImgRows / 256; ++u) glds16s(src + 4096 * u, voff, dst + 4096u * u);
        // tail: lane l of wave w copies chunk w * (kImgTail / 4) + l; the lane offset register carries w * 1024 too much
        if (lane < kImgTail / 4)
            glds16s(src + kImgRows * 16 + (long)wave_s * (kImgTail / 4 * 16) - (long)wave_s * 1024, voff,
                    lds_addr(smem) + (unsigned)buf * (kImg * 16u) + kImgRows * 16u + wave_s * (kImgTail / 4 * 16u));
    };

    // The two MFMA tiles of a staged tile are worked on one after the other (fragments and presets of a tile are read up
    // front — the k-step of one fp16 product is only kQT MFMAs long: too short to hide an LDS read issued one step ahead
    // — then its MFMA chains run back to back, then its scores are filtered): one tile's fragments and accumulators live
    // at a time keep the kernel at <= 168 VGPR, i.e. three blocks per CU (two cost 18 %, alone and with the other kernels
    // of the step overlapping on other streams: profiles/r2/filter_ablation.txt).
    // The other lane half holds the same query (its other rows): the second-largest of the union of the two halves'
    // top-2 is the tighter bound — half of the time the best two rows sit in different halves.  Once per staged tile,
    // unconditionally (some lane of the wave has a hit in nearly every staged tile; thr is the same in both halves; a bound
    // formed during priming is as valid as any), in two pieces of 4-5 vector instructions that are placed BETWEEN the
    // MFMAs of the first sub-tile: a wave's own MFMAs hide about four vector instructions per MFMA pair
    // (profiles/r2/mfma_valu_overlap.txt) and nothing else of the filter is independent of the accumulators.
    float sw[4];
    auto tighten_swap = [&](int ct) {
        both_halves(a1[ct], sw[0], sw[1]); both_halves(a2[ct], sw[2], sw[3]);
        asm volatile("" : "+v"(sw[0]), "+v"(sw[1]), "+v"(sw[2]), "+v"(sw[3]));
    };
    auto tighten_fold = [&](int ct) {
        const float second = vmax(fminf(sw[0], sw[1]), vmax(sw[2], sw[3]));
        thr[ct] = vmax(thr[ct], second - eps2[ct]);
        asm volatile("" : "+v"(thr[ct]));            // computed here, between the MFMAs, not sunk to its first use
    };
    // Per-lane byte offsets of the eight A fragments inside an MFMA tile and of the lane half's presets: computed once;
    // the ring slot and the sub-tile are compile-time constants (the tile loop is unrolled over the ring), so every LDS
    // read is "register + immediate" and no vector instruction stands between the barrier and the first MFMA.
    unsigned aoff[kNS];
#pragma unroll
    for (int s = 0; s < kNS; ++s) aoff[s] = (unsigned)(li * kCPR + ((2 * s + h) ^ (li & 15))) * 16u;
    const unsigned poff = (unsigned)h * 16u;
    auto tile_scores = [&](auto bufc, auto subc, f32x16 (&acc)[kQT]) {
        constexpr unsigned base = decltype(bufc)::value * (kImg * 16u) + decltype(subc)::value * (kTR * kCPR * 16u);
        constexpr unsigned pbase = decltype(bufc)::value * (kImg * 16u) + kImgRows * 16u + decltype(subc)::value * (kTR * 4u);
        const char* sm = (const char*)smem;
        auto frag = [&](int s) { return *(const uint4*)(sm + (aoff[s] + base)); };
        // issue order pinned by hand (scheduling barriers): presets and two fragments up front, then one fragment read
        // per k-step, two steps ahead of the MFMAs that consume it
        uint4 af[kNS];
        af[0] = frag(0);
        f32x16 preset;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 z = *(const float4*)(sm + (poff + (pbase + 32u * g)));     // rows 8g + 4h .. +3
            preset[4 * g + 0] = z.x; preset[4 * g + 1] = z.y; preset[4 * g + 2] = z.z; preset[4 * g + 3] = z.w;
        }
        if (FN && resc_bits != 0x3f800000u) {                  // wave-uniform, rare: the image's presets carry the provisional query scale
#pragma unroll
            for (int r = 0; r < 16; ++r) preset[r] *= resc_u;
        }
        af[1] = frag(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < kNS; ++s) {
            if (s + 2 < kNS) af[s + 2] = frag(s + 2);
#pragma unroll
            for (int ct = 0; ct < kQT; ++ct) acc[ct] = mfma_f16(af[s], bh[ct][s], s == 0 ? preset : acc[ct]);
            if (decltype(subc)::value == 0) {
                static_assert(kQT == 2 && kNS >= 6, "placement of the bound update");
                if (s == 1) tighten_swap(0);
                if (s == 2) tighten_fold(0);
                if (s == 4) tighten_swap(1);
                if (s == 5) tighten_fold(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // The bound is tracked on TILE MAXIMA: (a1, a2) = the two largest per-tile maxima this lane has seen.  Two tiles
    // are two different rows, so a2 is a lower bound of the lane's second-largest score — nearly as tight as the
    // element-wise one (it differs only when a lane's two best rows share a tile), at two instructions per tile,
    // without branches, instead of four per appended row inside a divergent block.
    auto tile_max = [&](const f32x16& acc, float (&m4)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) m4[i] = vmax(vmax(vmax(acc[4 * i], acc[4 * i + 1]), acc[4 * i + 2]), acc[4 * i + 3]);   // v_max3 + v_max
        return vmax(vmax(vmax(m4[0], m4[1]), m4[2]), m4[3]);
    };
    auto track = [&](float m, int ct) {
        a2[ct] = __builtin_amdgcn_fmed3f(a1[ct], a2[ct], m);
        a1[ct] = vmax(a1[ct], m);
    };
    // priming: nothing appended
    auto prime = [&](const f32x16& acc, int ct) {
        float m4[4];
        track(tile_max(acc, m4), ct);
    };
    // filtering: rows whose approximate score reaches the bound are appended.  The 16 accumulator registers are
    // reduced in four groups of four; a group is entered only if some lane of the wave has a hit in it, so a tile
    // with one hit costs 1 + 4 wave-level tests and 4 masked blocks instead of 16.  All tests of a tile use the bound
    // at its entry (an older, lower bound appends a superset).
    // A list entry is (row - 4h, sA): the row without the lane half's offset is wave-uniform (a scalar), and the half
    // is implied by the segment the entry sits in (the re-rank adds it back).
    // The append is a raw buffer store whose offset is moved out of the buffer's range for lanes without a hit (the
    // hardware drops out-of-range lanes): no exec mask, no branch — a block that holds a memory instruction keeps its
    // skip branch, and those branches, not the instructions, were what the slow path cost.
    auto hit = [&](float v, int jrow, int ct) {
        const bool c = v >= thr[ct];
        const unsigned off = (c && lpos[ct] < lend[ct]) ? lpos[ct] : 0xffffffffu;
        const i32x2 d = {jrow, __float_as_int(v)};
        __builtin_amdgcn_raw_buffer_store_b64(d, rsrc, off, 0, 0);
        lpos[ct] += c ? 8u : 0u;
    };
    auto any_ge = [&](float m, int ct) { return __builtin_amdgcn_ballot_w64(m >= thr[ct]) != 0ull; };
    auto filter = [&](const f32x16& acc, int ct, int t0, bool first_visit) {
        float m4[4];
        const float m = tile_max(acc, m4);
        if (first_visit) track(m, ct);               // wave-uniform; a row must not enter (a1, a2) twice
        if (!any_ge(m, ct)) return;
        const int t0s = __builtin_amdgcn_readfirstlane(t0);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            if (!any_ge(m4[g4], ct)) continue;
            // one append per entered group: the lane's largest of the four (a lane with a hit has, as a rule, one)
            const float a = acc[4 * g4], b = acc[4 * g4 + 1], c = acc[4 * g4 + 2], d = acc[4 * g4 + 3], mg = m4[g4];
            int k = (c == mg) ? 2 : 3;               // mg is one of the four
            k = (b == mg) ? 1 : k; k = (a == mg) ? 0 : k;
            hit(mg, t0s + 8 * g4 + k, ct);
            // a second row of the same four reaching the bound (second-largest of the four): the other three one by one
            // second-largest of four = median of (second-largest of three, largest of three, the fourth)
            const float second = __builtin_amdgcn_fmed3f(__builtin_amdgcn_fmed3f(a, b, c), vmax(vmax(a, b), c), d);
            if (any_ge(second, ct)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) hit(r == k ? -INFINITY : acc[4 * g4 + r], t0s + 8 * g4 + r, ct);
            }
        }
    };
    // Staging pipeline: a ring of kDepth + 1 LDS buffers, one barrier per staged tile, copies kDepth tiles ahead.  At
    // the top of iteration `it` every wave has waited for its own share of tile `it` and the barrier makes the whole
    // image visible (and tells that everybody is done reading tile it-1, whose buffer the newest copy now overwrites).
    // The MFMA chains of all sub-tiles run first; then the wave waits for its share of tile it+1 — a counted wait that
    // leaves the copies of the tiles behind it in flight (the counter retires in issue order) — and then appends
    // candidates.
    constexpr int kGlds = kImgRows / 256 + 1;        // LDS-DMA instructions per wave per staged tile
    // waits for this wave's share of the tile `ahead` tiles behind the newest copy: everything older than the last
    // `ahead` tiles' copies has retired — the copies of the wanted tile AND every list store issued before them
    auto wait_ahead = [&](int ahead) {
        static_assert(kDepth <= 4, "one case per possible count below");
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * kGlds) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kGlds) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kGlds) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // Order of the staged tiles: all n_tiles once — the first n_prime of them only tracked (priming: the bound is still
    // useless), the others tracked and filtered — and then the primed ones once more, filtered under what is by then the
    // FINAL bound: a primed row is listed only if it can be one of the exact top-2 (seen again FIRST, as before round 3, the
    // primed top-2 of every lane half re-entered the lists: 2 of ~10 listed rows per query).  (a1, a2) run over
    // the first visits only, so they always describe distinct rows.
    const int n_iter = n_tiles + n_prime;
    auto tile_of = [&](int it) { return t_begin / kSR + (it < n_tiles ? it : it - n_tiles); };
#pragma unroll
    for (int d = 0; d < kDepth; ++d)
        if (d < n_iter) stage(tile_of(d), d);
    wait_ahead(min(kDepth, n_iter) - 1);
    static_assert(kDepth + 1 == 3 && kSub == 2, "the tile loop below is unrolled over a ring of three buffers");
    auto iteration = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value, NXT = (BUF + kDepth) % (kDepth + 1);
        const bool priming = it < n_prime;
        const int t0 = tile_of(it) * kSR;
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (it + kDepth < n_iter) stage(tile_of(it + kDepth), NXT);
        auto work = [&](auto subc) {
            constexpr int sub = decltype(subc)::value;
            f32x16 acc[kQT];
            tile_scores(bufc, subc, acc);
            // tile it+1 landed (waited for before the first list store of this iteration is issued: the counter retires in order)
            if (sub == 0) wait_ahead(min(kDepth, n_iter - 1 - it) - 1 < 0 ? 0 : min(kDepth, n_iter - 1 - it) - 1);
            if (priming) {
#pragma unroll
                for (int ct = 0; ct < kQT; ++ct) prime(acc[ct], ct);
            } else {
#pragma unroll
                for (int ct = 0; ct < kQT; ++ct) filter(acc[ct], ct, t0 + sub * kTR, it < n_tiles);
            }
        };
        work(std::integral_constant<int, 0>{});
        work(std::integral_constant<int, 1>{});
    };
    for (int it = 0; it < n_iter; it += kDepth + 1) {
        iteration(std::integral_constant<int, 0>{}, it);
        if (it + 1 < n_iter) iteration(std::integral_constant<int, 1>{}, it + 1);
        if (it + 2 < n_iter) iteration(std::integral_constant<int, 2>{}, it + 2);
    }
#pragma unroll
    for (int ct = 0; ct < kQT; ++ct)
        if (qw0 + ct * 32 + li < nqv) cnt[qoff[ct]] = kCapH - (int)(lend[ct] - lpos[ct]) / 8;   // signed: lpos may have run past the end
}

// exact contract chain: s = fma chain over k in the order 8c + {0,4,1,5,2,6,3,7}, started at -|t|^2/2
template <int UNROLL = 8>
__device__ __forceinline__ float exact_score(const float4* __restrict__ q, const float4* __restrict__ t, float tn) {
    float acc = -0.5f * tn;
#pragma unroll UNROLL
    for (int c = 0; c < kDim / 8; ++c) {
        const float4 q0 = q[2 * c], q1 = q[2 * c + 1], t0 = t[2 * c], t1 = t[2 * c + 1];
        acc = ph_fma(t0.x, q0.x, acc); acc = ph_fma(t1.x, q1.x, acc);
        acc = ph_fma(t0.y, q0.y, acc); acc = ph_fma(t1.y, q1.y, acc);
        acc = ph_fma(t0.z, q0.z, acc); acc = ph_fma(t1.z, q1.z, acc);
        acc = ph_fma(t0.w, q0.w, acc); acc = ph_fma(t1.w, q1.w, acc);
    }
    return acc;
}

// Re-rank: kRL = 4 lanes per query, sixteen queries per wave (8 lanes: 418 us for the call's three small kernels, 4: 353,
// 2: 420, benchmarks/dbg/rr_time.py).  (1) Every lane reads its list entries of all segments at once and
// the query's final bound (second-largest listed sA - 2e) is formed by shuffles; (2) the surviving entries of all sixteen
// queries (2-3 each) are compacted into a wave queue in LDS; (3) lane i runs the exact fma chain of queued row i, reading
// the train row straight from global memory (a lane walks its 512-byte row front to back, 8 chunks = one cache line in
// flight per batch) and its query's row from a swizzled LDS copy made once per wave; (4) the lanes of a
// query fold the results of the pass they find under their query's number and merge at the end.
// The chain is a dependent sequence of 128 fmas per row whatever the layout, so what counts is how many rows one pass of
// the wave's instructions serves: with the rows staged through LDS (the previous version: 8 rows per pass, 4 KB of LDS per
// wave for them and 4 KB for the query rows) a wave issued three passes for its ~20 rows and a quarter of all issue
// slots of the kernel were chain instructions with 8 of 64 lanes on; now it is one pass for ~40 rows.
constexpr int kRL = 4;
constexpr int kRRUnroll = 4;           // chain unroll of the re-rank pass (8 spills the row loads of a pass)
constexpr int kRQ = 64 / kRL;          // queries per wave
constexpr int kSegGroup = 4;           // segments of a query whose list heads are loaded together (calls with split train rows)
constexpr int kRQueue = 128;           // queue length that triggers a flush (a typical wave queues about 20 rows in all)
struct RerankLds {
    int qj[kRQueue + 64];              // wave queue of candidates: train row ...
    int qq[kRQueue + 64];              // ... and query (0 .. kRQ-1) of the wave
    float rs[64];                      // exact scores of one pass of 64 queued rows
    float4 qrow[kRQ * 32];             // the wave's query rows: query d's chunk c at d*32 + (c ^ d)
};
// FUSE = the ratio test in the same kernel (ph_match_l2_ratio): keep = what ph_match_l2_top2 followed by ph_ratio_filter_l2
// give, bit for bit, idx = their idx where keep is set and PH_NO_INDEX where it is not, but a query whose answer already
// follows from the APPROXIMATE scores skips the exact chains altogether.  With a(1) >= a(2) the two largest listed approximate scores (unscaled) and e the filter's error bound,
// the exact best and runner-up scores lie in [a(1) - e, a(1) + e] and [a(2) - e, a(2) + e]; if a(1) - a(2) > 2e the best row is
// the row of a(1), uniquely; and d = max(0, |q|^2 - 2 m) turns the two intervals into intervals of d1sq and d2sq (widened by
// 2^-20 relative for the float32 roundings of the contract's own fma and product and of this arithmetic).  If these decide
// d1sq < r2 * d2sq one way or the other the query is done (a kept one also needs the separated best row) — 99 of 100
// are: a planted match wins by far, a non-match fails by far; the others take the exact path, so the outputs never differ
// from the two-kernel path.
// FN = the fused-norms path: tnorm / qnorm are the workspace copies ([B][nt], [B][nq], indexed by pair), the train scale
// follows from (tmax_bits, scale0) as in the filter, the query scale is the filter block's (qscale [B][nqb]).
template <bool FUSE, bool FN>
__global__ __launch_bounds__(256) void l2_rerank_kernel(
    const float* __restrict__ dq, const float* __restrict__ dt,
    const int32_t* __restrict__ view_q, const int32_t* __restrict__ view_t,
    const int32_t* __restrict__ n_q, const int32_t* __restrict__ n_t, int B, int nq, int nt,
    const float* __restrict__ tnorm, const float* __restrict__ qnorm,
    const float* __restrict__ tqmax, const int32_t* __restrict__ cnt, const int2* __restrict__ list, int slots,
    int32_t* __restrict__ idx, float* __restrict__ d1sq, float* __restrict__ d2sq, float r2, int32_t* __restrict__ keep_out,
    const uint32_t* __restrict__ tmax_bits, const float* __restrict__ scale0, const float* __restrict__ qscale, int nqb) {
    __shared__ RerankLds lds[4];
    // blocks of one pair share one residue of the block id mod 8 (one XCD, one L2 for the pair's train rows)
    constexpr int QPB = 256 / kRL;
    const int bpp = (nq + QPB - 1) / QPB;                 // blocks per pair
    int b, bq;
    if (blockIdx.x < (unsigned)((B & ~7) * bpp)) {          // whole groups of 8 pairs
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        b = xcd + 8 * (i / bpp); bq = i % bpp;
    } else {
        b = blockIdx.x / bpp; bq = blockIdx.x % bpp;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int qid = lane / kRL, sl = lane % kRL;          // query within the wave, lane within the query
    const int qi = bq * QPB + threadIdx.x / kRL;
    const int vq = view_q ? view_q[b] : b;
    const int vt = view_t ? view_t[b] : b;
    const int nqv = n_q ? min(max(n_q[b], 0), nq) : nq;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const bool live = qi < nqv;
    const size_t o = (size_t)b * nq + min(qi, nq - 1);    // (pair, query) index
    const float4* q = (const float4*)(dq + ((size_t)vq * nq + min(qi, nq - 1)) * kDim);
    const float4* Tt = (const float4*)(dt + (size_t)vt * nt * kDim);
    const float* TN = tnorm + (size_t)(FN ? b : vt) * nt;
    const float* QN = qnorm + (size_t)(FN ? b : vq) * nq;
    RerankLds& L = lds[wave];
    auto wave_sync = [] {                                  // the LDS block belongs to this wave alone; its LDS operations
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // complete in order: only the compiler needs the fence
        __builtin_amdgcn_wave_barrier();
    };

    // (1) list entries of this lane (entries sl, sl + kRL, ... of every segment) and the final bound.  The counts and first
    // cache lines of the first two segments (all there are unless the train rows are split) are loaded in one go and kept
    // in registers for (2).
    bool overflow = false;
    float p1 = -INFINITY, p2 = -INFINITY;
    constexpr int kE = 16 / kRL;                           // entries per lane in the first cache line of a segment
    int cn[2] = {0, 0};
    int2 ent[2][kE];
    int j1 = PH_NO_INDEX;                                  // FUSE: train row of the largest listed approximate score
    auto bump = [&](float v, int jrow) {
        if (FUSE) j1 = (v > p1) ? jrow : j1;
        p2 = __builtin_amdgcn_fmed3f(p1, p2, v); p1 = fmaxf(p1, v);
    };
    if (live) {
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {             // slots = 2 * splits >= 2
            cn[slot] = cnt[o * slots + slot];
#pragma unroll
            for (int k = 0; k < kE; ++k) ent[slot][k] = list[(o * slots + slot) * kCapH + sl + kRL * k];
        }
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {
            overflow |= cn[slot] > kCapH;
#pragma unroll
            for (int k = 0; k < kE; ++k)
                if (sl + kRL * k < cn[slot]) bump(__int_as_float(ent[slot][k].y), ent[slot][k].x + 4 * (slot & 1));
            for (int k = sl + 16; k < cn[slot] && k < kCapH; k += kRL) {  // second cache line: rarely used
                const int2 e2 = list[(o * slots + slot) * kCapH + k];
                bump(__int_as_float(e2.y), e2.x + 4 * (slot & 1));
            }
        }
        // further segments (train rows split over blocks: few pairs per call): four segments' counts and first cache lines in
        // flight together — one at a time, a single pair's 32 segments were 30 dependent round trips (21 of the call's 30 us)
        for (int s0 = 2; s0 < slots; s0 += kSegGroup) {
            int c[kSegGroup];
            int2 e[kSegGroup][kE];
#pragma unroll
            for (int g = 0; g < kSegGroup; ++g) {
                const int slot = min(s0 + g, slots - 1);   // slots is even and kSegGroup = 4: a group may end two short
                c[g] = s0 + g < slots ? cnt[o * slots + slot] : 0;
#pragma unroll
                for (int k = 0; k < kE; ++k) e[g][k] = list[(o * slots + slot) * kCapH + sl + kRL * k];
            }
#pragma unroll
            for (int g = 0; g < kSegGroup; ++g) {
                const int slot = s0 + g;
                overflow |= c[g] > kCapH;
#pragma unroll
                for (int k = 0; k < kE; ++k)
                    if (sl + kRL * k < c[g]) bump(__int_as_float(e[g][k].y), e[g][k].x + 4 * (slot & 1));
                for (int k = sl + 16; k < c[g] && k < kCapH; k += kRL) {
                    const int2 e2 = list[(o * slots + slot) * kCapH + k];
                    bump(__int_as_float(e2.y), e2.x + 4 * (slot & 1));
                }
            }
        }
    }
#pragma unroll
    for (int w = kRL / 2; w > 0; w >>= 1) {
        const float o1 = __shfl_xor(p1, w), o2 = __shfl_xor(p2, w);
        if (FUSE) { const int oj = __shfl_xor(j1, w); j1 = (o1 > p1) ? oj : j1; }
        p2 = fmaxf(fminf(p1, o1), fmaxf(p2, o2));
        p1 = fmaxf(p1, o1);
        overflow |= (bool)__shfl_xor((int)overflow, w);
    }
    // The two rows with the largest approximate scores are in the list (they passed every running bound), so the
    // second-largest LISTED sA is the query's true second-largest, and only rows with sA >= that - 2e can be one of
    // the exact top-2: typically 2-3 of the ~11 listed rows.
    float Tm, Qm;
    PairScale sc;
    if (FN) {
        const TrainScale ts = train_scale(tmax_bits[b], scale0[2 * b]);
        sc.sq = qscale[(size_t)b * nqb + min(qi, nq - 1) / kQB];
        sc.st = ts.st; sc.s = sc.sq * sc.st; sc.bad = ts.bad;
        Tm = ts.Tmax; Qm = 16384.0f / sc.sq;              // as in the filter
    } else {
        Tm = tqmax[b]; Qm = tqmax[B + b];
        sc = pair_scale(Tm, Qm);
    }
    const float qn_i = QN[min(qi, nq - 1)];
    const float band = band2_scaled(qn_i, Tm, Qm, sc.s);   // 2e, scaled
    const float bound = p2 - band;
    overflow |= sc.bad;                                    // wave-uniform per pair: the lists of such a pair are not used
    bool decided = false, decided_keep = false;            // FUSE: the query's idx and keep follow from the approximate scores
    const bool has1 = p1 > -INFINITY, has2 = p2 > -INFINITY;                    // no arithmetic on infinities below
    if (FUSE && live && !overflow && has1) {
        if (!has2) {                                       // one valid train row: no runner-up, the ratio test fails
            decided = true;
        } else {
            const float inv_s = 1.0f / sc.s;               // a power of two
            const float a1u = p1 * inv_s, a2u = p2 * inv_s, e2 = band * inv_s;             // e2 = 2e, unscaled
            const float w1 = e2 + 9.5367431640625e-7f * (fabsf(qn_i) + 2.0f * fabsf(a1u) + e2);   // 2e + 2^-20 (...)
            const float w2 = e2 + 9.5367431640625e-7f * (fabsf(qn_i) + 2.0f * fabsf(a2u) + e2);
            const float c1 = ph_fma(-2.0f, a1u, qn_i), c2 = ph_fma(-2.0f, a2u, qn_i);
            const float d1_hi = fmaxf(0.0f, c1 + w1), d1_lo = fmaxf(0.0f, c1 - w1);
            const float d2_hi = fmaxf(0.0f, c2 + w2), d2_lo = fmaxf(0.0f, c2 - w2);
            const float rhs_lo = r2 * d2_lo * (1.0f - 9.5367431640625e-7f), rhs_hi = r2 * d2_hi * (1.0f + 9.5367431640625e-7f);
            // kept for sure needs the best ROW too (a(1) - a(2) > 2e: the row of a(1) is the exact best, uniquely);
            // dropped for sure needs no row: a rejected query reports no match
            if (d1_hi < rhs_lo && (p1 - p2) > band) { decided = true; decided_keep = true; }
            else if (d1_lo > rhs_hi) decided = true;
        }
    }

    float m1 = -INFINITY, m2 = -INFINITY;
    int32_t bi = PH_NO_INDEX;
    auto keep = [&](float sc, int j) {
        const bool better = (sc > m1) || (sc == m1 && (bi < 0 || j < bi));
        m2 = better ? m1 : fmaxf(m2, sc);
        bi = better ? j : bi;
        m1 = better ? sc : m1;
    };

    // (3) + (4) for the first n queued rows
    auto flush = [&](int n) {
        for (int base = 0; base < n; base += 64) {
            const int k = min(64, n - base);
            if (lane < k) {
                const int j = L.qj[base + lane], qd = L.qq[base + lane];
                const float4* t = Tt + (size_t)j * (kDim / 4);
                const float4* qr = &L.qrow[qd * 32];
                float acc = -0.5f * TN[j];
#pragma unroll kRRUnroll
                for (int c = 0; c < kDim / 8; ++c) {
                    const float4 q0 = qr[(2 * c) ^ qd], q1 = qr[(2 * c + 1) ^ qd], t0 = t[2 * c], t1 = t[2 * c + 1];
                    acc = ph_fma(t0.x, q0.x, acc); acc = ph_fma(t1.x, q1.x, acc);
                    acc = ph_fma(t0.y, q0.y, acc); acc = ph_fma(t1.y, q1.y, acc);
                    acc = ph_fma(t0.z, q0.z, acc); acc = ph_fma(t1.z, q1.z, acc);
                    acc = ph_fma(t0.w, q0.w, acc); acc = ph_fma(t1.w, q1.w, acc);
                }
                L.rs[lane] = acc;
            }
            wave_sync();
            for (int r = sl; r < k; r += kRL)
                if (L.qq[base + r] == qid) keep(L.rs[r], L.qj[base + r]);
            wave_sync();
        }
    };

    // (2) compaction of the surviving entries of the wave's queries into the wave queue
    int n = 0;                                             // wave-uniform queue length
    int max_slots = (live && !overflow && !decided) ? slots : 0;
    for (int w = 32; w > 0; w >>= 1) max_slots = max(max_slots, __shfl_xor(max_slots, w));
    wave_sync();
    // one candidate row per lane (or -1) -> wave queue; the queue is flushed early only when it reaches kRQueue entries
    auto enqueue = [&](int j) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(j >= 0);
        if (m == 0ull) return;
        if (j >= 0) {
            const int pos = n + __builtin_popcountll(m & ((1ull << lane) - 1ull));
            L.qj[pos] = j; L.qq[pos] = qid;
        }
        n += __builtin_popcountll(m);
        if (n >= kRQueue) {                                // keep the queue within its kRQueue + 64 entries (long lists only)
            wave_sync();
            flush(n);
            n = 0;
            wave_sync();
        }
    };
    const bool use = live && !overflow && !decided;
    // the query rows of the queries that need exact scores -> LDS (16 B per lane and instruction, the four lanes of a query
    // side by side): the chains of a pass would otherwise ask the L1 for the same query line once per lane and chunk.
    // (Issued here, after the lists were read: a decided query of the fused kernel never touches its 512-byte row.)
    if (use) {
#pragma unroll
        for (int c = sl; c < 32; c += kRL) L.qrow[qid * 32 + (c ^ qid)] = q[c];
    }
    auto from_memory = [&](int slot, int pass) {
        int j = -1;
        if (use) {
            const int c = min(cnt[o * slots + slot], kCapH);
            if (sl + kRL * pass < c) {
                const int2 e = list[(o * slots + slot) * kCapH + sl + kRL * pass];
                if (__int_as_float(e.y) >= bound) j = e.x + 4 * (slot & 1);     // the segment's lane half: rows +4
            }
        }
        enqueue(j);
    };
    if (max_slots > 0) {
#pragma unroll
        for (int slot = 0; slot < 2; ++slot) {             // the first two segments: out of the registers of (1)
            const int c = use ? min(cn[slot], kCapH) : 0;
#pragma unroll
            for (int k = 0; k < kE; ++k)
                enqueue((sl + kRL * k < c && __int_as_float(ent[slot][k].y) >= bound) ? ent[slot][k].x + 4 * (slot & 1) : -1);
            if (__builtin_amdgcn_ballot_w64(c > 16) != 0ull)
                for (int pass = kE; pass < kCapH / kRL; ++pass) from_memory(slot, pass);
        }
    }
    for (int s0 = 2; s0 < max_slots; s0 += kSegGroup) {    // as in (1): a group's first cache lines together, the same queue order
        int c[kSegGroup];
        int2 e[kSegGroup][kE];
#pragma unroll
        for (int g = 0; g < kSegGroup; ++g) {
            const int slot = min(s0 + g, slots - 1);
            c[g] = (use && s0 + g < slots) ? min(cnt[o * slots + slot], kCapH) : 0;
#pragma unroll
            for (int k = 0; k < kE; ++k) e[g][k] = use ? list[(o * slots + slot) * kCapH + sl + kRL * k] : make_int2(0, 0);
        }
#pragma unroll
        for (int g = 0; g < kSegGroup; ++g) {
            const int slot = s0 + g;
            if (slot >= max_slots) break;                  // wave-uniform
#pragma unroll
            for (int k = 0; k < kE; ++k)
                enqueue((sl + kRL * k < c[g] && __int_as_float(e[g][k].y) >= bound) ? e[g][k].x + 4 * (slot & 1) : -1);
            if (__builtin_amdgcn_ballot_w64(c[g] > 16) != 0ull)
                for (int pass = kE; pass < kCapH / kRL; ++pass) from_memory(slot, pass);
        }
    }
    wave_sync();
    flush(n);

    if (overflow) { m1 = -INFINITY; m2 = -INFINITY; bi = PH_NO_INDEX; }
    if (overflow)                                          // a segment overflowed: exact scan of every row instead
        for (int j = sl; j < ntv; j += kRL) keep(exact_score(q, Tt + (size_t)j * (kDim / 4), TN[j]), j);
    // butterfly over the lanes of the query: every lane holds (best, runner-up) of the rows it folded
#pragma unroll
    for (int w = kRL / 2; w > 0; w >>= 1) {
        const float om1 = __shfl_xor(m1, w), om2 = __shfl_xor(m2, w);
        const int32_t obi = __shfl_xor(bi, w);
        const bool other = (obi >= 0) && ((om1 > m1) || (om1 == m1 && (bi < 0 || obi < bi)));
        const float lost = other ? m1 : om1;
        m2 = fmaxf(lost, fmaxf(m2, om2));
        m1 = other ? om1 : m1;
        bi = other ? obi : bi;
    }
    if (sl == 0 && qi < nq) {
        const float qn = QN[qi];
        const float d1 = (bi >= 0) ? fmaxf(0.0f, ph_fma(-2.0f, m1, qn)) : INFINITY;
        const float d2 = (m2 > -INFINITY) ? fmaxf(0.0f, ph_fma(-2.0f, m2, qn)) : INFINITY;
        if (FUSE) {
            bool k = false;
            int32_t ji = PH_NO_INDEX;
            if (live && decided) { k = decided_keep; ji = j1; }
            else if (live) { k = bi >= 0 && d2 < INFINITY && d1 < r2 * d2; ji = bi; }       // ph_ratio_filter_l2's test
            idx[o] = k ? ji : PH_NO_INDEX;                     // a rejected query has no match
            keep_out[o] = k ? 1 : 0;
        } else {
            if (!live) { idx[o] = PH_NO_INDEX; d1sq[o] = INFINITY; d2sq[o] = INFINITY; return; }
            idx[o] = bi; d1sq[o] = d1; d2sq[o] = d2;
        }
    }
}

constexpr int kWaves = 4;     // 256 queries per block

}  // namespace

L2FilterPlan plan_l2_filter(int B, int nq, int nt) {
    L2FilterPlan p;
    const long blocks = (long)B * ph_ceil_div(nq, kWaves * 32 * kQT);
    int s = (int)((256 * kBlocksPerCU + blocks - 1) / blocks);   // fill every CU once
    const int max_s = ph_ceil_div(nt, 16 * kSR);                // at least 16 staged tiles (1024 rows) per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    p.rows_per_split = ph_ceil_div(ph_ceil_div(nt, s), kSR) * kSR;
    if (p.rows_per_split < kSR) p.rows_per_split = kSR;
    p.splits = ph_ceil_div(nt, p.rows_per_split);
    if (p.splits < 1) p.splits = 1;
    p.slots = 2 * p.splits;
    p.tiles = ph_ceil_div(nt, kSR);
    p.off_cnt = ((size_t)2 * B * sizeof(float) + 255) / 256 * 256;
    p.off_list = p.off_cnt + ((size_t)B * nq * p.slots * sizeof(int32_t) + 255) / 256 * 256;
    p.off_img = p.off_list + ((size_t)B * nq * p.slots * kCapH * sizeof(int2) + 255) / 256 * 256;
    // the fused-norms path's own arrays: squared row norms of the call's views, per pair; largest train norm (bits),
    // provisional scales, per-block query scales
    p.off_tnorm = p.off_img + ((size_t)B * p.tiles * kImg * sizeof(uint4) + 255) / 256 * 256;
    p.off_qnorm = p.off_tnorm + ((size_t)B * nt * sizeof(float) + 255) / 256 * 256;
    p.off_misc = p.off_qnorm + ((size_t)B * nq * sizeof(float) + 255) / 256 * 256;
    p.total = p.off_misc + ((size_t)B * (3 + ph_ceil_div(nq, kQB)) * sizeof(float) + 255) / 256 * 256;
    return p;
}

int launch_l2_filter(const float* desc_q, const float* desc_t, const float* norm_q, const float* norm_t,
                     const int32_t* view_q, const int32_t* view_t, const int32_t* n_q, const int32_t* n_t,
                     int B, int nq, int nt, int32_t* idx, float* d1sq, float* d2sq, void* ws, hipStream_t st,
                     hipEvent_t ev_begin, hipEvent_t ev_end, float ratio_sq, int32_t* keep) {
    const L2FilterPlan p = plan_l2_filter(B, nq, nt > 0 ? nt : 1);
    float* tqmax = (float*)ws;                                   // Tmax[B] | Qmax[B]
    int32_t* cnt = (int32_t*)((char*)ws + p.off_cnt);
    int2* list = (int2*)((char*)ws + p.off_list);
    uint4* img = (uint4*)((char*)ws + p.off_img);
    const int nqb = ph_ceil_div(nq, kQB);
    const dim3 grid((unsigned)((size_t)nqb * p.splits * B));
    const size_t rr_blocks = (size_t)B * ph_ceil_div(nq, 256 / kRL);
    if (!norm_q) {
        // ---- norms computed here: one pass over each table (see l2_presplit_norm_kernel)
        if (nt <= 0) {                                           // no train rows: every query reports "no neighbour"
            const size_t n = (size_t)B * nq;
            hipLaunchKernelGGL(l2_none_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, idx, d1sq, d2sq, keep);
            return ph_launch_status();
        }
        float* tnorm_ws = (float*)((char*)ws + p.off_tnorm);
        float* qnorm_ws = (float*)((char*)ws + p.off_qnorm);
        uint32_t* tmax_bits = (uint32_t*)((char*)ws + p.off_misc);                // [B]
        float* scale0 = (float*)((char*)ws + p.off_misc) + B;                     // [B][2]: provisional train, query scale
        float* qscale = scale0 + 2 * (size_t)B;                                   // [B][nqb]
        const unsigned img_blocks = (unsigned)((size_t)B * p.tiles);
        ph_zero_words(tmax_bits, (size_t)B, st);
        hipLaunchKernelGGL(l2_presplit_norm_kernel, dim3((img_blocks + 3) / 4), dim3(256), 0, st, desc_t, desc_q, view_t, view_q, n_t,
                           nt, nq, p.tiles, (int)img_blocks, tnorm_ws, tmax_bits, scale0, img);
        hipLaunchKernelGGL(l2_presplit_kernel<true>, dim3(img_blocks), dim3(256), 0, st, desc_t, (const float*)tnorm_ws, view_t, n_t,
                           nt, p.tiles, B, (const float*)nullptr, (const uint32_t*)tmax_bits, (const float*)scale0, img);
        if (ev_begin && hipEventRecord(ev_begin, st) != hipSuccess) return PH_ERR_LAUNCH;
        hipLaunchKernelGGL((l2_filter_kernel<kWaves, true>), grid, dim3(kWaves * 64), 0, st, desc_q, img, p.tiles, view_q, view_t, n_q,
                           n_t, nq, nt, p.rows_per_split, (const float*)nullptr, (const float*)nullptr, cnt, list, p.slots, nqb, B,
                           qnorm_ws, (const uint32_t*)tmax_bits, (const float*)scale0, qscale);
        if (ev_end && hipEventRecord(ev_end, st) != hipSuccess) return PH_ERR_LAUNCH;
        if (keep)
            hipLaunchKernelGGL((l2_rerank_kernel<true, true>), dim3((unsigned)rr_blocks), dim3(256), 0, st, desc_q, desc_t, view_q,
                               view_t, n_q, n_t, B, nq, nt, (const float*)tnorm_ws, (const float*)qnorm_ws, (const float*)nullptr, cnt,
                               list, p.slots, idx, d1sq, d2sq, ratio_sq, keep, (const uint32_t*)tmax_bits, (const float*)scale0,
                               (const float*)qscale, nqb);
        else
            hipLaunchKernelGGL((l2_rerank_kernel<false, true>), dim3((unsigned)rr_blocks), dim3(256), 0, st, desc_q, desc_t, view_q,
                               view_t, n_q, n_t, B, nq, nt, (const float*)tnorm_ws, (const float*)qnorm_ws, (const float*)nullptr, cnt,
                               list, p.slots, idx, d1sq, d2sq, 0.0f, (int32_t*)nullptr, (const uint32_t*)tmax_bits,
                               (const float*)scale0, (const float*)qscale, nqb);
        return ph_launch_status();
    }
    hipLaunchKernelGGL(l2_filter_prep_kernel, dim3(B), dim3(256), 0, st, norm_t, norm_q, view_t, view_q, n_t, n_q, nq, nt,
                       p.slots, B, nt > 0 ? 0 : 1, tqmax, cnt);
    if (nt > 0) {
        hipLaunchKernelGGL(l2_presplit_kernel<false>, dim3((unsigned)((size_t)B * p.tiles)), dim3(256), 0, st, desc_t, norm_t, view_t,
                           n_t, nt, p.tiles, B, (const float*)tqmax, (const uint32_t*)nullptr, (const float*)nullptr, img);
        if (ev_begin && hipEventRecord(ev_begin, st) != hipSuccess) return PH_ERR_LAUNCH;
        hipLaunchKernelGGL((l2_filter_kernel<kWaves, false>), grid, dim3(kWaves * 64), 0, st, desc_q, img, p.tiles, view_q, view_t,
                           n_q, n_t, nq, nt, p.rows_per_split, norm_q, (const float*)tqmax, cnt, list, p.slots, nqb, B,
                           (float*)nullptr, (const uint32_t*)nullptr, (const float*)nullptr, (float*)nullptr);
        if (ev_end && hipEventRecord(ev_end, st) != hipSuccess) return PH_ERR_LAUNCH;
    }
    if (keep)
        hipLaunchKernelGGL((l2_rerank_kernel<true, false>), dim3((unsigned)rr_blocks), dim3(256), 0, st, desc_q, desc_t, view_q,
                           view_t, n_q, n_t, B, nq, nt, norm_t, norm_q, (const float*)tqmax, cnt, list, p.slots, idx, d1sq, d2sq,
                           ratio_sq, keep, (const uint32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, nqb);
    else
        hipLaunchKernelGGL((l2_rerank_kernel<false, false>), dim3((unsigned)rr_blocks), dim3(256), 0, st, desc_q, desc_t, view_q,
                           view_t, n_q, n_t, B, nq, nt, norm_t, norm_q, (const float*)tqmax, cnt, list, p.slots, idx, d1sq, d2sq,
                           0.0f, (int32_t*)nullptr, (const uint32_t*)nullptr, (const float*)nullptr, (const float*)nullptr, nqb);
    return ph_launch_status();
}
