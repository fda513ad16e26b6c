// match_l2.hip — brute-force squared-L2 matcher on float descriptors: exact-f32 MFMA
// contraction (v_mfma_f32_32x32x2_f32) with the top-2 reduction fused on the accumulators.
// SURVEY.md §8 row a-9 (no reference code; contract in include/posehot.h and DESIGN.md).
//
// Orientation.  A operand = 32 TRAIN rows, B operand = 32 QUERY rows, so the 32x32 result has
// the query on the lane (col = lane & 31) and 16 train rows in the lane's registers
// (row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)): the top-2 over train rows is an in-lane
// scan of the accumulator, the two lane halves are merged once at the very end.
//
// No epilogue arithmetic.  The accumulator of train row j starts at -0.5*|t_j|^2, so after the
// k loop it holds  s = q.t - |t|^2/2  and  argmin_j |q - t_j|^2 == argmax_j s  (|q|^2 is a
// per-query constant, added only when the squared distance is written out).
//
// k order.  Lane half h of MFMA step 4c+e multiplies k = 8c + 4h + e, so a lane reads its
// operands as 16-byte chunks and the f32 result is the fma chain over k in the order
// 8c + {0,4,1,5,2,6,3,7}; the oracle restates exactly that chain.
//
// Work cut: grid = (ceil(nq/256), splits, B), 256 threads; each wave owns 64 queries whose B
// fragments stay in registers (128 VGPRs at dim 128); the block's 4 waves share 32-row train
// tiles staged in LDS (XOR-swizzled 16-byte chunks, conflict-free ds_read_b128).
#include <math.h>

#include "common.hpp"
#include "match_l2_filter.hpp"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTR = 32;   // train rows per tile
constexpr int kQT = 2;    // 32-query column tiles per wave

struct L2Partial { float m1, m2; int32_t bi; int32_t pad; };

// squared row norms, natural k order fma chain from 0; one thread per row.
__global__ __launch_bounds__(256) void l2_sqnorm_kernel(const float* __restrict__ table, int64_t n_rows, int dim,
                                                       float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_rows) return;
    const float4* row = (const float4*)(table + (size_t)r * dim);
    float acc = 0.0f;
    for (int c = 0; c < dim / 4; ++c) {
        const float4 x = row[c];
        acc = ph_fma(x.x, x.x, acc); acc = ph_fma(x.y, x.y, acc);
        acc = ph_fma(x.z, x.z, acc); acc = ph_fma(x.w, x.w, acc);
    }
    out[r] = acc;
}

// The same chain with coalesced loads (dim % 32 == 0): a lane walking its own 512-byte row makes every load touch
// 64 different cache lines, so each wave fetches 64 rows cooperatively, 128 bytes of a row per 8 lanes, transposes
// them through a swizzled LDS block of its own, and every lane then sums its own row in natural k order.
__global__ __launch_bounds__(256) void l2_sqnorm_coalesced_kernel(const float* __restrict__ table, int64_t n_rows, int dim,
                                                                 float* __restrict__ out) {
    __shared__ float4 blk[4][64 * 8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + wave) * 64;
    if (r0 >= n_rows) return;
    float4* my = blk[wave];
    const float4* T = (const float4*)table;
    const int cpr = dim / 4;
    float acc = 0.0f;
    for (int c0 = 0; c0 < cpr; c0 += 8) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int r = 8 * i + (lane >> 3), c = lane & 7;
            if (r0 + r < n_rows) my[r * 8 + (c ^ ((r >> 1) & 7))] = T[(size_t)(r0 + r) * cpr + c0 + c];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the LDS block is this wave's own; a wave's LDS
        __builtin_amdgcn_wave_barrier();                         // operations complete in order
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 x = my[lane * 8 + (c ^ ((lane >> 1) & 7))];   // 128-byte rows: (row >> 1) & 7, not row & 7, keeps a 16-lane pass conflict-free
            acc = ph_fma(x.x, x.x, acc); acc = ph_fma(x.y, x.y, acc);
            acc = ph_fma(x.z, x.z, acc); acc = ph_fma(x.w, x.w, acc);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (r0 + lane < n_rows) out[r0 + lane] = acc;
}

__device__ __forceinline__ void write_final(float m1, float m2, int32_t bi, float qn, int32_t* idx, float* d1sq,
                                            float* d2sq, size_t o) {
    idx[o] = bi;
    d1sq[o] = (bi >= 0) ? fmaxf(0.0f, ph_fma(-2.0f, m1, qn)) : INFINITY;
    d2sq[o] = (m2 > -INFINITY) ? fmaxf(0.0f, ph_fma(-2.0f, m2, qn)) : INFINITY;
}

template <int DIM>
__global__ __launch_bounds__(256, 2) void l2_top2_f32_kernel(
    const float* __restrict__ dq, const float* __restrict__ dt,
    const int32_t* __restrict__ view_q, const int32_t* __restrict__ view_t,
    const int32_t* __restrict__ n_q, const int32_t* __restrict__ n_t,
    int nq, int nt, int splits, int rows_per_split,
    const float* __restrict__ tnorm, const float* __restrict__ qnorm,
    int32_t* __restrict__ idx, float* __restrict__ d1sq, float* __restrict__ d2sq,
    L2Partial* __restrict__ partial) {
    constexpr int CPR = DIM / 4;       // 16-byte chunks per row
    constexpr int NC = DIM / 8;        // chunk pairs = groups of 4 MFMA steps
    constexpr int STG = kTR * CPR / 256;  // chunks staged per thread per tile
    __shared__ float4 tile[2][kTR * CPR];
    __shared__ float4 tnl[2][kTR / 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int b = blockIdx.z, split = blockIdx.y;
    const int vq = view_q ? view_q[b] : b;
    const int vt = view_t ? view_t[b] : b;
    const int nqv = n_q ? min(max(n_q[b], 0), nq) : nq;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const float* Q = dq + (size_t)vq * nq * DIM;
    const float4* T = (const float4*)(dt + (size_t)vt * nt * DIM);
    const float* TN = tnorm + (size_t)vt * nt;
    const float* QN = qnorm + (size_t)vq * nq;

    const int qw0 = blockIdx.x * 256 + wave * (32 * kQT);   // first query of this wave

    // B fragments: query row (qw0 + ct*32 + li), chunks 2c+h
    float4 bq[kQT][NC];
#pragma unroll
    for (int ct = 0; ct < kQT; ++ct) {
        const int qi = qw0 + ct * 32 + li;
        const float4* qrow = (const float4*)(Q + (size_t)min(qi, nq - 1) * DIM);
#pragma unroll
        for (int c = 0; c < NC; ++c) bq[ct][c] = qrow[2 * c + h];
    }

    float m1[kQT], m2[kQT];
    int32_t bi[kQT];
#pragma unroll
    for (int ct = 0; ct < kQT; ++ct) { m1[ct] = -INFINITY; m2[ct] = -INFINITY; bi[ct] = PH_NO_INDEX; }

    const int t_begin = split * rows_per_split;
    const int t_end = min(ntv, t_begin + rows_per_split);

    float4 stage[STG];
    float stage_tn = 0.0f;
    auto fetch = [&](int t0) {
#pragma unroll
        for (int u = 0; u < STG; ++u) {
            const int id = tid + 256 * u, row = id / CPR, p = id % CPR;
            stage[u] = (t0 + row < t_end) ? T[(size_t)(t0 + row) * CPR + p] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < kTR) stage_tn = (t0 + tid < t_end) ? -0.5f * TN[t0 + tid] : -INFINITY;
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < STG; ++u) {
            const int id = tid + 256 * u, row = id / CPR, p = id % CPR;
            tile[buf][row * CPR + (p ^ (row & 15))] = stage[u];
        }
        if (tid < kTR) ((float*)tnl[buf])[tid] = stage_tn;
    };

    // top-2 update of one 32x32 result tile: in-lane scan, increasing train row => strict '>' keeps
    // the lowest index
    auto scan = [&](const f32x16& acc, int ct, int t0) {
        const int jb = t0 + 4 * h;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r];
            const int j = jb + (r & 3) + 8 * (r >> 2);
            const bool gt = v > m1[ct];
            m2[ct] = __builtin_amdgcn_fmed3f(m1[ct], m2[ct], v);
            m1[ct] = gt ? v : m1[ct];
            bi[ct] = gt ? j : bi[ct];
        }
    };

    if (t_begin < t_end) {
        fetch(t_begin);
        int cur = 0;
        // Column tiles are processed one after the other (ct-major) and the scan of a finished tile
        // is issued next to the MFMA chain of the following one, so the VALU epilogue overlaps
        // matrix work inside one wave: scan(acc of ct=1, previous row tile) runs under the ct=0
        // chain of the current row tile, scan(ct=0) under the ct=1 chain.
        f32x16 acc1;
        int t_prev = -1;
        for (int t0 = t_begin; t0 < t_end; t0 += kTR) {
            commit(cur);
            __syncthreads();
            if (t0 + kTR < t_end) fetch(t0 + kTR);

            const float4* trow = &tile[cur][li * CPR];
            f32x16 acc0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 z = tnl[cur][2 * g + h];      // rows 8g + 4h .. +3
                acc0[4 * g + 0] = z.x; acc0[4 * g + 1] = z.y; acc0[4 * g + 2] = z.z; acc0[4 * g + 3] = z.w;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 a = trow[(2 * c + h) ^ (li & 15)];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[0][c].x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[0][c].y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[0][c].z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[0][c].w, acc0, 0, 0, 0);
            }
            if (t_prev >= 0) scan(acc1, 1, t_prev);        // independent of the chain above
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 z = tnl[cur][2 * g + h];
                acc1[4 * g + 0] = z.x; acc1[4 * g + 1] = z.y; acc1[4 * g + 2] = z.z; acc1[4 * g + 3] = z.w;
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 a = trow[(2 * c + h) ^ (li & 15)];
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[1][c].x, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[1][c].y, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[1][c].z, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[1][c].w, acc1, 0, 0, 0);
            }
            scan(acc0, 0, t0);
            t_prev = t0;
            cur ^= 1;
        }
        scan(acc1, 1, t_prev);
    }

    // merge the two lane halves (same query, disjoint train rows)
#pragma unroll
    for (int ct = 0; ct < kQT; ++ct) {
        const float om1 = __shfl_xor(m1[ct], 32), om2 = __shfl_xor(m2[ct], 32);
        const int32_t obi = __shfl_xor(bi[ct], 32);
        const bool other = (om1 > m1[ct]) || (om1 == m1[ct] && obi >= 0 && (bi[ct] < 0 || obi < bi[ct]));
        const float lo = other ? m1[ct] : om1;            // the m1 that lost
        const float nm2 = fmaxf(lo, fmaxf(m2[ct], om2));
        const float nm1 = other ? om1 : m1[ct];
        const int32_t nbi = other ? obi : bi[ct];
        const int qi = qw0 + ct * 32 + li;
        if (h == 0 && qi < nq) {
            const size_t o = (size_t)b * nq + qi;
            if (qi >= nqv) {
                if (splits == 1) { idx[o] = PH_NO_INDEX; d1sq[o] = INFINITY; d2sq[o] = INFINITY; }
                else partial[((size_t)b * splits + split) * nq + qi] = L2Partial{-INFINITY, -INFINITY, PH_NO_INDEX, 0};
            } else if (splits == 1) {
                write_final(nm1, nm2, nbi, QN[qi], idx, d1sq, d2sq, o);
            } else {
                partial[((size_t)b * splits + split) * nq + qi] = L2Partial{nm1, nm2, nbi, 0};
            }
        }
    }
}

__global__ __launch_bounds__(256) void l2_merge_kernel(const L2Partial* __restrict__ partial, const float* __restrict__ qnorm,
                                                      const int32_t* __restrict__ view_q,
                                                      int B, int nq, int splits, int32_t* __restrict__ idx,
                                                      float* __restrict__ d1sq, float* __restrict__ d2sq) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (size_t)B * nq) return;
    const size_t b = o / nq, i = o % nq;
    float m1 = -INFINITY, m2 = -INFINITY;
    int32_t bi = PH_NO_INDEX;
    for (int s = 0; s < splits; ++s) {          // splits ascend in train index: strict '>' keeps the lowest
        const L2Partial p = partial[(b * splits + s) * nq + i];
        const bool other = p.m1 > m1;
        const float lo = other ? m1 : p.m1;
        m2 = fmaxf(lo, fmaxf(m2, p.m2));
        m1 = other ? p.m1 : m1;
        bi = other ? p.bi : bi;
    }
    const size_t vq = view_q ? (size_t)view_q[b] : b;
    write_final(m1, m2, bi, qnorm[vq * nq + i], idx, d1sq, d2sq, o);
}

__global__ __launch_bounds__(256) void ratio_l2_kernel(const int32_t* __restrict__ idx, const float* __restrict__ d1sq,
                                                      const float* __restrict__ d2sq, int64_t n, float r2,
                                                      int32_t* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keep[i] = (idx[i] >= 0 && d2sq[i] < INFINITY && d1sq[i] < r2 * d2sq[i]) ? 1 : 0;
}

struct L2Plan { int splits, rows_per_split; size_t total; };
L2Plan plan_l2(int B, int nq, int nt) {
    L2Plan p;
    const long blocks = (long)B * ph_ceil_div(nq, 256);
    const long target = 512;                                   // 2 blocks of 4 waves per CU
    int s = (int)((target + blocks - 1) / blocks);
    const int max_s = ph_ceil_div(nt, 4 * kTR);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 128) s = 128;
    p.rows_per_split = ph_ceil_div(ph_ceil_div(nt, s), kTR) * kTR;
    if (p.rows_per_split < kTR) p.rows_per_split = kTR;
    p.splits = ph_ceil_div(nt, p.rows_per_split);
    if (p.splits < 1) p.splits = 1;
    p.total = p.splits > 1 ? sizeof(L2Partial) * (size_t)B * p.splits * nq : 0;
    return p;
}

}  // namespace

extern "C" int ph_row_sqnorm(const float* table, int64_t n_rows, int dim, float* out, void* stream) {
    if (n_rows < 0 || dim <= 0 || dim % 4) return PH_ERR_INVALID_ARG;
    if (n_rows == 0) return PH_OK;
    if (!table || !out || !ph_aligned(table, 16)) return PH_ERR_INVALID_ARG;
    if (dim % 32 == 0)
        hipLaunchKernelGGL(l2_sqnorm_coalesced_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, table, n_rows, dim, out);
    else
        hipLaunchKernelGGL(l2_sqnorm_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, table,
                           n_rows, dim, out);
    return ph_launch_status();
}

extern "C" size_t ph_match_l2_ws_bytes(int B, int nq, int nt, int dim, int precision) {
    if (B <= 0 || nq <= 0 || nt < 0) return 0;
    if (nt == 0) nt = 1;              // an empty train view still runs the launch that writes the "no neighbour" outputs
    if (precision == PH_PRECISION_F16_RERANK && dim == 128) return plan_l2_filter(B, nq, nt).total;
    return plan_l2(B, nq, nt).total;
}

extern "C" int ph_match_l2_top2_timed(const float* desc_q, const float* desc_t,
                                      const float* norm_q, const float* norm_t,
                                      const int32_t* view_q, const int32_t* view_t,
                                      const int32_t* n_q, const int32_t* n_t,
                                      int B, int nq, int nt, int dim, int precision,
                                      int32_t* idx, float* d1sq, float* d2sq,
                                      void* ws, size_t ws_bytes, void* stream, void* ev_begin, void* ev_end) {
    if (B < 0 || nq < 0 || nt < 0) return PH_ERR_INVALID_ARG;
    if (dim != 128 && dim != 64) return PH_ERR_UNSUPPORTED;
    if (precision != PH_PRECISION_F32 && precision != PH_PRECISION_F16_RERANK) return PH_ERR_UNSUPPORTED;
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || nq == 0) return PH_OK;
    const bool filter_path = precision == PH_PRECISION_F16_RERANK && dim == 128;
    const bool own_norms = !norm_q && !norm_t && filter_path;        // the call computes the norms itself (one pass per table)
    if (!desc_q || !idx || !d1sq || !d2sq || (nt > 0 && !desc_t)) return PH_ERR_INVALID_ARG;
    if (!own_norms && (!norm_q || (nt > 0 && !norm_t))) return PH_ERR_INVALID_ARG;
    if (!ph_aligned(desc_q, 16) || !ph_aligned(desc_t, 16)) return PH_ERR_INVALID_ARG;
    if (filter_path) {                                               // dim 64 takes the f32 kernel: same outputs
        const L2FilterPlan fp = plan_l2_filter(B, nq, nt > 0 ? nt : 1);
        if (!ws || ws_bytes < fp.total || !ph_aligned(ws, 16)) return PH_ERR_WORKSPACE;
        return launch_l2_filter(desc_q, desc_t, own_norms ? nullptr : norm_q, own_norms ? nullptr : norm_t, view_q, view_t, n_q, n_t,
                                B, nq, nt, idx, d1sq, d2sq, ws, (hipStream_t)stream, (hipEvent_t)ev_begin, (hipEvent_t)ev_end);
    }
    const L2Plan p = plan_l2(B, nq, nt > 0 ? nt : 1);
    if (p.splits > 1 && (!ws || ws_bytes < p.total || !ph_aligned(ws, 16))) return PH_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    L2Partial* part = (L2Partial*)ws;

    const dim3 grid(ph_ceil_div(nq, 256), p.splits, B);
    if (ev_begin && hipEventRecord((hipEvent_t)ev_begin, st) != hipSuccess) return PH_ERR_LAUNCH;
    if (dim == 128)
        hipLaunchKernelGGL(l2_top2_f32_kernel<128>, grid, dim3(256), 0, st, desc_q, desc_t, view_q, view_t, n_q, n_t,
                           nq, nt, p.splits, p.rows_per_split, norm_t, norm_q, idx, d1sq, d2sq, part);
    else
        hipLaunchKernelGGL(l2_top2_f32_kernel<64>, grid, dim3(256), 0, st, desc_q, desc_t, view_q, view_t, n_q, n_t,
                           nq, nt, p.splits, p.rows_per_split, norm_t, norm_q, idx, d1sq, d2sq, part);
    if (ev_end && hipEventRecord((hipEvent_t)ev_end, st) != hipSuccess) return PH_ERR_LAUNCH;
    if (p.splits > 1) {
        const size_t n = (size_t)B * nq;
        hipLaunchKernelGGL(l2_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, norm_q, view_q, B,
                           nq, p.splits, idx, d1sq, d2sq);
    }
    return ph_launch_status();
}

extern "C" int ph_match_l2_top2(const float* desc_q, const float* desc_t,
                                const float* norm_q, const float* norm_t,
                                const int32_t* view_q, const int32_t* view_t,
                                const int32_t* n_q, const int32_t* n_t,
                                int B, int nq, int nt, int dim, int precision,
                                int32_t* idx, float* d1sq, float* d2sq,
                                void* ws, size_t ws_bytes, void* stream) {
    return ph_match_l2_top2_timed(desc_q, desc_t, norm_q, norm_t, view_q, view_t, n_q, n_t, B, nq, nt, dim, precision, idx, d1sq,
                                  d2sq, ws, ws_bytes, stream, nullptr, nullptr);
}

extern "C" int ph_match_l2_ratio(const float* desc_q, const float* desc_t,
                                 const float* norm_q, const float* norm_t,
                                 const int32_t* view_q, const int32_t* view_t,
                                 const int32_t* n_q, const int32_t* n_t,
                                 int B, int nq, int nt, int dim, float ratio,
                                 int32_t* idx, int32_t* keep,
                                 void* ws, size_t ws_bytes, void* stream, void* ev_begin, void* ev_end) {
    if (B < 0 || nq < 0 || nt < 0) return PH_ERR_INVALID_ARG;
    if (dim != 128) return PH_ERR_UNSUPPORTED;             // the fused path is the fp16 filter's (dim 128)
    if (B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || nq == 0) return PH_OK;
    const bool own_norms = !norm_q && !norm_t;                       // the call computes the norms itself (one pass per table)
    if (!desc_q || !idx || !keep || (nt > 0 && !desc_t)) return PH_ERR_INVALID_ARG;
    if (!own_norms && (!norm_q || (nt > 0 && !norm_t))) return PH_ERR_INVALID_ARG;
    if (!ph_aligned(desc_q, 16) || !ph_aligned(desc_t, 16)) return PH_ERR_INVALID_ARG;
    const L2FilterPlan fp = plan_l2_filter(B, nq, nt > 0 ? nt : 1);
    if (!ws || ws_bytes < fp.total || !ph_aligned(ws, 16)) return PH_ERR_WORKSPACE;
    return launch_l2_filter(desc_q, desc_t, own_norms ? nullptr : norm_q, own_norms ? nullptr : norm_t, view_q, view_t, n_q, n_t, B, nq, nt, idx, nullptr, nullptr, ws,
                            (hipStream_t)stream, (hipEvent_t)ev_begin, (hipEvent_t)ev_end, ratio * ratio, keep);
}

extern "C" int ph_ratio_filter_l2(const int32_t* idx, const float* d1sq, const float* d2sq, int64_t n, float ratio,
                                  int32_t* keep, void* stream) {
    if (n < 0) return PH_ERR_INVALID_ARG;
    if (n == 0) return PH_OK;
    if (!idx || !d1sq || !d2sq || !keep) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ratio_l2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, d1sq,
                       d2sq, n, ratio * ratio, keep);
    return ph_launch_status();
}
