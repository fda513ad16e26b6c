// match_hamming.hip — brute-force 256-bit Hamming matcher, nearest + second nearest, on the matrix cores.
// SURVEY.md §8 row a-9 (no reference code; contract in include/posehot.h).
//
// With the bits as +-1 (set -> +1, clear -> -1)  q.t = 256 - 2 d(q, t), so the matcher is a GEMM without norm terms: a
// descriptor bit becomes a 4-bit float (e2m1: +1.0 = 0x2, -1.0 = 0xA, block scales 1.0) and
// v_mfma_scale_f32_32x32x64_f8f6f4 accumulates S = q.t from a zero C operand — 256 exact products, exact in f32.
// (The first MFMA version used 0/1 bits and S = 2 q.t - |t| with -|t| preset per train row through the C operand: those
// presets were half of the kernel's LDS reads, and the LDS pipe — one 128-bit fragment read per MFMA at one column tile per
// wave — is what bounds this kernel.)
// For one query, min d <=> max S.  The accumulator has the query on the lane and 16 train rows in the lane's
// registers (same orientation as the L2 matcher), so top-2 is an in-lane scan in increasing row order (ties -> lowest
// index); a tile is scanned only if some lane's tile maximum beats the query's current runner-up.  Rows at or beyond the
// pair's valid count (all-zero nibbles in the image: S = 0) are set to "no score" in the one staged tile they can occur in.
//   grid = (ceil(nq / 128), splits, B), block = 256 threads = 4 waves, one 32-query column tile per wave.
//   hamming_expand_kernel prepares the pair's train view once (LDS tile images), the matcher streams them by LDS-DMA.
// Results travel as keys (distance << 22) + train_index, which order lexicographically by (distance, index).  When the
// train set is split over blockIdx.y (few pairs, large sets) every split writes its two keys to the workspace and
// hamming_merge_kernel merges them.
// Earlier generations of this kernel (8 xor + 8 popcount VALU ops per pair: 16.6 ms at 28 pairs of 32 768^2; int8
// MFMA with in-kernel expansion: 5.84 ms; this one: 4.0 ms) are in the git history.
#include <type_traits>

#include "common.hpp"

namespace {

constexpr int kIdxBits = 22;                      // train rows per view < 2^22
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1;
constexpr uint32_t kNoKey = 0xffffffffu;

__device__ __forceinline__ void top2_insert(uint32_t& m1, uint32_t& m2, uint32_t key) {
    uint32_t med;                                  // m1 <= m2 always: med3 = new second-best
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(med) : "v"(m1), "v"(m2), "v"(key));
    m2 = med;
    m1 = min(m1, key);
}

__device__ __forceinline__ void emit(uint32_t m1, uint32_t m2, int32_t* idx, int32_t* d1, int32_t* d2, size_t o) {
    idx[o] = (m1 == kNoKey) ? PH_NO_INDEX : (int32_t)(m1 & kIdxMask);
    d1[o] = (m1 == kNoKey) ? PH_HAMMING_NONE : (int32_t)(m1 >> kIdxBits);
    d2[o] = (m2 == kNoKey) ? PH_HAMMING_NONE : (int32_t)(m2 >> kIdxBits);
}

constexpr int kMR = 32;                           // train rows per MFMA tile

// The block-scaled fp4 MFMA takes twice the K per instruction at the cycles of the int8 form: half the matrix
// instructions and half the LDS bytes of an int8 formulation.  1.0 is 0x2 and 2.0 is 0x4 in e2m1.  The order of K inside
// a lane does not matter as long as both operands are expanded by the same function (they are).
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float kNoScoreF = -1.0e9f;              // exactly representable, below every real score (>= -256); its own floor
constexpr float kTopFrac = 0.75f;                 // the largest group fraction of a key: key > score + kTopFrac <=> a larger score

// 8 bits -> 8 nibbles, bit i at bit 4i
__device__ __forceinline__ uint32_t spread8(uint32_t x) {
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    return (t | (t << 3)) & 0x11111111u;
}
// 32 bits -> 32 e2m1 values: bit set -> +1.0 (0x2), clear -> -1.0 (0xA)
__device__ __forceinline__ uint4 expand32_fp4(uint32_t w) {
    return make_uint4(0xAAAAAAAAu ^ (spread8(w & 0xFFu) << 3), 0xAAAAAAAAu ^ (spread8((w >> 8) & 0xFFu) << 3),
                      0xAAAAAAAAu ^ (spread8((w >> 16) & 0xFFu) << 3), 0xAAAAAAAAu ^ (spread8(w >> 24) << 3));
}

// ---- fp4 matcher with the train view expanded ONCE per pair (not once per query block: 128 blocks of a 32 768-row
// view used to repeat the 32-bit -> 32-nibble expansion of every tile) into ready-made LDS tile images that the query
// blocks stream with LDS-DMA: no staging registers, VALU or LDS writes in the loop (the scheme of match_l2_filter.hip).
//   image of a staged tile (64 rows): uint4 [sub*256 + r*8 + (c ^ ((r >> 1) & 7))], sub = 0..1, r = 0..31, c = 16-byte chunk
//                                     (a row is 128 bytes = half of the 64 banks: the sixteen lanes of one pass of a ds_read_b128
//                                     read rows r .. r+15, i.e. eight 256-byte bank lines twice — swizzled with r & 7 two of them met
//                                     in every bank quad, SQ_LDS_BANK_CONFLICT = 2 cycles per MFMA; with (r >> 1) & 7 none do);
//                                     rows at or beyond the pair's valid count are all-zero nibbles
constexpr int kHSub = 2;
constexpr int kHSR = kMR * kHSub;                       // rows per staged tile
constexpr int kHImgRows = kHSub * kMR * 8;               // uint4
constexpr int kHImg = kHImgRows;
__global__ __launch_bounds__(256) void hamming_expand_kernel(const uint32_t* __restrict__ dt, const int32_t* __restrict__ view_t,
                                                            const int32_t* __restrict__ n_t, int nt, int tiles,
                                                            uint4* __restrict__ img) {
    const int b = blockIdx.x / tiles, tile = blockIdx.x % tiles, tid = threadIdx.x;
    const int vt = view_t ? view_t[b] : b;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const uint32_t* T = dt + (size_t)vt * nt * 8;
    uint4* out = img + (size_t)blockIdx.x * kHImg;
    const int t0 = tile * kHSR;
#pragma unroll
    for (int u = 0; u < kHSR * 8 / 256; ++u) {
        const int id = tid + 256 * u, row = id >> 3, part = id & 7, sub = row >> 5, r = row & 31;
        const bool ok = t0 + row < ntv;
        const uint32_t wd = ok ? T[(size_t)(t0 + row) * 8 + part] : 0u;
        out[sub * (kMR * 8) + r * 8 + (part ^ ((r >> 1) & 7))] = ok ? expand32_fp4(wd) : make_uint4(0u, 0u, 0u, 0u);
    }
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {      // see match_l2_filter.hip
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// the same copy with a scalar base address and a 32-bit per-lane byte offset: no vector address arithmetic per copy
__device__ __forceinline__ void glds16s(const char* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// x of the same-numbered lane of the lower (lo) and upper (hi) half of the wave, in every lane (one VALU swap instead of
// an LDS-pipe shuffle); callers use lo and hi symmetrically
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void both_halves(float x, float& lo, float& hi) {
    const u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    lo = __uint_as_float(r.x); hi = __uint_as_float(r.y);
}

template <int QT>
__global__ __launch_bounds__(256, QT == 1 ? 4 : 3) void hamming_fp4_dma_kernel(
    const uint4* __restrict__ dq, const uint4* __restrict__ img, int tiles,
    const int32_t* __restrict__ view_q, const int32_t* __restrict__ n_q, const int32_t* __restrict__ n_t,
    int nq, int nt, int splits, int rows_per_split,
    int32_t* __restrict__ idx, int32_t* __restrict__ d1, int32_t* __restrict__ d2,
    uint2* __restrict__ partial, const uint4* __restrict__ dt_raw, const int32_t* __restrict__ view_t, int img_by_view) {
    __shared__ uint4 smem[3 * kHImg];             // ring of three staged tiles (ONE LDS object beside the DMA target)
    static_assert(kHImgRows % 256 == 0, "staging deal");
    constexpr int kGlds = kHImgRows / 256;        // LDS-DMA instructions per wave per staged tile

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, h = lane >> 5;
    // Block -> (pair, split, query block).  Blocks are dispatched in linear id order, round-robin over the 8 XCDs, each with
    // its own L2: all blocks of a pair get ids of one residue mod 8, so a pair's tile images stream through ONE L2
    // (match_l2_filter.hip does the same).
    const unsigned gx = gridDim.x, per_pair = gridDim.x * gridDim.y, nb = gridDim.z;
    const unsigned id = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
    unsigned bb, rr;
    if (id < (nb & ~7u) * per_pair) { const unsigned i = id >> 3; bb = (id & 7u) + 8u * (i / per_pair); rr = i % per_pair; }   // whole groups of 8 pairs
    else { bb = id / per_pair; rr = id % per_pair; }
    const int b = (int)bb, split = (int)(rr / gx), xq = (int)(rr % gx);
    const int vq = view_q ? view_q[b] : b;
    const int nqv = n_q ? min(max(n_q[b], 0), nq) : nq;
    const int ntv = n_t ? min(max(n_t[b], 0), nt) : nt;
    const uint4* Q = dq + (size_t)vq * nq * 2;
    // the pair's tile images: written for this call (by pair), or the caller's resident images of every view (by train view:
    // ph_hamming_expand_views; rows at or beyond the pair's valid count are masked below whatever the image holds)
    const uint4* IMG = img + (size_t)(img_by_view ? (view_t ? view_t[b] : b) : b) * tiles * kHImg;
    const int qw0 = (xq * 4 + wave) * (32 * QT);

    uint4 bq[QT][4];
    int bi[QT];
    float m1[QT], m2[QT], bound[QT];
#pragma unroll
    for (int ct = 0; ct < QT; ++ct) {
        const int qi = qw0 + ct * 32 + li;
        uint4 a = make_uint4(0, 0, 0, 0), c = a;
        if (qi < nqv) { a = Q[(size_t)qi * 2]; c = Q[(size_t)qi * 2 + 1]; }
        const uint32_t w[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int s = 0; s < 4; ++s) bq[ct][s] = expand32_fp4(h ? w[2 * s + 1] : w[2 * s]);
        m1[ct] = kNoScoreF; m2[ct] = kNoScoreF; bound[ct] = kNoScoreF; bi[ct] = PH_NO_INDEX;
    }

    const int t_begin = split * rows_per_split;          // a multiple of kHSR
    const int t_end = min(ntv, t_begin + rows_per_split);

    // Top-2 upkeep on GROUPS of four rows.  A lane's sixteen scores of a scan are four groups of four consecutive rows; when
    // some lane's maximum beats the entry bound the four group maxima (which the entry test has computed anyway) become keys
    // that carry their group: key = max q.t of the group + (3 - g) / 4 — q.t is an even integer, the fraction orders equal
    // maxima of ONE scan by row order and floor() gives the score back.  The lane keeps the best two group keys seen (max +
    // med3 per group, no index bookkeeping; the best remembers its tile); across scans the comparison is on scores alone
    // (strict '>': the earlier tile keeps a tie).  Which row of the best group it is, and whether the group holds the
    // runner-up as well, is settled ONCE per query after the loop from the 4 x 256 bits themselves (resolve below): every
    // row outside the best group scores at most its group's maximum, so the runner-up is the larger of the second-best
    // group maximum and the second-largest score inside the best group.  The hit path of a scan is 26 instructions
    // instead of 64 (keys on all sixteen rows) or ~100 (per-row compare / select).
    auto scan = [&](const f32x16& acc, int ct, int t0) {
        float m4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) m4[g] = fmaxf(fmaxf(acc[4 * g], acc[4 * g + 1]), fmaxf(acc[4 * g + 2], acc[4 * g + 3]));
        if (__builtin_amdgcn_ballot_w64(fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3])) > bound[ct]) == 0ull) return;
        const float ka = m4[0] + kTopFrac, kb = m4[1] + 0.5f, kc = m4[2] + 0.25f, kd = m4[3];
        float k1 = fmaxf(ka, kb), k2 = fminf(ka, kb);
        k2 = __builtin_amdgcn_fmed3f(k1, k2, kc); k1 = fmaxf(k1, kc);
        k2 = __builtin_amdgcn_fmed3f(k1, k2, kd); k1 = fmaxf(k1, kd);
        const bool gt = k1 > floorf(m1[ct]) + kTopFrac;            // a strictly better score
        m2[ct] = fmaxf(fmaxf(fminf(m1[ct], k1), k2), m2[ct]);      // only the score of the runner-up group is ever used
        m1[ct] = gt ? k1 : m1[ct];
        bi[ct] = gt ? t0 : bi[ct];
        // The other lane half scans the other rows of the same query.  A row can only enter the query's top-2 (ties:
        // lowest index for the best, distance only for the runner-up) if its score beats the second-largest score of BOTH
        // halves' best two groups together, all from earlier rows — the entry test, refreshed after every scan.
        float l1, h1, l2, h2;
        both_halves(m1[ct], l1, h1); both_halves(m2[ct], l2, h2);
        bound[ct] = floorf(fmaxf(fminf(l1, h1), fmaxf(l2, h2)));
    };

    if (t_begin < t_end) {
        const int n_iter = (t_end - t_begin + kHSR - 1) / kHSR;
        const int tile0 = t_begin / kHSR;
        // Copies: scalar base + ONE constant per-lane offset (tid * 16), LDS destination in scalar registers; the tile loop is
        // unrolled over the three ring buffers so that every LDS read is register + immediate: no vector instruction stands
        // between the barrier and a wave's first MFMA (match_l2_filter.hip, DESIGN.md Appendix A.2 "ablation study").
        const unsigned wave_s = __builtin_amdgcn_readfirstlane(wave);
        const unsigned voff = (unsigned)tid * 16u;
        auto stage = [&](int it, int buf) {
            const char* src = (const char*)(IMG + (size_t)(tile0 + it) * kHImg);
            const unsigned dst = lds_addr(smem) + (unsigned)buf * (kHImg * 16u) + wave_s * 1024u;
#pragma unroll
            for (int u = 0; u < kHImgRows / 256; ++u) glds16s(src + 4096 * u, voff, dst + 4096u * u);
        };
        unsigned aoff[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) aoff[s] = (unsigned)(li * 8 + ((2 * s + h) ^ ((li >> 1) & 7))) * 16u;
        stage(0, 0);
        if (n_iter > 1) { stage(1, 1); asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kGlds) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        auto iteration = [&](auto bufc, int it) {
            constexpr int BUF = decltype(bufc)::value, NXT = (BUF + 2) % 3;
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const bool more = it + 2 < n_iter;
            if (more) stage(it + 2, NXT);
            const char* sm = (const char*)smem;
            f32x16 acc[kHSub][QT];
#pragma unroll
            for (int sub = 0; sub < kHSub; ++sub) {
                const unsigned base = BUF * (kHImg * 16u) + sub * (kMR * 8 * 16u);
#pragma unroll
                for (int ct = 0; ct < QT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[sub][ct][r] = 0.0f;      // the zero C operand is free; a register C is not (9 %)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint4 a4 = *(const uint4*)(sm + (aoff[s] + base));
                    const i32x8 a = {(int)a4.x, (int)a4.y, (int)a4.z, (int)a4.w, 0, 0, 0, 0};
#pragma unroll
                    for (int ct = 0; ct < QT; ++ct) {
                        const i32x8 bb = {(int)bq[ct][s].x, (int)bq[ct][s].y, (int)bq[ct][s].z, (int)bq[ct][s].w, 0, 0, 0, 0};
                        acc[sub][ct] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, bb, acc[sub][ct], 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                    }
                }
            }
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kGlds) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int t0 = t_begin + it * kHSR;
            if (t0 + kHSR > t_end) {                    // the staged tile the valid rows end in (wave-uniform, once per block)
#pragma unroll
                for (int sub = 0; sub < kHSub; ++sub)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const bool pad = t0 + sub * kMR + 4 * h + (r & 3) + 8 * (r >> 2) >= t_end;
#pragma unroll
                        for (int ct = 0; ct < QT; ++ct) acc[sub][ct][r] = pad ? kNoScoreF : acc[sub][ct][r];
                    }
            }
#pragma unroll
            for (int sub = 0; sub < kHSub; ++sub)
#pragma unroll
                for (int ct = 0; ct < QT; ++ct) scan(acc[sub][ct], ct, t0 + sub * kMR);
        };
        for (int it = 0; it < n_iter; it += 3) {
            iteration(std::integral_constant<int, 0>{}, it);
            if (it + 1 < n_iter) iteration(std::integral_constant<int, 1>{}, it + 1);
            if (it + 2 < n_iter) iteration(std::integral_constant<int, 2>{}, it + 2);
        }
    }

#pragma unroll
    for (int ct = 0; ct < QT; ++ct) {
        // back to (score, row): the fraction of the best key is its group; the group's four rows are scored again from
        // their bits (the lowest row with the best score is the match; the others may hold the runner-up)
        const int qi = qw0 + ct * 32 + li;
        const float s1 = floorf(m1[ct]);
        float s2 = floorf(m2[ct]);
        int row = PH_NO_INDEX;
        if (bi[ct] >= 0) {
            const int grp = 3 - (int)((m1[ct] - s1) * 4.0f);
            const int base = bi[ct] + 4 * h + 8 * grp;
            const int vt = view_t ? view_t[b] : b;
            const uint4* T = dt_raw + ((size_t)vt * nt) * 2;
            const int qc = min(qi, nq - 1);
            const uint4 qa = Q[(size_t)qc * 2], qb = Q[(size_t)qc * 2 + 1];
            float second = kNoScoreF;
#pragma unroll
            for (int r = 3; r >= 0; --r) {
                const int j = base + r;
                if (j >= t_end) continue;                   // padded rows of the last staged tile
                const uint4 ta = T[(size_t)j * 2], tb = T[(size_t)j * 2 + 1];
                const int d = __popc(qa.x ^ ta.x) + __popc(qa.y ^ ta.y) + __popc(qa.z ^ ta.z) + __popc(qa.w ^ ta.w) +
                              __popc(qb.x ^ tb.x) + __popc(qb.y ^ tb.y) + __popc(qb.z ^ tb.z) + __popc(qb.w ^ tb.w);
                const float sr = (float)(256 - 2 * d);
                if (sr == s1) { if (row >= 0) second = s1; row = j; }     // descending r: the lowest such row stays
                else second = fmaxf(second, sr);
            }
            s2 = fmaxf(s2, second);
        }
        const float om1 = __shfl_xor(s1, 32), om2 = __shfl_xor(s2, 32);
        const int obi = __shfl_xor(row, 32);
        const bool other = (om1 > s1) || (om1 == s1 && obi >= 0 && (row < 0 || obi < row));
        const float lost = other ? s1 : om1;
        const float nm2 = fmaxf(lost, fmaxf(s2, om2));
        const float nm1 = other ? om1 : s1;
        const int nbi = other ? obi : row;
        if (h != 0 || qi >= nq) continue;
        uint32_t k1 = kNoKey, k2 = kNoKey;
        if (qi < nqv && nbi >= 0) {
            k1 = ((uint32_t)((256 - (int)nm1) >> 1) << kIdxBits) + (uint32_t)nbi;          // d = (256 - q.t) / 2
            if (nm2 != kNoScoreF) k2 = ((uint32_t)((256 - (int)nm2) >> 1) << kIdxBits) + kIdxMask;
        }
        if (splits == 1) emit(k1, k2, idx, d1, d2, (size_t)b * nq + qi);
        else partial[((size_t)b * splits + split) * nq + qi] = make_uint2(k1, k2);
    }
}


struct HammingMfmaPlan { int qt, splits, rows_per_split, tiles; size_t off_img, total; };
HammingMfmaPlan plan_hamming_mfma(int B, int nq, int nt) {
    HammingMfmaPlan p;
    const long target = 1024;                                  // four 4-wave blocks per CU
    p.qt = 2;      // two column tiles per wave at 3 waves per SIMD (<= 168 VGPR): every fragment read from LDS feeds two
                   // MFMAs — 2.96 ms at 28 pairs of 32768^2 where one tile per wave at 4 waves per SIMD takes 3.17 (the LDS
                   // pipe bounds the kernel: at one tile per wave it would have to deliver a 128-bit read per lane per MFMA);
                   // before the presets were dropped and the staging cleaned up the order was the reverse (4.0 against 4.4)
    while (p.qt > 1 && (long)B * ph_ceil_div(nq, 128 * p.qt) < target) p.qt >>= 1;
    const long blocks = (long)B * ph_ceil_div(nq, 128 * p.qt);
    int s = (int)((target + blocks - 1) / blocks);
    const int max_s = ph_ceil_div(nt, 8 * kMR);                // at least 8 tiles per split
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    p.rows_per_split = ph_ceil_div(ph_ceil_div(nt, s), kHSR) * kHSR;
    if (p.rows_per_split < kHSR) p.rows_per_split = kHSR;
    p.splits = ph_ceil_div(nt, p.rows_per_split);
    if (p.splits < 1) p.splits = 1;
    p.tiles = ph_ceil_div(nt, kHSR);
    p.off_img = p.splits > 1 ? (sizeof(uint2) * (size_t)B * p.splits * nq + 255) / 256 * 256 : 0;
    p.total = p.off_img + sizeof(uint4) * (size_t)B * p.tiles * kHImg;
    return p;
}

__global__ __launch_bounds__(256) void hamming_merge_kernel(const uint2* __restrict__ partial, int B, int nq,
                                                           int splits, int32_t* __restrict__ idx,
                                                           int32_t* __restrict__ d1, int32_t* __restrict__ d2) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= (size_t)B * nq) return;
    const size_t b = o / nq, i = o % nq;
    uint32_t m1 = kNoKey, m2 = kNoKey;
    for (int s = 0; s < splits; ++s) {
        const uint2 p = partial[(b * splits + s) * nq + i];
        top2_insert(m1, m2, p.x);
        top2_insert(m1, m2, p.y);
    }
    emit(m1, m2, idx, d1, d2, o);
}

__global__ __launch_bounds__(256) void ratio_hamming_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ d1,
                                                           const int32_t* __restrict__ d2, int64_t n, float ratio,
                                                           int32_t* __restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keep[i] = (idx[i] >= 0 && d2[i] != PH_HAMMING_NONE && (float)d1[i] < ratio * (float)d2[i]) ? 1 : 0;
}

}  // namespace

extern "C" size_t ph_match_hamming_ws_bytes(int B, int nq, int nt) {
    if (B <= 0 || nq <= 0 || nt <= 0) return 0;
    return plan_hamming_mfma(B, nq, nt).total;
}

namespace {
// images == nullptr: the train views of the call's pairs are expanded into the workspace first (by pair); else `images` holds
// the expanded tiles of EVERY view of desc_t (ph_hamming_expand_views) and nothing is expanded here.
int hamming_launch(const uint8_t* desc_q, const uint8_t* desc_t, const void* images,
                   const int32_t* view_q, const int32_t* view_t, const int32_t* n_q, const int32_t* n_t,
                   int B, int nq, int nt, int bytes, int32_t* idx, int32_t* d1, int32_t* d2,
                   void* ws, size_t ws_bytes, void* stream) {
    if (B < 0 || nq < 0 || nt < 0) return PH_ERR_INVALID_ARG;
    if (bytes != 32) return PH_ERR_UNSUPPORTED;
    if (nt > (int)kIdxMask || B > 65535) return PH_ERR_UNSUPPORTED;
    if (B == 0 || nq == 0) return PH_OK;
    if (!desc_q || !idx || !d1 || !d2 || (nt > 0 && !desc_t)) return PH_ERR_INVALID_ARG;
    if (!ph_aligned(desc_q, 16) || !ph_aligned(desc_t, 16) || !ph_aligned(images, 16)) return PH_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const HammingMfmaPlan p = (nt > 0) ? plan_hamming_mfma(B, nq, nt) : HammingMfmaPlan{1, 1, kHSR, 0, 0, 0};
    const size_t need = images ? p.off_img : p.total;          // resident images: only the split partials live in the workspace
    if (need > 0) {
        if (!ws || ws_bytes < need || !ph_aligned(ws, 16)) return PH_ERR_WORKSPACE;
    }
    const dim3 grid(ph_ceil_div(nq, 128 * p.qt), p.splits, B);
    const uint4* q4 = (const uint4*)desc_q;
    const uint32_t* t1 = (const uint32_t*)desc_t;
    uint2* part = (uint2*)ws;
    if (nt > 0) {
        const uint4* img = (const uint4*)images;
        if (!images) {
            uint4* own = (uint4*)((char*)ws + p.off_img);
            hipLaunchKernelGGL(hamming_expand_kernel, dim3((unsigned)((size_t)B * p.tiles)), dim3(256), 0, st, t1, view_t, n_t, nt,
                               p.tiles, own);
            img = own;
        }
        const int by_view = images ? 1 : 0;
#define PH_LAUNCH(QT) hipLaunchKernelGGL(hamming_fp4_dma_kernel<QT>, grid, dim3(256), 0, st, q4, img, p.tiles, view_q, n_q, n_t, \
                                         nq, nt, p.splits, p.rows_per_split, idx, d1, d2, part, (const uint4*)t1, view_t, by_view)
        if (p.qt == 2) PH_LAUNCH(2); else PH_LAUNCH(1);
#undef PH_LAUNCH
    } else {                                   // no train rows: every query gets "no neighbour"
        hipLaunchKernelGGL(hamming_fp4_dma_kernel<1>, grid, dim3(256), 0, st, q4, (const uint4*)nullptr, 0, view_q, n_q, n_t, nq,
                           nt, p.splits, p.rows_per_split, idx, d1, d2, part, (const uint4*)nullptr, view_t, 0);
    }
    if (p.splits > 1) {
        const size_t n = (size_t)B * nq;
        hipLaunchKernelGGL(hamming_merge_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, part, B, nq,
                           p.splits, idx, d1, d2);
    }
    return ph_launch_status();
}
}  // namespace

extern "C" int ph_match_hamming_top2(const uint8_t* desc_q, const uint8_t* desc_t,
                                     const int32_t* view_q, const int32_t* view_t,
                                     const int32_t* n_q, const int32_t* n_t,
                                     int B, int nq, int nt, int bytes,
                                     int32_t* idx, int32_t* d1, int32_t* d2,
                                     void* ws, size_t ws_bytes, void* stream) {
    return hamming_launch(desc_q, desc_t, nullptr, view_q, view_t, n_q, n_t, B, nq, nt, bytes, idx, d1, d2, ws, ws_bytes, stream);
}

extern "C" size_t ph_hamming_view_images_bytes(int V, int nt) {
    if (V <= 0 || nt <= 0) return 0;
    return sizeof(uint4) * (size_t)V * ph_ceil_div(nt, kHSR) * kHImg;
}

extern "C" int ph_hamming_expand_views(const uint8_t* desc_t, int V, int nt, int bytes, void* images, size_t images_bytes,
                                       void* stream) {
    if (V < 0 || nt < 0) return PH_ERR_INVALID_ARG;
    if (bytes != 32) return PH_ERR_UNSUPPORTED;
    if (V == 0 || nt == 0) return PH_OK;
    if (!desc_t || !images || !ph_aligned(desc_t, 16) || !ph_aligned(images, 16)) return PH_ERR_INVALID_ARG;
    if (images_bytes < ph_hamming_view_images_bytes(V, nt)) return PH_ERR_WORKSPACE;
    const int tiles = ph_ceil_div(nt, kHSR);
    hipLaunchKernelGGL(hamming_expand_kernel, dim3((unsigned)((size_t)V * tiles)), dim3(256), 0, (hipStream_t)stream,
                       (const uint32_t*)desc_t, (const int32_t*)nullptr, (const int32_t*)nullptr, nt, tiles, (uint4*)images);
    return ph_launch_status();
}

extern "C" size_t ph_match_hamming_images_ws_bytes(int B, int nq, int nt) {
    if (B <= 0 || nq <= 0 || nt <= 0) return 0;
    return plan_hamming_mfma(B, nq, nt).off_img;
}

extern "C" int ph_match_hamming_top2_images(const uint8_t* desc_q, const uint8_t* desc_t, const void* images,
                                            const int32_t* view_q, const int32_t* view_t,
                                            const int32_t* n_q, const int32_t* n_t,
                                            int B, int nq, int nt, int bytes,
                                            int32_t* idx, int32_t* d1, int32_t* d2,
                                            void* ws, size_t ws_bytes, void* stream) {
    if (nt > 0 && !images) return PH_ERR_INVALID_ARG;
    return hamming_launch(desc_q, desc_t, images, view_q, view_t, n_q, n_t, B, nq, nt, bytes, idx, d1, d2, ws, ws_bytes, stream);
}

extern "C" int ph_ratio_filter_hamming(const int32_t* idx, const int32_t* d1, const int32_t* d2, int64_t n,
                                       float ratio, int32_t* keep, void* stream) {
    if (n < 0) return PH_ERR_INVALID_ARG;
    if (n == 0) return PH_OK;
    if (!idx || !d1 || !d2 || !keep) return PH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(ratio_hamming_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       idx, d1, d2, n, ratio, keep);
    return ph_launch_status();
}
