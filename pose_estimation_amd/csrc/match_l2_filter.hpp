// Internal interface between match_l2.hip (C-ABI entry points) and match_l2_filter.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

struct L2FilterPlan {
    int splits, rows_per_split, slots;
    int tiles;                               // staged train tiles per pair
    size_t off_cnt, off_list, off_img, total;   // workspace: Tmax[B], Qmax[B] | entry counts [B][nq][slots] | candidate lists [B][nq][slots][32] of
                                             // (row, approximate score), slots = 2 * splits | train tile images [B][tiles] (scaled fp16 rows + score presets)
    size_t off_tnorm, off_qnorm, off_misc;   // norms computed inside the call (norm_q = norm_t = NULL): |t|^2 [B][nt] | |q|^2 [B][nq] | largest train norm,
                                             // provisional scales, per-block query scales
};
L2FilterPlan plan_l2_filter(int B, int nq, int nt);
// norm_q == norm_t == NULL: the call computes the norms itself, in the passes that read the tables anyway
int launch_l2_filter(const float* desc_q, const float* desc_t, const float* norm_q, const float* norm_t,
                     const int32_t* view_q, const int32_t* view_t, const int32_t* n_q, const int32_t* n_t,
                     int B, int nq, int nt, int32_t* idx, float* d1sq, float* d2sq, void* ws, hipStream_t st,
                     hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr,    // optional: bracket l2_filter_kernel alone
                     float ratio_sq = 0.0f, int32_t* keep = nullptr);               // keep != NULL: fused ratio test, d1sq / d2sq unused
