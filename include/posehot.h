/*
 * posehot.h — C ABI of libposehot.so, the MI355X (gfx950) camera-pose hot path.
 *
 * Boundary: the reference (alirezafalah/pose_estimation) has no FFI; its only
 * seam is the set of module-level Python functions called from the per-image loops
 * (main.py:187,196; validate_pose.py:360,369).  Each entry point below cites the reference
 * lines it replaces or, for the three north-star subsystems that have no reference code
 * (matching, RANSAC scoring, triangulation), the SURVEY.md §8 row that specifies it.
 *
 * Conventions (all entry points):
 *  - every data pointer is a DEVICE pointer owned by the caller (torch tensors); the library
 *    allocates nothing; scratch comes in through (ws, ws_bytes), sized by the *_ws_bytes query;
 *  - launches are asynchronous on `stream` (a hipStream_t passed as void*, NULL = default);
 *  - return value: PH_OK (0) or a negative PH_ERR_* code; never throws, never aborts — the
 *    reference's "sentinel return, caller skips the item" convention (main.py:91-107);
 *  - batched: B independent items (image pairs / frames) per call; per-item valid counts
 *    (`n_valid`-style arrays, device int32) allow ragged items inside a fixed row stride;
 *  - re-entrant; no global state.
 *
 * The float arithmetic of every decision (which hypothesis wins, which point is an inlier,
 * which neighbour is best) is a fixed sequence of IEEE-754 binary32 fma/mul operations that is
 * written out in DESIGN.md §"Arithmetic contracts"; the CPU oracle (oracle/) restates the same
 * sequence, which is what makes inlier masks and match indices bit-exact across CPU and GPU.
 */
#ifndef POSEHOT_H
#define POSEHOT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PH_VERSION 100 /* 0.1.0 */

#define PH_OK 0
#define PH_ERR_INVALID_ARG (-1) /* null pointer, negative size, misaligned pointer              */
#define PH_ERR_UNSUPPORTED (-2) /* a size/dtype this build has no kernel for                    */
#define PH_ERR_WORKSPACE (-3)   /* ws_bytes smaller than the matching *_ws_bytes() answer       */
#define PH_ERR_LAUNCH (-4)      /* hipLaunch / hipMemsetAsync reported an error                 */
#define PH_ERR_TOO_LARGE (-5)   /* beyond the static scratch of a no-allocation entry point     */

/* Sentinels written for "no such neighbour" (empty / single-row train sets). */
#define PH_NO_INDEX (-1)
#define PH_HAMMING_NONE 0x7fffffff

int ph_version(void);
const char* ph_status_string(int status);

/* ------------------------------------------------------------------------------------------
 * (1) Brute-force descriptor matching + Lowe ratio test.            SURVEY.md §8 row a-9
 *     No reference code (main.py has no matcher); semantics are this library's contract:
 *     nearest and second-nearest train row per query row, ties -> lowest train index.
 *
 * Descriptor tables are [V][n_rows][bytes] (binary) or [V][n_rows][dim] (float); a pair b
 * matches view view_q[b] (queries) against view view_t[b] (train).  view_q/view_t may be NULL,
 * meaning view b of each table.  n_q/n_t (device int32 [B], may be NULL) give the number of
 * valid rows of the pair's query/train view (<= nq/nt, the row strides of the tables).
 * Outputs are [B][nq]; rows >= n_q[b] get idx = PH_NO_INDEX.
 * ------------------------------------------------------------------------------------------ */

/* Hamming, 256-bit (bytes == 32) descriptors.  d1/d2 = best / second-best distance. */
size_t ph_match_hamming_ws_bytes(int B, int nq, int nt);
int ph_match_hamming_top2(const uint8_t* desc_q, const uint8_t* desc_t,
                          const int32_t* view_q, const int32_t* view_t,
                          const int32_t* n_q, const int32_t* n_t,
                          int B, int nq, int nt, int bytes,
                          int32_t* idx, int32_t* d1, int32_t* d2,
                          void* ws, size_t ws_bytes, void* stream);

/* The same matcher over RESIDENT train-side images (views that are matched again and again — all-pairs runs: BASELINE.json
 * configs[2], [3]): ph_match_hamming_top2 first expands the train view of every pair of the call into the form the matrix cores
 * read (128 bytes per row); ph_hamming_expand_views does that ONCE for all V views of a table into `images`
 * (ph_hamming_view_images_bytes(V, nt) bytes: 1.05 GB for 1 000 views of 8 192 rows), and ph_match_hamming_top2_images reads
 * them by view_t[b] instead.  Same outputs, bit for bit; the workspace holds the split partials only
 * (ph_match_hamming_images_ws_bytes, 0 for calls that do not split the train rows).  desc_t is still read (a query's best
 * group of four rows is resolved from the bits).  No reference counterpart. */
size_t ph_hamming_view_images_bytes(int V, int nt);
int ph_hamming_expand_views(const uint8_t* desc_t, int V, int nt, int bytes, void* images, size_t images_bytes,
                            void* stream);
size_t ph_match_hamming_images_ws_bytes(int B, int nq, int nt);
int ph_match_hamming_top2_images(const uint8_t* desc_q, const uint8_t* desc_t, const void* images,
                                 const int32_t* view_q, const int32_t* view_t,
                                 const int32_t* n_q, const int32_t* n_t,
                                 int B, int nq, int nt, int bytes,
                                 int32_t* idx, int32_t* d1, int32_t* d2,
                                 void* ws, size_t ws_bytes, void* stream);

/* Squared row norms |a_r|^2 of a [n_rows][dim] f32 table (natural-order fma chain from 0),
 * computed ONCE per view table and reused by every pair that touches the view. */
int ph_row_sqnorm(const float* table, int64_t n_rows, int dim, float* out, void* stream);

/* Squared-L2 on float descriptors (dim == 128 or 64), fp32 MFMA contraction with a fused
 * top-2 reduction.  norm_q / norm_t are the ph_row_sqnorm outputs of the two tables
 * ([Vq][nq], [Vt][nt]).  precision selects the ALGORITHM, never the result — both give the same
 * bit-exact outputs (the f32 fma-chain contract of DESIGN.md §4):
 *   PH_PRECISION_F32           every score on v_mfma_f32_32x32x2_f32;
 *   PH_PRECISION_F16_RERANK    one fp16 MFMA product per score (rows scaled per pair by powers of two) selects
 *                              ~11 candidate rows per query inside a proven error band, the contract's f32
 *                              chain re-ranks them (dim 128; dim 64 runs the f32 kernel).  Needs workspace.
 * d1sq/d2sq = squared distances of the best / second-best train row (+inf when absent).
 * norm_q = norm_t = NULL (PH_PRECISION_F16_RERANK, dim 128 only; both or neither): the call computes the squared row norms itself, in
 * the passes that read the two tables anyway (one read of each table instead of two; same outputs bit for bit) — for views that are
 * matched once.  Views shared by many pairs should keep their ph_row_sqnorm norms and hand them in.  The workspace size is the same.
 * Descriptors must be finite.  A NaN or infinite element makes the neighbours of the queries and train rows it touches
 * unspecified (they may come back as PH_NO_INDEX, and the two algorithms may then differ): comparisons with NaN are
 * false, so such rows are simply never preferred — no launch fails and nothing is read or written out of bounds. */
#define PH_PRECISION_F32 0
#define PH_PRECISION_F16_RERANK 1
size_t ph_match_l2_ws_bytes(int B, int nq, int nt, int dim, int precision);
int ph_match_l2_top2(const float* desc_q, const float* desc_t,
                     const float* norm_q, const float* norm_t,
                     const int32_t* view_q, const int32_t* view_t,
                     const int32_t* n_q, const int32_t* n_t,
                     int B, int nq, int nt, int dim, int precision,
                     int32_t* idx, float* d1sq, float* d2sq,
                     void* ws, size_t ws_bytes, void* stream);

/* Measurement hook (no reference counterpart; bench.py's roofline object): ph_match_l2_top2 with two caller-created
 * hipEvent_t (as void*, either may be NULL) recorded on `stream` immediately before and after the call's DOMINANT
 * kernel alone — l2_filter_kernel for PH_PRECISION_F16_RERANK, l2_top2_f32_kernel for PH_PRECISION_F32 — so its
 * duration can be read with hipEventElapsedTime while the call runs inside a whole step. */
int ph_match_l2_top2_timed(const float* desc_q, const float* desc_t,
                           const float* norm_q, const float* norm_t,
                           const int32_t* view_q, const int32_t* view_t,
                           const int32_t* n_q, const int32_t* n_t,
                           int B, int nq, int nt, int dim, int precision,
                           int32_t* idx, float* d1sq, float* d2sq,
                           void* ws, size_t ws_bytes, void* stream, void* ev_begin, void* ev_end);

/* ph_match_l2_ratio: matching and ratio test in one call (north_star 1: "descriptor matching + ratio test"; no reference
 * counterpart): the accepted matches.  keep is, bit for bit, what ph_match_l2_top2(precision = PH_PRECISION_F16_RERANK)
 * followed by ph_ratio_filter_l2(ratio) write; idx is their idx where keep = 1 and PH_NO_INDEX where keep = 0 (a rejected
 * query reports no match).  A query whose decision (and, if kept, whose best row) already follows from the candidate
 * filter's approximate scores and their proven error band is never re-scored exactly (a planted match wins by far, a
 * non-match fails by far: 99 queries of 100); the others take the exact path.  The distances are not returned.
 * dim = 128 only; workspace as for ph_match_l2_top2 with that precision; ev_begin / ev_end as in ph_match_l2_top2_timed
 * (NULL = no events). */
int ph_match_l2_ratio(const float* desc_q, const float* desc_t,
                      const float* norm_q, const float* norm_t,
                      const int32_t* view_q, const int32_t* view_t,
                      const int32_t* n_q, const int32_t* n_t,
                      int B, int nq, int nt, int dim, float ratio,
                      int32_t* idx, int32_t* keep,
                      void* ws, size_t ws_bytes, void* stream, void* ev_begin, void* ev_end);

/* Lowe ratio test, strict '<':  keep[i] = idx[i] >= 0 && second exists && d1 < ratio * d2.
 * Hamming form compares d1 < ratio*d2 in f32; L2 form takes SQUARED distances and compares
 * d1sq < ratio^2 * d2sq (== d1 < ratio*d2 on true distances).  n = B*nq elements. */
int ph_ratio_filter_hamming(const int32_t* idx, const int32_t* d1, const int32_t* d2, int64_t n,
                            float ratio, int32_t* keep, void* stream);
int ph_ratio_filter_l2(const int32_t* idx, const float* d1sq, const float* d2sq, int64_t n,
                       float ratio, int32_t* keep, void* stream);

/* Stable compaction of kept matches into SoA correspondences.  kp_q / kp_t are keypoint tables
 * [V][n_rows][2] (x,y) f32 in whatever coordinates the scorer wants (normalised for E).
 * For pair b, every kept query i (in increasing i) appends (kp_q[i], kp_t[idx[i]]) to
 * x1,y1,x2,y2 [B][nq] and match_q / match_t [B][nq] (may be NULL); n_match[b] = number kept. */
int ph_compact_matches(const int32_t* idx, const int32_t* keep,
                       const float* kp_q, const float* kp_t,
                       const int32_t* view_q, const int32_t* view_t,
                       int B, int nq, int nt,
                       float* x1, float* y1, float* x2, float* y2,
                       int32_t* match_q, int32_t* match_t, int32_t* n_match, void* stream);

/* ------------------------------------------------------------------------------------------
 * (2) RANSAC hypothesis scoring.                                    SURVEY.md §8 row a-10
 *     PnP residual = the reference's reprojection error, validate_pose.py:260-288 (project,
 *     per-point Euclidean pixel distance); the essential-matrix side has no reference code.
 *     One launch scores H hypotheses against N correspondences for each of B items and
 *     returns inlier counts; points are SoA rows of stride N; n_valid[b] (NULL = N) points
 *     of item b are valid.
 * ------------------------------------------------------------------------------------------ */

/* Minimal-sample indices, counter-based so CPU and GPU draw identical sample sets (north_star:
 * "with identical RANSAC sample sets"): sample s of item b holds m (<= 8) distinct indices in
 * [0, n_valid[b]); draw t is word 0 of Philox4x32-10(counter = (s, t, 0, 0), key = (seed, b)) mapped by
 * idx = (word * n_valid) >> 32, repeated indices skipped.  samples is [B][S][m]; n_valid < m -> -1.
 * No reference counterpart (the reference has no RANSAC); SURVEY.md §7 "Counter-based RNG". */
int ph_sample_minimal(uint32_t seed, const int32_t* n_valid, int B, int N, int S, int m,
                      int32_t* samples, void* stream);
/* The same with the Philox key's second word taken from item_key[b] (device uint32 [B]; NULL = b) instead of the
 * item's position in the batch: a pair keyed by its GLOBAL pair id draws the same samples whichever rank, chunk or
 * batch slot it lands in, so a sharded run reproduces the unsharded one row for row (the pair loop
 * being sharded is main.py:458-463).  No reference counterpart. */
int ph_sample_minimal_keyed(uint32_t seed, const uint32_t* item_key, const int32_t* n_valid, int B, int N, int S,
                            int m, int32_t* samples, void* stream);

/* Nister 5-point essential-matrix solver (float64 inside).  Two launches: one thread per minimal
 * sample builds the degree-10 polynomial, then one lane per (sample, root) runs the Aberth-Ehrlich
 * root iteration and turns its root into an E.  samples is [B][S][5] (indices into the item's N
 * correspondences, normalised coordinates).  E is [B][S*10][9] f32: sample s owns hypothesis
 * slots 10s .. 10s+9 (slot = root index); a real root gives a unit-Frobenius-norm E, every other
 * slot is all-zero (it scores zero inliers).  n_solutions ([B][S], may be NULL) = real solutions
 * per sample.  ws: ph_solve_essential_5pt_ws_bytes(B, S) bytes.  SURVEY.md §8f item 1. */
size_t ph_solve_essential_5pt_ws_bytes(int B, int S);
int ph_solve_essential_5pt(const float* x1, const float* y1, const float* x2, const float* y2,
                           const int32_t* samples, int B, int S, int N,
                           float* E, int32_t* n_solutions, void* ws, size_t ws_bytes, void* stream);

/* The one entry point that ALLOCATES NOTHING and takes no workspace.  It has no counterpart in the reference; it is
 * for callers that cannot call hipMalloc or bring a workspace: code under HIP graph capture, tight serving loops,
 * harnesses that forbid allocation.  Same arguments, kernels and bit-identical outputs as ph_solve_essential_5pt; the
 * polynomial records that call keeps in (ws, ws_bytes) live here in a file-scope __device__ array of the library,
 * PH_NOALLOC_MAX_SAMPLES records = 5.7 MB, which every GPU the library runs on holds until the process ends (its
 * address is looked up once per device, on the first call made with that device current — make that first call
 * outside graph capture).  B * S > PH_NOALLOC_MAX_SAMPLES returns PH_ERR_TOO_LARGE and launches nothing.  Every call
 * on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race on it — serialise
 * them or use ph_solve_essential_5pt with separate workspaces.  Nothing in the array has to start at zero (every
 * record a call reads, that call has written first), so the contents a previous call left behind do no harm. */
#define PH_NOALLOC_MAX_SAMPLES 8192
int ph_solve_essential_5pt_noalloc(const float* x1, const float* y1, const float* x2, const float* y2,
                                   const int32_t* samples, int B, int S, int N,
                                   float* E, int32_t* n_solutions, void* stream);

/* P3P minimal solver (float64 inside, one thread per sample): absolute pose from three 3-D points and
 * their image points.  samples is [B][S][3] (indices into the item's N correspondences); (u, v) are
 * normalised, undistorted image coordinates (ph_undistort_points).  P is [B][S*4][12] f32, 16-byte
 * aligned: sample s owns hypothesis slots 4s .. 4s+3 (slot = root index of the quartic); a solution is
 * the row-major 3x4 matrix K [R | t] when K4 = (fx, fy, cx, cy) is given — ready for ph_score_pnp on
 * undistorted pixels — or [R | t] when K4 is NULL (normalised coordinates); every other slot is
 * all-zero (w = 0: it scores zero inliers).  n_solutions ([B][S], may be NULL) = solutions per sample.
 * No reference counterpart (main.py:101-104 calls cv2.solvePnP on all points); SURVEY.md §8f item 1. */
int ph_solve_p3p(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                 const int32_t* samples, int B, int S, int N, const double* K4,
                 float* P, int32_t* n_solutions, void* stream);

/* Sampson distance of (x1,y1)->(x2,y2) (normalised image coordinates) under E (row-major 9):
 * inlier iff (x2' E x1)^2 < thr_sq * ((Ex1)_0^2 + (Ex1)_1^2 + (E'x2)_0^2 + (E'x2)_1^2). */
int ph_score_essential(const float* E, const float* x1, const float* y1,
                       const float* x2, const float* y2, const int32_t* n_valid,
                       int B, int H, int N, float thr_sq, int32_t* count, void* stream);

/* Reprojection error of (X,Y,Z)->(u,v) under a 3x4 projection P (row-major 12; K[R|t] for
 * pixel thresholds, [R|t] for normalised ones; (u,v) already undistorted):
 * inlier iff w > 0 and (pu - u*w)^2 + (pv - v*w)^2 < thr_sq * w^2,  (pu,pv,w) = P [X Y Z 1]'. */
int ph_score_pnp(const float* P, const float* X, const float* Y, const float* Z,
                 const float* u, const float* v, const int32_t* n_valid,
                 int B, int H, int N, float thr_sq, int32_t* count, void* stream);

/* The same two scorings for callers that only want the winner (RANSAC): a hypothesis is abandoned as soon as its
 * count so far plus the correspondences not yet seen cannot reach the best count already known for the item.
 * count[b][h] is then EXACT for every hypothesis that attains the item's maximum and a lower bound (below the
 * maximum) for the others, so ph_select_best of it returns exactly what it returns for ph_score_*'s exhaustive
 * counts, ties included.  best_lb is [B][PH_BAILOUT_SLOTS] int32 scratch (zeroed by the call).  When the launch
 * splits the correspondences over blocks (few items, many points) the counts are the exhaustive ones.
 * No reference counterpart (bail-out scoring is the standard RANSAC practice, e.g. Capel 2005). */
#define PH_BAILOUT_SLOTS 2048
int ph_score_essential_bailout(const float* E, const float* x1, const float* y1,
                               const float* x2, const float* y2, const int32_t* n_valid,
                               int B, int H, int N, float thr_sq, int32_t* count, int32_t* best_lb,
                               void* stream);
int ph_score_pnp_bailout(const float* P, const float* X, const float* Y, const float* Z,
                         const float* u, const float* v, const int32_t* n_valid,
                         int B, int H, int N, float thr_sq, int32_t* count, int32_t* best_lb,
                         void* stream);

/* argmax over hypotheses, ties -> lowest hypothesis index.  best_idx/best_count are [B]. */
int ph_select_best(const int32_t* count, int B, int H, int32_t* best_idx, int32_t* best_count,
                   void* stream);
/* ph_select_best over the counts of a COMPACT block (ph_compact_models*), reporting the winner's SLOT: best_idx[b] =
 * table[b][arg-max] (table = the compaction's slot map, [B][H]; -1 for an item without a live model), best_count as
 * ph_select_best.  One launch for ph_select_best + ph_lookup_index.  No reference counterpart (glue). */
int ph_select_best_mapped(const int32_t* count, const int32_t* table, int B, int H, int32_t* best_idx,
                          int32_t* best_count, void* stream);

/* Inlier mask of ONE model per item: model[b] = models + (sel ? sel[b] : 0)*stride floats
 * inside item b's [H][stride] block (stride 9 / 12).  mask is [B][ceil(N/64)] 64-bit words,
 * bit (i & 63) of word (i >> 6) = point i; bits >= n_valid[b] are zero. */
int ph_inlier_mask_essential(const float* E, const int32_t* sel, int H,
                             const float* x1, const float* y1, const float* x2, const float* y2,
                             const int32_t* n_valid, int B, int N, float thr_sq,
                             uint64_t* mask, void* stream);
int ph_inlier_mask_pnp(const float* P, const int32_t* sel, int H,
                       const float* X, const float* Y, const float* Z,
                       const float* u, const float* v,
                       const int32_t* n_valid, int B, int N, float thr_sq,
                       uint64_t* mask, void* stream);

/* ------------------------------------------------------------------------------------------
 * (3) Batched linear (DLT) triangulation.                           SURVEY.md §8 row a-11
 *     Rows x*P[2]-P[0], y*P[2]-P[1] for both views; the solution is the eigenvector of the
 *     smallest eigenvalue of A'A (= smallest right singular vector of A), de-homogenised.
 *     P1/P2 are [B][C][12]: C candidate second-view/first-view projection pairs per item
 *     (C = 1 for plain triangulation, 4 for recoverPose-style cheirality voting).
 *     XYZ is [B][C][N][3] (may be NULL when only the vote is wanted); front[B][C] (may be
 *     NULL) counts points with positive depth in BOTH views among points whose bit is set in
 *     `mask` ([B][ceil(N/64)], NULL = all valid points).
 * ------------------------------------------------------------------------------------------ */
int ph_triangulate(const float* P1, const float* P2,
                   const float* x1, const float* y1, const float* x2, const float* y2,
                   const int32_t* n_valid, const uint64_t* mask,
                   int B, int C, int N, float* XYZ, int32_t* front, void* stream);

/* Glue of the batched two-view pipeline (no reference counterpart; SURVEY.md §8e "pose table"):
 * ph_gather_models copies hypothesis sel[b] (NULL = 0) of every item's [H][stride] block to out [B][stride];
 * ph_pose_table picks the cheirality winner win[b] = argmax_c front[b][c] (first maximum; win may be NULL) among
 * the C candidate projections P2 [B][C][12] = [R | t] and writes the pose-table row the ranks gather:
 * R (9, row-major) | t (3) | inliers | matches | winning hypothesis | status (1 if inliers >= min_inliers). */
int ph_gather_models(const float* models, const int32_t* sel, int B, int H, int stride,
                     float* out, void* stream);
/* ph_compact_models: stable compaction of the non-empty models of every item's [H][stride] block (the solvers leave a slot
 * all-zero when its root is not a solution: about half of the slots).  out [B][H][stride]: the live models first, in slot
 * order, the tail zero-filled (must not alias `models`); slot [B][H]: the slot a compact position came from, -1 in the tail;
 * n_live [B].  Scoring the compact block costs what its live models cost — a wave of eight empty slots leaves at once — and
 * ph_select_best of its counts, mapped back with ph_lookup_index (out[b] = table[b][sel[b]], -1 outside [0, H)), is the
 * winner of the slot layout, ties included (the order is kept) — as long as some model has an inlier.  When the best count
 * is 0 the two layouts differ: the slot layout's argmax is slot 0 (possibly an empty slot), the compact layout's is the first
 * LIVE slot; and an item without any live model (fewer matches than a minimal sample: every slot empty) maps to -1.  Every
 * consumer of such an index clamps it, the pose-table status of such an item is 0 (inliers < min_inliers), nothing faults.
 * No reference counterpart (glue of north_star's RANSAC). */
int ph_compact_models(const float* models, int B, int H, int stride, float* out, int32_t* slot, int32_t* n_live,
                      void* stream);
/* The same with the number of live models of every group of `group` consecutive slots given: counts [B][H / group] (what
 * the solvers report as n_solutions: group = 10 for ph_solve_essential_5pt, 4 for ph_solve_p3p; H a multiple of group).
 * The counts place the pieces an item is cut into, so many blocks work on one item (14 us instead of 27 for 64 x 9 200 x 9
 * floats); which slots are live is still read from the models.  PRECONDITION: counts[b][g] equals the number of non-zero
 * models among slots g*group .. g*group + group - 1 (otherwise the output has gaps or overlaps; nothing is written outside
 * the item).  Outputs as ph_compact_models.  Shapes other than (stride 9, group 10) and (12, 4) take ph_compact_models. */
int ph_compact_models_counted(const float* models, const int32_t* counts, int group, int B, int H, int stride,
                              float* out, int32_t* slot, int32_t* n_live, void* stream);
int ph_lookup_index(const int32_t* table, const int32_t* sel, int B, int H, int32_t* out, void* stream);
/* ph_decompose_essential: the recoverPose-style decomposition of hypothesis sel[b] (NULL = 0) of every item's
 * [H][9] block of essential matrices into its four (R, t) candidates, on the device (closed form, float64; no
 * reference counterpart — north_star leaves this step on the host, where pose_estimation_amd.geometry does it by
 * SVD; this entry point removes the host round trip from the batched pipeline).  P2 [B][4][12] = [R | t] in the
 * order (Ra, t), (Ra, -t), (Rb, t), (Rb, -t), |t| = 1; P1 [B][4][12] (may be NULL) is filled with [I | 0].
 * The candidate SET equals the SVD construction's; their order and the sign of t are this function's own. */
int ph_decompose_essential(const float* E, const int32_t* sel, int B, int H,
                           float* P1, float* P2, void* stream);
int ph_pose_table(const float* P2, const int32_t* front, int C,
                  const int32_t* best_count, const int32_t* n_match, const int32_t* best_idx,
                  int B, int min_inliers, float* table, int32_t* win, void* stream);

/* Refinement of B relative poses on their correspondences (north_star: "the per-hypothesis refinement"; no
 * reference counterpart — the reference's only refinement is the LM inside cv2.solvePnP, main.py:101-104, which
 * ph_refine_pose_lm restates; this is its two-view analogue).  Levenberg-Marquardt, float64, one block per item, on
 *     sum_i min(r_i^2, trunc_sq),   r_i = the signed Sampson distance of correspondence i under E = [t]x R,
 * over the 5 degrees of freedom of (R, t), |t| = 1 (R <- exp([dw]x) R, t moved in its tangent plane).  A
 * correspondence enters the normal equations of a pass iff r_i^2 < trunc_sq at that pass's pose (truncated
 * quadratic / MSAC loss), so the inlier set follows the pose; `mask` ([B][ceil(N/64)], may be NULL) restricts the
 * correspondences considered at all.  pose_rt: B rows of `stride` (>= 12) floats, R row-major (9) then t (3) —
 * e.g. the pose table of ph_pose_table with stride 16 — refined IN PLACE (a row with a zero/non-finite t, fewer
 * than 6 correspondences or max_iters == 0 stays bit for bit).  rms [B] f64 (may be NULL) = root mean square of r
 * over the final inliers, n_inliers [B] their number, iters [B] the iterations run.  CPU twin: oracle/refine_e.py. */
int ph_refine_relative_pose(const float* x1, const float* y1, const float* x2, const float* y2,
                            const uint64_t* mask, const int32_t* n_valid, int B, int N, float trunc_sq,
                            int max_iters, float* pose_rt, int stride, double* rms, int32_t* n_inliers,
                            int32_t* iters, void* stream);

/* ------------------------------------------------------------------------------------------
 * (4) Board-pose helpers: the reference's own arithmetic, batched.
 * ------------------------------------------------------------------------------------------ */

/* cv2.projectPoints with the 5-coefficient model (k1,k2,p1,p2,k3), validate_pose.py:181-184
 * and :275.  Computed in float64 like the reference (rvec, tvec, K, dist are float64 there,
 * main.py:28-34).  pose is [B][6] = rvec(3), tvec(3); K4 = fx,fy,cx,cy; obj is [B][N][3] f32
 * (matchImagePoints hands back float32); uv is [B][N][2] f64 (may be NULL); uv_int (may be
 * NULL) is the reference's astype(int32), i.e. truncation toward zero. */
int ph_project_points(const float* obj, const double* pose, const double* K4, const double* dist5,
                      const int32_t* n_valid, int B, int N,
                      double* uv, int32_t* uv_int, void* stream);

/* calculate_reprojection_error, validate_pose.py:260-288: per-point error err[B][N] (may be
 * NULL) and the five statistics stats[B][5] = mean, std (ddof 0), max, min, num_points. */
int ph_reprojection_error(const float* obj, const float* img, const double* pose,
                          const double* K4, const double* dist5, const int32_t* n_valid,
                          int B, int N, double* err, double* stats, void* stream);

/* Levenberg-Marquardt refinement of B poses on the distorted reprojection error: the iterative
 * half of cv2.solvePnP(..., SOLVEPNP_ITERATIVE) at main.py:101-104 (== validate_pose.py:93-96),
 * batched.  pose [B][6] f64 (rvec, tvec) holds the initial guess on entry
 * and the refined pose on return; obj [B][N][3] f32, img [B][N][2] f32 (distorted pixels);
 * rms [B] (may be NULL) = sqrt(mean squared residual); iters [B] (may be NULL) = iterations run. */
int ph_refine_pose_lm(const float* obj, const float* img, const int32_t* n_valid, int B, int N,
                      const double* K4, const double* dist5, int max_iters,
                      double* pose, double* rms, int32_t* iters, void* stream);

/* The same refinement for the RANSAC path: points as SoA rows X, Y, Z, u, v [B][N] f32 (as ph_score_pnp takes
 * them; (u, v) distorted pixels under K4/dist5 — pass dist5 = 0 for undistorted pixels, K4 = (1,1,0,0) for
 * normalised coordinates) and an optional inlier mask [B][ceil(N/64)] (ph_inlier_mask_pnp): only points whose
 * bit is set enter the cost (NULL = all n_valid[b] points).  rms is over the points used. */
int ph_refine_pose_lm_soa(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                          const uint64_t* mask, const int32_t* n_valid, int B, int N,
                          const double* K4, const double* dist5, int max_iters,
                          double* pose, double* rms, int32_t* iters, void* stream);

/* The same refinement with several blocks per item for LARGE point sets (one block per item leaves all but B of the
 * 256 CUs idle): every LM iteration is one launch over (blocks per item) x B blocks, max_iters + 2 launches per call,
 * state and partial sums in the workspace (ph_refine_pose_lm_soa_ws_bytes; 0 = the point sets are small or the batch is
 * large enough, and the call is ph_refine_pose_lm_soa itself).  Same iteration, same stopping rules; the sums are added in
 * another order, so the poses of the two entry points agree to rounding (1e-9), not bit for bit.  Deterministic. */
size_t ph_refine_pose_lm_soa_ws_bytes(int B, int N);
int ph_refine_pose_lm_soa_ws(const float* X, const float* Y, const float* Z, const float* u, const float* v,
                             const uint64_t* mask, const int32_t* n_valid, int B, int N,
                             const double* K4, const double* dist5, int max_iters,
                             double* pose, double* rms, int32_t* iters, void* ws, size_t ws_bytes, void* stream);

/* Initial pose of coplanar (z = 0) point sets from the plane-to-image homography: the first half of
 * cv2.solvePnP(..., SOLVEPNP_ITERATIVE) on board points (main.py:101-104 == validate_pose.py:93-96; the algorithm inside OpenCV,
 * restated: normalised DLT on undistorted points, decomposition, nearest rotation) — what ph_refine_pose_lm starts from.
 * obj [B][N][3] f32 (z ignored), img [B][N][2] f32 pixels, n_valid [B] or NULL; pose [B][6] f64 (rvec, tvec);
 * ok [B] (may be NULL): 0 for fewer than 4 points or a non-finite result (pose then NaN), else 1. */
int ph_initial_board_pose(const float* obj, const float* img, const int32_t* n_valid, int B, int N,
                          const double* K4, const double* dist5, double* pose, int32_t* ok, void* stream);

/* (rvec, tvec) [B][6] f64 of hypothesis sel[b] (NULL = 0) of item b's [H][12] f32 block of 3x4 matrices
 * K [R|t] (K4 given) or [R|t] (K4 NULL): the initial pose ph_refine_pose_lm* starts from after ph_select_best.
 * No reference counterpart (glue between north_star's RANSAC and main.py:101-104's refinement). */
int ph_pose_from_projection(const float* P, const int32_t* sel, int H, int B, const double* K4,
                            double* pose, void* stream);

/* get_camera_position_and_orientation, main.py:112-143, batched: pose [B][6] f64 (rvec, tvec) ->
 * camera [B][22] f64 = position -R't (3) | forward R'[0,0,1]' (3, main.py:132) | up R'[0,-1,0]' (3,
 * main.py:135) | R_cam_to_world row-major (9) | quaternion [w, x, y, z] (4, main.py:138-141: SciPy's
 * from_matrix().as_quat() re-ordered, no sign canonicalisation). */
int ph_camera_from_pose(const double* pose, int B, double* camera, void* stream);

/* Inverse of the distortion model for 2-D points (pixels [n][2] f32 -> normalised, undistorted
 * [n][2] f32), `iters` fixed-point iterations in float64: the one-off undistort ahead of
 * ph_score_pnp / ph_score_essential. */
int ph_undistort_points(const float* uv, const double* K4, const double* dist5,
                        int64_t n, int iters, float* xy_norm, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* POSEHOT_H */
