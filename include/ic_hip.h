/*
 * ic_hip.h -- C ABI of libic_hip.so: nearest-codeword / nearest-neighbour search over
 * local image descriptors on AMD MI355X (gfx950).
 *
 * The reference (winkash/image-classification) has no FFI of its own: this path is reached
 * through numpy-in / numpy-out Python calls into scipy / scikit-learn.  Each entry point below
 * names the reference call it replaces (path:line relative to the reference tree).
 *
 * Conventions
 *   - every function returns 0 on success, otherwise a non-zero code (a hipError_t value, or
 *     one of the IC_E* codes below); ic_last_error() gives the text for the calling thread's
 *     last failure on that handle.
 *   - all array pointers are DEVICE pointers owned by the caller (row-major, contiguous) unless
 *     a parameter is documented as "host".  The library never frees caller memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls enqueue work
 *     and return; the caller synchronises the stream before reading outputs.
 *   - a handle owns a grow-only device workspace (partial results, padded norms).  A call that
 *     needs more than the handle holds first waits for the handle's own previous call (an event,
 *     not a device synchronisation), then frees and reallocates; ic_reserve_workspace does that
 *     up front.  Calls on one handle are serialised by a mutex, so a handle may be shared by
 *     threads.  Handles must not cross fork(): create one per process.
 *   - row width d of the descriptor-search entry points (ic_assign_*, ic_codebook_aux):
 *     1 <= d <= 2048.  Rows are searched in slabs of 128 columns; a width that is not a multiple
 *     of 128 is zero-padded into workspace copies first.  With d a multiple of 128 (the 128-d
 *     local descriptors above all) x and c are read in place, 16 bytes at a time: both must be
 *     16-byte aligned (IC_EINVAL otherwise).  d <= 128 is the tuned case; wider rows (the
 *     projected face signatures of vision/facerec/face_indexing.py:141,158) run the same
 *     algorithm with the row slabs re-read once per 64 codewords.
 */
#ifndef IC_HIP_H
#define IC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IC_OK 0
#define IC_EINVAL 10001   /* bad argument (null pointer, negative size, unsupported d, ...) */
#define IC_ENOMEM 10002   /* workspace allocation failed */
#define IC_EHANDLE 10003  /* null / destroyed handle */
#define IC_ELIMIT 10004   /* input larger than the static scratch of ic_assign_top1_static */

/* Size of the device scratch buffer ic_assign_top1_static works in (per GPU, held from the moment
 * the library is loaded on that GPU until the process ends). */
#define IC_STATIC_SCRATCH_BYTES (64u << 20)

/* normalisation modes of ic_hist */
#define IC_NORM_NONE 0     /* raw counts */
#define IC_NORM_L1 1       /* count * (1.0 / (sum or 1))   -- model/features.py:1178-1179 */
#define IC_NORM_L2 2       /* count / (sqrt(sum count^2) or 1) */
#define IC_NORM_POWER_L2 3 /* y = count^power, then y / (sqrt(sum y^2) or 1) */

/* Metrics of ic_pairwise / ic_knn_rows (a = search row, b = indexed row). */
#define IC_METRIC_EUCLIDEAN 0    /* sqrt(sum (a-b)^2)        sklearn 'euclidean'               */
#define IC_METRIC_SQEUCLIDEAN 1  /* sum (a-b)^2                                                 */
#define IC_METRIC_L1 2           /* sum |a-b|                sklearn 'manhattan'                */
#define IC_METRIC_INTERSECTION 3 /* sum min(a,b)             model/clustering.py:309-310        */
#define IC_METRIC_CHISQR 4       /* sum (a-b)^2/a over a!=0  model/clustering.py:305-306        */

/* per-row partial result exchanged between shards / K-splits (16 bytes) */
typedef struct ic_top2_rec {
  float d1;   /* best distance (Euclidean, not squared) */
  int32_t i1; /* its index inside the shard, -1 if none */
  float d2;   /* second best distance */
  int32_t i2; /* its index inside the shard, -1 if none */
} ic_top2_rec;

/* Library / handle management.  No counterpart in the reference (scipy/sklearn need none). */
int ic_version(void);
int ic_create(int device, void** handle);
int ic_destroy(void* handle);
const char* ic_last_error(void* handle); /* handle may be NULL: error of the last failed ic_create */

/* The handle's device workspace (partial results of the calls below) grows on demand: a call
 * that needs more than the handle holds waits for the handle's previous call, frees the block and
 * allocates a larger one.  ic_reserve_workspace does that up front (grow only; at least `bytes`
 * afterwards), so that a loop of calls of known size never allocates; ic_workspace_bytes reports
 * the current size -- run the largest call once, read it, reserve that on the next start.  No
 * counterpart in the reference. */
int ic_reserve_workspace(void* handle, size_t bytes);
int ic_workspace_bytes(void* handle, size_t* bytes);

/* Timing of the distance kernel with HIP events recorded on the caller's stream, one pair per
 * ic_assign_* call (the descriptor search kernel) or per ic_knn_rows batch (the pairwise distance
 * kernel) between begin and end.  ic_profile_end waits for those events and returns
 * the summed kernel time in milliseconds and the number of launches.  No counterpart in the
 * reference; used by bench.py for the roofline figure. */
int ic_profile_begin(void* handle);
int ic_profile_end(void* handle, double* kernel_ms_sum, int64_t* launches);

/* out[i] = sum_j x[i][j]^2.  Replaces the squared-norm terms scipy's _vq.vq and sklearn's
 * euclidean_distances compute implicitly (model/features.py:1166,
 * vision/utils/scene_functions.py:391). */
int ic_row_sqnorm(void* handle, const float* x, int64_t n, int d, float* out, void* stream);

/* Auxiliary block of a codebook: its mean vector, the squared norms of the codewords about that
 * mean and two maxima that bound the search kernel's rounding error.  Every ic_assign_* call
 * needs it; a caller that searches one codebook many times (a vocabulary, model/features.py:1166;
 * a logo database, model/robust_matching.py:73-80) computes it once and passes it as `c_aux`
 * instead of NULL.  It replaces the squared-norm terms scipy's _vq.vq and sklearn's
 * euclidean_distances recompute on every call.  aux [ic_codebook_aux_floats_d(k, d)] f32, device
 * (ic_codebook_aux_floats(k) is the size for d <= 128: wider rows keep their mean vector, which
 * no longer fits the block's fixed head, behind the norms).  For d <= 128 and 257 ... 16384
 * codewords the block also holds the codebook once more, in the order the search scans it
 * (ascending distance to the mean), with the permutation both ways: about k * 520 bytes more,
 * included in the sizes these functions report.  A block belongs to the codebook contents it was
 * computed from: compute it again when the codebook changes. */
int64_t ic_codebook_aux_floats(int64_t k);
int64_t ic_codebook_aux_floats_d(int64_t k, int d);
int ic_codebook_aux(void* handle, const float* c, int64_t k, int d, float* aux, void* stream);

/* Test hook, no counterpart in the reference: after every following ic_assign_* call on this
 * handle the number of rows that took the second, direct look is copied (on the call's stream) to
 * the device word `count`; NULL switches it off. */
int ic_debug_flagged(void* handle, uint32_t* count); /* `count` must stay valid until reset */

/* Test hook, no counterpart in the reference, host arithmetic only (no handle, no GPU): how a
 * search of n rows x k codewords x d columns would be cut into workgroups on a device with
 * `num_cus` compute units.  out [15] int64: waves per workgroup, codewords per stage, rows per
 * workgroup, stages, places a row tile of the tail takes in the line of (row tile, stage) units,
 * row tiles, row tiles scanned whole, stages per workgroup of the tail, most pieces a row tile
 * falls into, workgroups of the tail, pieces per row tile of the wide search (0: none), records
 * of the partial buffer, 1 when the codebook stays in LDS whole and every wave walks 32-row tiles
 * on its own (else 0), workgroups of that launch, workgroups of the tail whose run has the full
 * number of stages (the others have one less).  The CPU tests replay the kernels' index
 * arithmetic against it. */
int ic_debug_plan(int64_t n, int64_t k, int d, int num_cus, int64_t* out);
/* The same plan's hand-out of the tail's runs: order[b] = the run that workgroup b of the tail
 * walks (the runs are dealt to the XCDs by the codebook stage they begin at); *count = entries
 * written, 0 when the tail is walked in plain order.  Test hook like ic_debug_plan. */
int ic_debug_plan_order(int64_t n, int64_t k, int d, int num_cus, int32_t* order, int cap,
                        int* count);

/* Nearest codeword per row: idx[i] = argmin_j ||x_i - c_j|| (lowest j on ties), dist[i] = that
 * Euclidean distance.  Replaces scipy.cluster.vq.vq at model/features.py:1166 and
 * vision/facerec/face_indexing.py:141, and argmin(euclidean_distances(..., squared=True)) at
 * vision/utils/scene_functions.py:391.
 *   x [n][d] f32, c [k][d] f32, c_aux = the codebook's auxiliary block from ic_codebook_aux or
 *   NULL (computed on every call), idx [n] int32, dist [n] f32 or NULL.
 * Results are those of the direct float32 search (ic_assign_top2_exact) on every row: the
 * matrix-core pass knows the rounding error of the values it ranks by, and rows it cannot decide
 * with certainty are searched again by direct differences inside the same call. */
int ic_assign_top1(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                   const float* c_aux, int32_t* idx, float* dist, void* stream);

/* ic_assign_top1 for callers that cannot allocate.  It has no counterpart in the reference; it
 * exists for code under HIP graph capture, tight serving loops and harnesses that forbid
 * hipMalloc.  Same arguments and the same results as ic_assign_top1, but the partial results
 * (padded norms, per-row records of the K-splits, zero-padded copies when d < 128) live in a
 * static device array of IC_STATIC_SCRATCH_BYTES instead of the handle's growable workspace, and
 * the call consists of kernel launches on `stream` only: no hipMalloc / hipFree, no event, no
 * synchronisation.  Inputs that need more scratch are rejected with IC_ELIMIT before anything is
 * launched (about 50 bytes per row, more for the rows of the last partly filled round of
 * workgroups, plus 4 bytes per codeword: 1M rows x 10k codewords fits, about 1.3M rows is the
 * ceiling).  All calls on one GPU share that array -- from every handle and every stream --
 * so two calls that overlap in time on one GPU race; order them on one stream or with events. */
int ic_assign_top1_static(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                          const float* c_aux, int32_t* idx, float* dist, void* stream);

/* ic_assign_top1 with a guess of every row's nearest codeword (k-means: the labels of the
 * previous Lloyd iteration inside kmeans2, vision/facerec/face_indexing.py:158).  The result is
 * the same as without the guess; a good guess lets the kernel skip ranking candidates that
 * cannot beat it.  hint [n] int32, values outside [0,k) mean "no guess". */
int ic_assign_top1_hint(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                        const float* c_aux, const int32_t* hint, int32_t* idx, float* dist,
                        void* stream);

/* Two nearest index rows per search row, ascending (distance, index), plus the ratio test
 * mask[i] = d1[i] < d2[i] * ratio.  Replaces KNeighborsClassifier.kneighbors(x, 2) at
 * model/classifiers.py:195-197 as used by model/robust_matching.py:80, and the Python ratio
 * loop at model/robust_matching.py:81-84.
 *   i2/d2 may be NULL (top-1 only); mask may be NULL; when k == 1, i2 = -1 and d2 = +inf. */
int ic_assign_top2(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                   const float* c_aux, int32_t* i1, float* d1, int32_t* i2, float* d2,
                   float ratio, uint8_t* mask, void* stream);

/* Same contract as ic_assign_top2, computed by direct differences sum (x-c)^2 in float32 on the
 * vector ALU (what scipy's _vq.vq does for small inputs, model/features.py:1166) instead of the
 * norm expansion on the matrix cores: about four times slower (78.9 ms against 18.6 ms for 1M x 10k), free of the expansion's rounding.
 * It is the on-device checker of the fast path at sizes no CPU oracle can handle. */
int ic_assign_top2_exact(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                         int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                         uint8_t* mask, void* stream);

/* Same search, but the result is written as one ic_top2_rec per row (indices local to this
 * index shard) for a later ic_top2_merge.  No counterpart in the reference: it exists because
 * the index side of model/robust_matching.py:73-80 is sharded over GPUs. */
int ic_assign_top2_rec(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                       const float* c_aux, ic_top2_rec* rec, void* stream);

/* Merge per-shard top-2 records: gathered [n_shards][n] records; shard_base (HOST pointer,
 * n_shards int64) is added to each shard's local indices.  With shard_base == NULL the bases
 * travel inside the data instead: gathered is [n_shards][n + 1] records and the last record of a
 * shard's block holds its base as an int64 in its first eight bytes -- what ranks exchange in one
 * all-gather, so that nothing has to come back to the host before the merge.
 * Order is (distance, global index).
 * Outputs as ic_assign_top2 but with int64 indices (sklearn returns int64,
 * model/classifiers.py:196).  i2/d2/mask may be NULL. */
int ic_top2_merge(void* handle, const ic_top2_rec* gathered, int n_shards, int64_t n,
                  const int64_t* shard_base, int64_t* i1, float* d1, int64_t* i2, float* d2,
                  float ratio, uint8_t* mask, void* stream);

/* Per-image bag-of-words histogram.  Replaces np.histogram(code, bins=range(0, K+1))[0] and the
 * L1 normalisation at model/features.py:1174-1181 (one call for all images instead of the
 * reference's per-image Python loop).
 *   codes [offsets[n_img]] int32 (values outside [0,k) are ignored, as np.histogram does),
 *   offsets [n_img+1] int64 ascending (image i owns codes[offsets[i]:offsets[i+1]]; empty
 *   images give an all-zero row), out [n_img][k] float64. */
int ic_hist(void* handle, const int32_t* codes, const int64_t* offsets, int64_t n_img, int64_t k,
            int norm_mode, double power, double* out, void* stream);

/* Spatial-pyramid histogram.  Replaces _get_spatial_histogram + compute_spatial_features
 * (vision/utils/scene_functions.py:279-373): every keypoint adds 1 to bin (cell, code) of the
 * cell it falls in on each pyramid level, and the counts of a cell are multiplied by the cell's
 * weight.  The keypoint grid is shared by all images (model/features.py:1301-1304), so the
 * caller passes the cell of every keypoint on every level.
 *   codes [n_img][n_kp] int32, kp_cells [n_levels][n_kp] int32 (cell numbers 0 .. n_cells-1 with
 *   n_cells = sum 4^l, a negative cell skips the vote), cell_weights [n_cells] float64 (device),
 *   out [n_img][n_cells * k] float64, cells in ascending order, k words each. */
int ic_spatial_hist(void* handle, const int32_t* codes, const int32_t* kp_cells, int64_t n_img,
                    int64_t n_kp, int n_levels, int64_t k, const double* cell_weights,
                    double* out, void* stream);

/* k-means update step, part 1: sums[j] = sum of rows with codes[i] == j (float64), counts[j] =
 * number of them.  Replaces scipy's _vq.update_cluster_means inside kmeans2
 * (vision/facerec/face_indexing.py:158).  Deterministic: fixed summation order, no atomics.
 *   x [n][d] f32, codes [n] int32 in [0,k), sums [k][d] f64, counts [k] int64. */
int ic_centroid_partials(void* handle, const float* x, const int32_t* codes, int64_t n, int64_t k,
                         int d, double* sums, int64_t* counts, void* stream);

/* k-means update step, part 2: new_c[j] = sums[j] / counts[j], or prev_c[j] when counts[j] == 0
 * (scipy kmeans2 missing='warn' keeps the previous centroid, vision/facerec/face_indexing.py:158).
 *   has_members [k] uint8 or NULL receives counts[j] != 0. */
int ic_centroid_finalize(void* handle, const double* sums, const int64_t* counts,
                         const float* prev_c, int64_t k, int d, float* new_c,
                         uint8_t* has_members, void* stream);

/* Distance matrix between rows of any width under one of the IC_METRIC_* metrics.  Replaces the
 * matrix scikit-learn's brute-force neighbour search builds for KNNScikit over bag-of-words
 * histograms (model/classifiers.py:170-172 with the metrics vision/logo_recognition/training.py:41
 * and knn_training.py:24 configure) and scipy's cdist at vision/facerec/face_indexing.py:473.
 *   x [m][d] f32, c [n][d] f32, out [m][n] float64 (terms in f32, accumulated in float64). */
int ic_pairwise(void* handle, const float* x, int64_t m, const float* c, int64_t n, int d,
                int metric, double* out, void* stream);

/* The k nearest indexed rows of every search row, ascending by (distance, index).  Replaces
 * KNeighborsClassifier.kneighbors(features, n_neighbors, return_distance) behind
 * KNNScikit.get_neighbors (model/classifiers.py:195-197) for metrics and widths the 128-d
 * descriptor search does not cover.  Note that IC_METRIC_INTERSECTION is a similarity: like the
 * reference, which hands it to scikit-learn as a distance, the rows with the SMALLEST value come
 * first.
 *   x [m][d] f32, c [n][d] f32, 1 <= k <= n, out_dist [m][k] float64, out_idx [m][k] int32. */
int ic_knn_rows(void* handle, const float* x, int64_t m, const float* c, int64_t n, int d,
                int metric, int k, double* out_dist, int32_t* out_idx, void* stream);

/* Ratio-test mask -> per-candidate match lists.  Replaces the Python loop at
 * model/robust_matching.py:81-84 and the keypoint gathers at :85-86 for all candidate images of a
 * query at once.  Candidate c owns search rows row_offsets[c] .. row_offsets[c+1]; its matches,
 * in ascending row order, go to positions out_offsets[c] .. out_offsets[c+1] of the outputs.
 *   mask [rows] uint8 and i1 [rows] int32 as written by ic_assign_top2, row_offsets [n_cand+1]
 *   int64 (device), out_offsets [n_cand+1] int64, match_q / match_im [rows] int32 (query row,
 *   row inside the candidate).  With kp_query [n_query][2] and kp_rows [rows][2] (float64 keypoint
 *   coordinates) and pts [rows][4] non-NULL, pts receives (x_q, y_q, x_im, y_im) per match. */
int ic_compact_matches(void* handle, const uint8_t* mask, const int32_t* i1,
                       const int64_t* row_offsets, int64_t n_cand, const double* kp_query,
                       int64_t n_query, const double* kp_rows, int64_t* out_offsets,
                       int32_t* match_q, int32_t* match_im, double* pts, void* stream);

/* Homography RANSAC over the matched pairs of every candidate image at once.  Replaces
 * RobustMatcher._robust_h_estimation (model/robust_matching.py:196-225) in both of its forms,
 * chosen by `flags`:
 *  - without IC_RANSAC_CV_RULES, the reference's own ransac(data, HomographyModel, 4, ...)
 *    (:291-365): per iteration a homography through four sampled pairs, inliers = other pairs
 *    with reprojection error (HomographyModel.get_error, :259-283) below th; the first iteration
 *    with the most inliers is refitted over sample + inliers by the direct linear transform and
 *    kept when it has at least min_inliers pairs and (IC_RANSAC_MEAN_ERR) a mean reprojection
 *    error <= th, as :359-363 demands; all n_iter iterations count.
 *  - with IC_RANSAC_CV_RULES, the published loop of cv2.findHomography(src, dst, cv2.RANSAC, th),
 *    the reference's default (:214-216): every pair whose squared error is <= th^2 is an inlier
 *    (the sample gets no special treatment), a model is kept when it has strictly more inliers
 *    than every earlier one and more than three, every kept model lowers the iteration limit to
 *    log(1 - 0.995) / log(1 - (inliers / pairs)^4) (n_iter is the initial limit, OpenCV's is
 *    2000), the mask is the kept minimal model's, and H is refitted over those inliers and
 *    polished by at most ten Levenberg-Marquardt steps.  best_iter is the kept iteration.
 *   pts [total][4] float64 (x_src, y_src, x_dst, y_dst), offsets [n_cand+1] int64 (device),
 *   samples [n_cand][n_iter][4] int32 pair indices inside the candidate, or NULL to draw them
 *   from a counter-based generator seeded with `seed`;
 *   H [n_cand][9] float64 (zeros when rejected), n_inliers [n_cand] int32 (0 when rejected),
 *   best_err [n_cand] float64, best_iter [n_cand] int32 or NULL, inlier_mask [total] uint8.
 * Candidates with fewer than max(4, min_inliers) pairs can never be accepted and are rejected
 * without scoring (best_iter -1, best_err +inf). */
#define IC_RANSAC_MEAN_ERR 1
#define IC_RANSAC_CV_RULES 2
int ic_ransac_homography(void* handle, const double* pts, const int64_t* offsets, int64_t n_cand,
                         int n_iter, const int32_t* samples, uint64_t seed, double th,
                         int min_inliers, int flags, double* H, int32_t* n_inliers,
                         double* best_err, int32_t* best_iter, uint8_t* inlier_mask, void* stream);

/* Host-side helper of the upload path; touches no GPU and needs no handle.  Replaces the
 * np.concatenate of the per-image descriptor arrays that precedes every search in the reference
 * (model/features.py:1164): the n_jobs source blocks are laid one after the other into dst
 * (normally a page-locked upload buffer) by `threads` plain threads that share the output
 * evenly.  src[j] points to elems[j] values, float32 when kind[j] == 0 (kind may be NULL: all
 * float32) or float64 when kind[j] == 1 (converted to float32).  All pointers are HOST memory. */
int ic_host_gather_f32(float* dst, const void* const* src, const int64_t* elems,
                       const int32_t* kind, int64_t n_jobs, int threads);

#ifdef __cplusplus
}
#endif
#endif /* IC_HIP_H */
