// Internal launcher interface between the C-ABI layer (ic_api.hip) and the kernel files.
// Every launcher enqueues on `stream`, never allocates and never synchronises; it returns the
// hipError_t of the launch (hipSuccess == 0).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ic_hip.h"

namespace ic {

constexpr int kD = 128;      // padded descriptor dimension the MFMA kernel contracts over
constexpr int kStageK = 128;  // codebook norms are padded to a multiple of this (the largest stage)
constexpr int kMaxShards = 64;  // shard bases of ic_top2_merge travel as a kernel argument

static_assert(sizeof(ic_top2_rec) == 16, "ic_top2_rec must be 16 bytes");

// ---- ic_assign.hip ---------------------------------------------------------------------------
// Decomposition of one search.  Geometry: workgroups of `waves` waves hold rows_per_wg rows in
// registers and stream the codebook through LDS in stages of stage_k codewords.  The first
// full_tiles row tiles are scanned over the whole codebook by one workgroup each.  The remaining
// row tiles' (row tile, stage) pairs form one line, tile after tile, that tail_wgs workgroups
// walk in runs of tail_quota stages; a row tile's stages then fall into at most max_pieces runs.
constexpr int kMaxTailOrder = 1024;   // tails of more workgroups are walked in plain order
struct AssignPlan {
  int waves, stage_k, rows_per_wg;
  int stages;                      // codebook stages
  int tail_span;                   // units a tail row tile takes in the line of (row tile, stage)
                                   // pairs: `stages`, or more when the runs are aligned (padding)
  int64_t row_tiles, full_tiles;
  int tail_quota, max_pieces;
  int64_t tail_wgs;
  // The first tail_long runs of the tail have tail_quota stages, the others one less (at least
  // tail_wgs: every run has tail_quota).  The line of a partly filled round rarely divides by
  // the number of workgroup slots: 391 row tiles x 16 stages on 512 slots are 12.2 stages a
  // slot, and runs of 13 would fill only 482 slots, each for 13 stages' time.  With 112 runs of
  // 13 and 400 of 12 every slot is used, and the long runs go to workgroup ids whose partner on
  // the same CU (the id 256 further on, when the tail is the whole grid) has a short one: a CU
  // then multiplies 25 stages instead of 26.
  int64_t tail_long;
  int wide_pieces;                 // rows of 512 columns and more: every row tile is cut into this
                                   // many pieces of tail_quota stages whose workgroups share an
                                   // XCD (0: the line walk above)
  // Which run of the tail a workgroup of the tail walks (tail_ordered != 0): the runs begin at
  // codebook stages spread over the whole codebook, and the workgroups of one XCD -- ids that
  // differ by multiples of 8 -- are handed runs whose first stages follow each other, so that a
  // stage one of them has fetched into the XCD's L2 is still there when its neighbours ask for
  // it a few stages later (in plain order the XCD fetched nearly every stage once per workgroup).
  int tail_ordered;
  uint16_t tail_order[kMaxTailOrder];
  // Codebooks of at most kResidentMaxK words under many rows (resident != 0): the whole codebook
  // stays in LDS, resident_wgs persistent workgroups of 16 waves (one per CU) are launched, and
  // every WAVE walks tiles of 32 rows on its own -- no stage loads, no barriers after the first.
  // The other fields then describe the same search as row tiles of 512 rows scanned whole
  // (full_tiles = row_tiles, no tail): two lists per row, which is all k_rescore needs to know.
  int resident, resident_wgs;
};
constexpr int64_t kResidentMaxK = 256;
// D: columns of a row after padding to a multiple of 128 (more than 128: always the 8-wave
// geometry)
void assign_plan(int64_t n, int64_t k, int num_cus, AssignPlan* p, int D = kD);
// records the partial buffer of a plan holds: two per row (one per lane half), plus
// 2 max_pieces - 2 more for every row of the tail
int64_t assign_part_records(const AssignPlan& p, int64_t n);
// once per device, with that device current: function attributes of the search kernels
hipError_t assign_init();

// The codebook's auxiliary block, float [kAuxNorms + kpad + D] (kpad = k rounded up to kStageK, D
// = columns rounded up to a multiple of 128):
constexpr int kAuxCnMax = 128;    // [0, 128) mean vector m (D = 128); [128] max_j |c_j - m|^2
constexpr int kAuxCr2Max = 129;   // [129] max_j |c_j|^2
constexpr int kAuxMeanNorm = 130; // [130] |m|, rounded up
constexpr int kAuxNorms = 256;    // [256 + j] = |c_j - m|^2 for j < k, 3.2e38 for k <= j < kpad
constexpr int kMaxWideD = 2048;   // widest (padded) row the search handles
// behind the norms, D floats: for D = 128 the DOUBLED mean 2 m (what the search kernel centres
// with: -2 (x - m) = fma(-2, x, 2 m) in one instruction), for wider rows the mean itself, which
// does not fit in front
__host__ __device__ inline int64_t aux_tail_offset(int64_t kpad) { return kAuxNorms + kpad; }
inline int64_t aux_mean_offset(int64_t kpad, int D) { return D > kD ? aux_tail_offset(kpad) : 0; }
// Codebooks of 128 columns and kOrderMinK < k <= kOrderMaxK codewords are SEARCHED in ascending
// order of |c_j - m|^2 (a codeword near the mean is a likelier nearest neighbour of any row, and
// the sooner a scan meets its eventual winners, the fewer of the later tiles pass the ranking
// gate: 1.3 points of the matrix peak at 1M rows x 10k words).  The block then also holds, behind
// the doubled mean: perm int32 [kpad] (search position -> codeword index), rank int32 [kpad]
// (codeword index -> search position) and the codebook in search order, float [k][128]; the norms
// at kAuxNorms are in search order.  Only the matrix-core kernel sees search positions: it
// translates them back as it writes its lists, and every tie is broken by codeword index as before.
constexpr int64_t kOrderMinK = 256;
constexpr int64_t kOrderMaxK = 16384;
static_assert(kResidentMaxK <= kOrderMinK, "codebooks that stay in LDS are searched as stored");
inline bool aux_ordered(int64_t k, int D) { return D == kD && k > kOrderMinK && k <= kOrderMaxK; }
inline int64_t aux_perm_offset(int64_t kpad) { return kAuxNorms + kpad + kD; }
inline int64_t aux_rank_offset(int64_t kpad) { return aux_perm_offset(kpad) + kpad; }
inline int64_t aux_cperm_offset(int64_t kpad) { return aux_rank_offset(kpad) + kpad; }
inline int64_t aux_floats(int64_t k, int64_t kpad, int D) {
  return kAuxNorms + kpad + D + (aux_ordered(k, D) ? 2 * kpad + k * kD : 0);
}
// the flagged rows' second look runs on a fixed grid; cut into pieces it holds at most one
// partial record per (row, piece) with groups x pieces <= the grid, i.e. 16 * kExactGrid records
constexpr int kExactGrid = 1024;   // (a launch of 4096 workgroups that find nothing to do costs 3 us more)

// dst [n][D] <- src [n][d] with zeros beyond column d
hipError_t launch_pad_rows(const float* src, int64_t n, int d, int D, float* dst, hipStream_t s);
// aux <- auxiliary block of c [k][D] (D a multiple of 128); scratch:
// codebook_aux_scratch_bytes(k, D) bytes
size_t codebook_aux_scratch_bytes(int64_t k, int D);
hipError_t launch_codebook_aux(const float* c, int64_t k, int D, float* aux, char* scratch,
                               hipStream_t s);
hipError_t launch_row_sqnorm(const float* x, int64_t n, int d, float* out, hipStream_t s);
// The lists of a search, assign_part_records entries each: per row, piece of its row tile and
// lane half the (|c-m|^2 - 2 (x-m).c, index) of the best and the second best, the value of the
// third best and -- only when the caller wants the second neighbour too (mode2) -- the index of
// the third best and the value of the fourth.
struct PartBufs {
  ic_top2_rec* rec;
  float* d3;
  int32_t* i3;   // mode2 only
  float* d4;     // mode2 only
};
// rowband (may be NULL) [n]: the constant part of each row's band, for launch_rescore's shortcut.
// zero_word (may be NULL): a device word the kernel clears (the flagged-row counter).
// x [n][D], c [k][D]; mean = the mean vector inside aux (aux itself when D = 128).  Rows wider
// than 128 columns need rowband [n] (always) and wide_xc [n][D] floats of workspace, the centred
// copy of the rows.  perm (may be NULL): c is the codebook in scan order and perm [k] maps a scan
// position to the codeword's index, which is what the lists then hold.
hipError_t launch_top2_partial(const float* x, int64_t n, const float* c, int64_t k,
                               const float* aux, const float* mean, int D, const AssignPlan& plan,
                               bool mode2, const float* thr, const PartBufs& pb, float* rowband,
                               float* wide_xc, uint32_t* zero_word, const int32_t* perm,
                               hipStream_t s);
// thr[row] = upper bound on the kernel's value of the row's nearest codeword, from a guessed index
// rank (may be NULL): codeword index -> position of its norm in aux (codebooks in search order)
hipError_t launch_hint_threshold(const float* x, int64_t n, const float* c, int64_t k,
                                 const float* aux, const float* mean, int D, const int32_t* hint,
                                 const int32_t* rank, float* thr, hipStream_t s);
// merge the pieces, evaluate the two candidates as sum (x-c)^2, order, sqrt, ratio test; the
// rows whose pair is not certain to hold the answer are appended to list [n] i32 (any order),
// *count (cleared beforehand, see zero_word above) counts them; tickets [kExactGrid] u32 are
// cleared for launch_exact_rows.
// rowband != NULL (only i1 wanted, top-1): rows whose second-best value lies outside the band of
// the best are answered from the records alone.
int64_t rescore_chunk(int64_t n);
hipError_t launch_rescore(const float* x, int64_t n, const float* c, const float* aux,
                          const float* mean, int D, const PartBufs& pb, const float* rowband,
                          const AssignPlan& plan, bool mode2, int32_t* i1, float* d1, int32_t* i2,
                          float* d2, float ratio, uint8_t* mask, ic_top2_rec* rec, int32_t* list,
                          uint32_t* count, uint32_t* tickets, hipStream_t s);
// search the listed rows again by direct differences and overwrite their outputs.
// partial: 16 * kExactGrid records; tickets [kExactGrid], zero on entry.
hipError_t launch_exact_rows(const float* x, int64_t n, const float* c, int64_t k, int D,
                             const int32_t* list, const uint32_t* count, ic_top2_rec* partial,
                             uint32_t* tickets, int32_t* i1, float* d1, int32_t* i2, float* d2,
                             float ratio, uint8_t* mask, ic_top2_rec* rec, hipStream_t s);
hipError_t launch_top2_merge(const ic_top2_rec* gathered, int n_shards, int64_t n,
                             const int64_t* shard_base_host, int64_t* i1, float* d1, int64_t* i2,
                             float* d2, float ratio, uint8_t* mask, hipStream_t s);

// direct float32 search (sum (x-c)^2 per pair, no expansion): the on-device checker
hipError_t launch_exact_top2(const float* x, int64_t n, const float* c, int64_t k, int D,
                             int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                             uint8_t* mask, hipStream_t s);

// ---- ic_reduce.hip ---------------------------------------------------------------------------
hipError_t reduce_init();   // once per device: function attributes of the histogram kernel
// cell_w (nullable): per-cell weights, bin b belongs to cell b / cell_k (raw counts only).
hipError_t launch_hist(const int32_t* codes, const int64_t* offsets, int64_t n_img, int64_t k,
                       int norm_mode, double power, const double* cell_w, int64_t cell_k,
                       double* out, hipStream_t s);
// keys [n_img * n_kp * n_levels] int32 and offsets [n_img + 1] int64 are workspace.
hipError_t launch_spatial_hist(const int32_t* codes, const int32_t* kp_cells, int64_t n_img,
                               int64_t n_kp, int n_levels, int64_t k, int64_t n_cells,
                               const double* cell_w, int32_t* keys, int64_t* offsets, double* out,
                               hipStream_t s);

// ---- ic_centroid.hip -------------------------------------------------------------------------
// Decomposition and workspace layout of the partition-based centroid sums.
struct CentroidPlan {
  int64_t nblk;      // 4096-row sort blocks
  int64_t nb;        // buckets of 32 clusters
  int n_ranges;      // chunk ranges per bucket (partial sums folded afterwards)
  bool use_scan;     // table too large: fall back to the scan-based kernel
  size_t o_keys, o_tab, tab_bytes, o_total, o_start, o_perm, o_psums, o_pcounts, bytes;
};
void centroid_plan(int64_t n, int64_t k, int d, int num_cus, CentroidPlan* p);
hipError_t launch_centroid_partials(const float* x, const int32_t* codes, int64_t n, int64_t k,
                                    int d, const CentroidPlan& p, char* ws, double* sums,
                                    int64_t* counts, hipStream_t s);

// ---- ic_reduce.hip (continued) ----------------------------------------------------------------
void centroid_plan_scan(int64_t n, int64_t k, int num_cus, int* n_rr);
hipError_t launch_centroid_partials_scan(const float* x, const int32_t* codes, int64_t n,
                                         int64_t k, int d, int n_rr, double* part_sums,
                                         int64_t* part_counts, double* sums, int64_t* counts,
                                         hipStream_t s);
hipError_t launch_centroid_reduce(const double* part_sums, const int64_t* part_counts, int64_t k,
                                  int d, int n_ranges, double* sums, int64_t* counts,
                                  hipStream_t s);
hipError_t launch_centroid_finalize(const double* sums, const int64_t* counts, const float* prev_c,
                                    int64_t k, int d, float* new_c, uint8_t* has_members,
                                    hipStream_t s);

// ---- ic_knn.hip ------------------------------------------------------------------------------
hipError_t launch_pairdist(const float* x, int64_t m, const float* c, int64_t n, int d, int metric,
                           double* out, hipStream_t s);
// chunks a row is cut into for the top-k (1 = one workgroup per row); partial buffers hold
// m * chunks * k entries
int topk_chunks(int64_t m, int64_t n, int k);
hipError_t launch_topk_rows(const double* dist, int64_t m, int64_t n, int k, int nch,
                            double* part_d, int32_t* part_i, double* out_d, int32_t* out_i,
                            hipStream_t s);

// ---- ic_match.hip ----------------------------------------------------------------------------
hipError_t launch_compact_matches(const uint8_t* mask, const int32_t* i1, const int64_t* row_off,
                                  int64_t n_cand, const double* kp_q, int64_t n_q,
                                  const double* kp_rows, int64_t* counts_ws, int64_t* out_off,
                                  int32_t* match_q, int32_t* match_im, double* pts, hipStream_t s);
hipError_t launch_ransac(const double* pts, const int64_t* off, int64_t n_cand, int n_iter,
                         const int32_t* samples, uint64_t seed, double th, int min_inliers,
                         int flags, int32_t* n_inl_ws, double* H, int32_t* n_inliers,
                         double* best_err, int32_t* best_it, uint8_t* mask, hipStream_t s);

}  // namespace ic
