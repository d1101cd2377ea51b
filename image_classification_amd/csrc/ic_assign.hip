// Distance search kernels for gfx950 (MI355X): fused  cn - 2 x.c  on the fp32 matrix cores with a
// running top-2 argmin kept in registers, the exact re-scoring pass, norms and the shard merge.
//
// What they replace in the reference (paths relative to the reference tree):
//   scipy.cluster.vq.vq                       model/features.py:1166, vision/facerec/face_indexing.py:141
//   argmin(euclidean_distances(squared=True)) vision/utils/scene_functions.py:391
//   KNeighborsClassifier.kneighbors(x, 2)     model/classifiers.py:195-197 <- model/robust_matching.py:80
//   the ratio-test loop                       model/robust_matching.py:81-84
//
// Layout of the main kernel (k_top2_partial):
//   * MFMA v_mfma_f32_32x32x2_f32 with A = codeword tile (32 codewords, from LDS) and B = 32
//     descriptors (registers).  The 32x32 result then has one DESCRIPTOR per lane and 16 CODEWORDS
//     in that lane's registers, so the running top-2 of a descriptor never leaves its lane.
//   * the contraction index is permuted: lane half h multiplies the components of the even
//     (h = 0) or odd (h = 1) 16-byte chunks of "its" row, for both operands.  The two lanes of a
//     descriptor then fetch neighbouring chunks, so a load instruction touches 32 cache lines,
//     not 64.
//   * a wave keeps one tile of 32 descriptors (pre-scaled by -2) in registers for the whole sweep
//     over the codebook; the accumulator is initialised with |c|^2, so the MFMA chain leaves
//     |c|^2 - 2 x.c directly (|x|^2 is constant per row and cannot change the argmin).
//   * codeword tiles are staged global -> LDS with global_load_lds (16 B per lane), two stage
//     buffers; rows are padded to 528 bytes (through the SOURCE address: the LDS image of such a
//     load is lane-linear) so that ds_read_b128 is bank-conflict free with immediate offsets.
//   * grid = the row tiles that fill whole rounds of the machine, each scanning the entire
//     codebook, followed by the workgroups of the last, partly filled round: those walk equal
//     shares of the linearised (row tile, stage) space of the remaining row tiles, crossing from
//     the end of one row tile into the start of the next, so every CU gets the same amount of
//     work whatever the number of row tiles; every piece of a row tile writes one
//     (value, index) x 2 record per row and lane half, k_rescore merges the pieces.
//   * ranking is gated twice (tile minimum, then per register) because on gfx950 the fp32 MFMA
//     and the vector ALU never co-execute: every ranking instruction costs MFMA time.
//
// Exactness.  The values the matrix cores rank by, |c - m|^2 - 2 (x - m).c (m = the codebook's
// mean vector: distances do not change under a common shift, the magnitudes that drive the
// rounding error do), carry an absolute error of up to `B` (see error_bound).  Besides the two
// best candidates every lane therefore tracks the VALUE of the third best, and ranks every
// candidate that comes within the error band of the best (top-1 searches) or of the second best
// (top-2 searches).  k_rescore then knows, per row, whether any candidate outside the pair it holds
// could be the true nearest (or second nearest): if not, evaluating the pair with direct
// differences decides the row; if so the row is flagged and searched again by direct differences
// over the whole codebook (k_exact_rows).  The direct-difference arithmetic is the same function in
// both places, so results equal those of the all-direct search (ic_assign_top2_exact) on every row.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "ic_internal.h"

namespace ic {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// One 16-byte chunk of a row centred on the codebook mean and scaled by -2: fma(-2, x, 2 m) rounds
// once, exactly like x - m, and doubling is exact.  Two components per instruction (v_pk_fma_f32);
// sq gathers the squares, even components in one half and odd ones in the other.
__device__ __forceinline__ void centre_chunk(const f32x4 x, const f32x4 mu2, float* out, f32x2& sq) {
  const f32x2 m2 = {-2.0f, -2.0f};
  const f32x2 a = __builtin_elementwise_fma(m2, x.lo, mu2.lo);
  const f32x2 b = __builtin_elementwise_fma(m2, x.hi, mu2.hi);
  sq = __builtin_elementwise_fma(a, a, sq);
  sq = __builtin_elementwise_fma(b, b, sq);
  out[0] = a.x;
  out[1] = a.y;
  out[2] = b.x;
  out[3] = b.y;
}

constexpr int kTileK = 32;                          // codewords per MFMA tile
constexpr int kRowBytes = kD * 4 + 16;              // 528: one 16-byte pad chunk per codeword row
constexpr int kRowChunks = kRowBytes / 16;          // 33
// Sizes of one pipeline stage of SK codewords in LDS.
template <int SK>
struct Stage {
  static constexpr int kTileBytes = SK * kRowBytes;      // 1 KiB = one wave-instruction of 16 B lanes
  static constexpr int kGlds = kTileBytes / 1024;        // 33 per 64 codewords
  static constexpr int kTiles = SK / kTileK;
  static constexpr int kBytes = kTileBytes + SK * 4;     // + the stage's squared norms
  static_assert(kTileBytes % 1024 == 0, "a stage must be a whole number of 1 KiB pieces");
  static_assert(kGlds * 64 <= 8448, "issue_stage divides by 33 with a multiply-shift that is "
                                    "exact only below 8448");
  static_assert(SK % 64 == 0, "the stage's norms are fetched 64 at a time");
  static_assert(kBytes % 16 == 0, "stage buffers must stay 16-byte aligned");
  static_assert(kStageK % SK == 0, "the padded norms must cover whole stages");
};

// Geometry.  Every wave holds ONE tile of 32 descriptors (64 components per lane, 96 VGPRs), so
// four waves fit a SIMD: while one wave ranks, waits for the last MFMA of its tile, for an LDS
// read or at the stage barrier, three others have MFMAs to issue.  (Two tiles per wave and two
// waves per SIMD, the first layout, left the matrix pipe idle about 6 % of the time; four tiles
// per wave made hipcc spill.)  Two shapes of workgroup:
//   large codebooks: one workgroup of 16 waves per CU, stages of 128 codewords (136 KB of LDS);
//   others:          two workgroups of 8 waves per CU, stages of 64 codewords (68 KB each).  A
//                    workgroup spends about 10 us loading its rows and first stages before it
//                    multiplies; with two independent workgroups on a CU the other one multiplies
//                    meanwhile, which matters while the scan is short.
// Measured (% of the fp32 MFMA peak, two workgroups / one workgroup): 256 codewords 74.1 / 65.1,
// 1000: 83.8 / 80.2, 4000: 89.0 / 88.0, 6000: 88.4 / 88.5, 10 000: 89.3 / 89.5, 20 000: 90.3 / 90.7.
constexpr int kXT = 1;
constexpr int kSmallCodebook = 8192;   // below this: two 8-wave workgroups per CU

// Row tiles that fill whole rounds of the machine are searched over the entire codebook by one
// workgroup each (the longer a lane's scan, the rarer a tile improves its top-2, so ranking gets
// cheaper, and X is read once).  The row tiles of the last, partly filled round would leave CUs
// idle (391 row tiles on 512 workgroup slots) or, cut into equal K-pieces, cost a row-tile load
// and a pipeline fill per piece (six pieces per row tile were the best such cut for 100 000 x
// 1000, at 60 % of peak).  Instead the (row tile, stage) pairs of those row tiles are laid out in
// one line, tile after tile, and cut into `tail_wgs` runs of `tail_quota` stages: workgroup g of
// the tail takes run g, which may end one row tile and begin the next.  Every workgroup then has
// the same amount of matrix work, a row tile is loaded by at most max_pieces workgroups, and the
// codebook stages keep streaming through LDS across the change of row tile.
// Rows of kWidePieceD columns and more are cut differently (see k_top2_wide).
constexpr int kWidePieceD = 512;
constexpr int kWidePieces = 16;
// Codebooks that fit in LDS whole are kept there from this many rows on (k_top2_resident); below,
// loading 132 KB into every CU before the first multiplication costs more than it saves
// (256 codewords, streamed / resident: 32 000 rows 27 / 31 us, 64 000 and 100 000 equal, 200 000
// 141 / 123 us, 1 M 536 / 502 us).
constexpr int64_t kResidentMinRows = 98304;

// The tail of the grid: tail row tile tt (counted from the first row tile of the tail) owns the
// units [tt S, (tt + 1) S) of the line of (row tile, stage) pairs; run g of the tail owns q units
// when g < n_long and q - 1 otherwise, one run after the other.  The pieces of a row tile are
// the runs its units fall into, numbered from the first such run.
struct TailGeom {
  int stages;        // S: units a row tile takes in the line (AssignPlan::tail_span)
  int quota;         // q
  int rows_per_wg;
  int64_t n_long;    // AssignPlan::tail_long
};
__host__ __device__ __forceinline__ int64_t tail_run_start(const TailGeom& t, int64_t run) {
  return run * t.quota - (run > t.n_long ? run - t.n_long : 0);
}
// the run unit u belongs to
__host__ __device__ __forceinline__ int64_t tail_run_of(const TailGeom& t, int64_t u) {
  const int64_t long_units = t.n_long * t.quota;
  if (u < long_units || t.quota < 2) return u / t.quota;
  return t.n_long + (u - long_units) / (t.quota - 1);
}
__host__ __device__ __forceinline__ int tail_first_run(const TailGeom& t, int64_t tt) {
  return (int)tail_run_of(t, tt * t.stages);
}
__host__ __device__ __forceinline__ int tail_pieces(const TailGeom& t, int64_t tt) {
  return (int)tail_run_of(t, (tt + 1) * t.stages - 1) - tail_first_run(t, tt) + 1;
}

void assign_plan(int64_t n, int64_t k, int num_cus, AssignPlan* p, int D) {
  const bool wide = D > kD;
  // IC_AMD_GEOMETRY forces a shape (1 = two 8-wave workgroups per CU, 2 = one of 16; tuning aid)
  static const int geom = [] {
    const char* v = getenv("IC_AMD_GEOMETRY");
    return v ? atoi(v) : 0;
  }();
  // rows wider than 128 columns always take the 8-wave geometry (k_top2_wide has no other)
  // ... and so do few rows against a long codebook: when the 16-wave geometry has fewer than two
  // (row tile, stage) units per CU, most CUs would stay idle while the others multiply runs of
  // two 128-word stages (one image's 1000 descriptors against 10 000 words: 79 workgroups,
  // 77 us; the 8-wave geometry's 256-row tiles and 64-word stages give 314)
  const bool few = geom == 0 && !wide && k >= kSmallCodebook &&
                   ((n + 511) / 512) * ((k + 127) / 128) < 2 * (int64_t)num_cus;
  const bool small = wide || few || geom == 1 || (geom != 2 && k < kSmallCodebook);
  p->waves = small ? 8 : 16;
  p->stage_k = small ? 64 : 128;
  p->rows_per_wg = kXT * p->waves * 32;
  const int slots = num_cus * (small ? 2 : 1);  // workgroups resident at once
  const int64_t stages = (k + p->stage_k - 1) / p->stage_k;
  p->stages = (int)stages;
  p->row_tiles = (n + p->rows_per_wg - 1) / p->rows_per_wg;
  // IC_AMD_QUOTA forces the tail decomposition on every row tile with that many stages per
  // workgroup (tuning aid, and the tests' way to reach many pieces per row tile)
  static const int forced = [] {
    const char* v = getenv("IC_AMD_QUOTA");
    return v ? atoi(v) : 0;
  }();
  p->full_tiles = forced > 0 ? 0 : p->row_tiles / slots * slots;
  const int64_t tail = p->row_tiles - p->full_tiles;
  p->tail_quota = (int)stages;
  p->tail_span = (int)stages;
  p->tail_wgs = tail;
  p->tail_long = 0x7fffffff;
  p->max_pieces = 1;
  p->wide_pieces = 0;
  p->tail_ordered = 0;
  p->resident = 0;
  p->resident_wgs = 0;
  // IC_AMD_RESIDENT: 0 = never keep the codebook resident, 1 = whenever it fits (tuning aid and
  // the tests' way to reach that kernel with few rows)
  static const int resident_mode = [] {
    const char* v = getenv("IC_AMD_RESIDENT");
    return v ? atoi(v) : -1;
  }();
  if (!wide && k <= kResidentMaxK && forced <= 0 && geom == 0 && resident_mode != 0 &&
      (resident_mode == 1 || n >= kResidentMinRows)) {
    p->waves = 16;
    p->stage_k = 128;
    p->rows_per_wg = 512;
    p->stages = (int)((k + 127) / 128);
    p->row_tiles = (n + 511) / 512;
    p->full_tiles = p->row_tiles;
    p->tail_quota = p->stages;
    p->tail_span = p->stages;
    p->tail_wgs = 0;
    p->resident = 1;
    const int64_t wave_tiles = (n + 31) / 32;
    const int64_t wgs = (wave_tiles + 3) / 4;     // at least one tile per SIMD
    p->resident_wgs = (int)(wgs < num_cus ? wgs : num_cus);
    return;
  }
  if (wide && D >= kWidePieceD && forced <= 0) {
    // A workgroup of the wide search reads its 256 rows again for every chunk of 64 codewords,
    // and the rows of all resident workgroups together (0.5 MB x D / 128 ... 1 GB at 2048
    // columns) fit no cache: walked whole, a row tile's scan is bound by those reads.  So every
    // row tile is cut into P pieces of q chunks, and the P workgroups of a row tile get
    // consecutive places on ONE XCD (workgroup id = 8 * place + XCD): they start together, walk
    // the slabs of the same rows at the same pace, and the XCD's L2 fetches each slab once for
    // all of them.  P = 16 (every chunk its own piece below 16 chunks), more when the row tiles
    // are too few to fill the machine otherwise.
    int64_t want = (slots + p->row_tiles - 1) / p->row_tiles;
    if (want < kWidePieces) want = kWidePieces;
    if (want > stages) want = stages;
    const int64_t q = (stages + want - 1) / want;
    const int64_t P = (stages + q - 1) / q;
    p->full_tiles = 0;
    p->tail_quota = (int)q;
    p->tail_span = (int)(P * q);
    p->tail_wgs = p->row_tiles * P;
    p->max_pieces = (int)P;
    p->wide_pieces = (int)P;
    return;
  }
  if (tail == 0) return;
  // cutting pays when the idle slots of the round are worth more than the extra row-tile loads
  // (about two stages' time per workgroup)
  if (forced <= 0 && (slots - tail) * stages <= 2 * (int64_t)slots) return;
  const int64_t units = tail * stages;
  int64_t q = forced > 0 ? forced : (units + slots - 1) / slots;
  // one-stage runs only reload row tiles -- unless the whole search is less than one stage a
  // slot: then nothing is gained by leaving slots empty, and a call's latency is what counts
  const int64_t q_min = (stages < 2 || units <= slots) ? 1 : 2;
  if (q < q_min) q = q_min;
  if (q > stages) q = stages;                      // (only when forced)
  p->tail_quota = (int)q;
  p->tail_wgs = (units + q - 1) / q;
  const int64_t pieces = (stages - 1) / q + 2;     // runs a row tile's stages can fall into
  p->max_pieces = (int)(pieces < stages ? pieces : stages);
  // Few row tiles against a long codebook (a handful of queries against a 2M-row index): cut
  // every row tile at the SAME stages instead, P pieces each with P a multiple of 8, so that
  // the workgroups that scan one range of the codebook for different row tiles are resident
  // together on one XCD (workgroup ids that differ by a multiple of 8) and its L2 fetches the
  // range once for all of them; with runs that start anywhere, every row tile's scan fetched the
  // whole codebook from HBM again.  In the line of units a row tile then takes P q >= stages
  // places (the last piece of a tile is shorter).  Taken when it costs at most 2 % of balance.
  if (forced <= 0 && !wide && tail * 8 <= slots) {
    const int64_t P = slots / tail / 8 * 8;
    const int64_t qa = (stages + P - 1) / P;
    const int64_t Pa = (stages + qa - 1) / qa;     // pieces that are not empty
    if (qa >= q_min && Pa * qa * 100 <= stages * 102 && qa * 100 <= q * 102) {
      p->tail_quota = (int)qa;
      p->tail_span = (int)(Pa * qa);
      p->tail_wgs = tail * Pa;
      p->max_pieces = (int)Pa;
    }
  }
  // Runs of two lengths (see AssignPlan::tail_long): q stages for the first n_long runs, q - 1
  // for the others, one run per slot.
  if (forced <= 0 && !wide && p->tail_span == (int)stages && q >= 2 && units > (q - 1) * slots &&
      units < q * slots) {
    p->tail_long = units - (q - 1) * slots;
    p->tail_wgs = slots;
    const int64_t most = (stages - 1) / (q - 1) + 2;
    p->max_pieces = (int)(most < stages ? most : stages);
  }
  // runs that begin anywhere in the codebook: hand them to the XCDs by first stage
  if (!wide && p->tail_span == (int)stages && p->tail_wgs > 8 && p->tail_wgs <= kMaxTailOrder &&
      stages > 2) {
    const int nw = (int)p->tail_wgs;
    std::vector<int> runs((size_t)nw);
    for (int r = 0; r < nw; ++r) runs[(size_t)r] = r;
    const TailGeom geom{p->tail_span, p->tail_quota, p->rows_per_wg, p->tail_long};
    std::stable_sort(runs.begin(), runs.end(), [&](int a, int b) {
      return tail_run_start(geom, a) % stages < tail_run_start(geom, b) % stages;
    });
    // an XCD gets a window of consecutive first stages; inside the window the long runs take
    // the XCD's lowest workgroup ids, so that a CU's two workgroups are not both long ones
    int next = 0;
    for (int xcd = 0; xcd < 8; ++xcd) {
      int mine = 0;
      for (int b = 0; b < nw; ++b) mine += (p->full_tiles + b) % 8 == xcd;
      std::stable_partition(runs.begin() + next, runs.begin() + next + mine,
                            [&](int r) { return r < p->tail_long; });
      for (int b = 0; b < nw; ++b)
        if ((p->full_tiles + b) % 8 == xcd) p->tail_order[b] = (uint16_t)runs[(size_t)next++];
    }
    p->tail_ordered = 1;
  }
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void glds16(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void glds4(const float* src, char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
}

// Stage `st` = codewords [SK st, SK (st + 1)) -> LDS buffer sb.  The LDS image of a
// global_load_lds is lane-linear (wave base + 16 * lane), so the padded image (528-byte rows) is
// produced through the SOURCE address: the lane that lands on 16-byte chunk `pos` of the buffer
// fetches chunk pos % 33 of codeword row pos / 33 (the 33rd chunk of a row is padding and re-reads
// chunk 0).  Rows past the codebook end re-read the last row; their norm slot holds +inf, so they
// can never win.
// The byte offsets a lane's loads of a stage start at (inside the stage's 64 KB or 32 KB of
// codeword rows) do not depend on the stage: they are worked out once per kernel and kept as
// 16-bit halves of a few registers, so that issuing a stage costs one vector instruction per
// load instead of the nine of the division by 33 (vector instructions are MFMA time on this
// part: 2 % of the 8-wave geometry's stage, 1 % of the 16-wave one's).
template <int NW, int SK>
struct StageOffsets {
  static constexpr int kLoads = (Stage<SK>::kGlds + NW - 1) / NW;   // per wave and stage
  unsigned packed[(kLoads + 1) / 2];
  static_assert((SK - 1) * 512 + 31 * 16 < 65536, "offsets are kept in 16 bits");
};
__device__ __forceinline__ unsigned stage_offset(int pos, int last) {
  int r = (pos * 1986) >> 16;  // pos / 33, exact for pos < 8448
  int c = pos - r * kRowChunks;
  c = c == kRowChunks - 1 ? 0 : c;
  r = r < last ? r : last;
  return (unsigned)r * 512u + (unsigned)c * 16u;
}
template <int NW, int SK>
__device__ __forceinline__ void stage_offsets(StageOffsets<NW, SK>& o, int wave, int lane) {
#pragma unroll
  for (int i = 0; i < StageOffsets<NW, SK>::kLoads; ++i) {
    int m = wave + NW * i;
    m = m < Stage<SK>::kGlds ? m : Stage<SK>::kGlds - 1;
    const unsigned v = stage_offset(m * 64 + lane, SK - 1);
    if (i & 1) o.packed[i / 2] |= v << 16;
    else o.packed[i / 2] = v;
  }
}

template <int NW, int SK>
__device__ __forceinline__ void issue_stage(const float* __restrict__ C, int K,
                                            const float* __restrict__ cnp, int st, char* sb,
                                            int wave, int lane, const StageOffsets<NW, SK>& o) {
  // wave-uniform 64-bit base + 32-bit per-lane byte offset (keeps the address math off the VALU)
  const char* base = reinterpret_cast<const char*>(C) + (size_t)st * (SK * kD * 4);
  const int last = K - 1 - st * SK;  // last valid row of this stage (>= 0)
  if (last >= SK - 1) {
#pragma unroll
    for (int i = 0; i < StageOffsets<NW, SK>::kLoads; ++i) {
      const int m = wave + NW * i;
      if (m < Stage<SK>::kGlds) {
        // (through asm so that the halves are taken apart here, one instruction each, and not
        // hoisted out of the sweep into registers of their own)
        unsigned voff;
        if (i & 1) asm volatile("v_lshrrev_b32 %0, 16, %1" : "=v"(voff) : "v"(o.packed[i / 2]));
        else asm volatile("v_and_b32 %0, 0xffff, %1" : "=v"(voff) : "v"(o.packed[i / 2]));
        glds16(reinterpret_cast<const float*>(base + voff), sb + m * 1024);
      }
    }
  } else {
    // the codebook ends inside this stage: rows past its end re-read the last one
    // (opaque copy: these offsets are never hoisted out of the sweep loop)
    asm volatile("" : "+v"(lane));
#pragma unroll
    for (int i = 0; i < StageOffsets<NW, SK>::kLoads; ++i) {
      const int m = wave + NW * i;
      if (m < Stage<SK>::kGlds)
        glds16(reinterpret_cast<const float*>(base + stage_offset(m * 64 + lane, last)),
               sb + m * 1024);
    }
  }
#pragma unroll
  for (int part = 0; part < SK / 64; ++part)
    if (wave == NW - 1 - part)
      glds4(cnp + (size_t)st * SK + part * 64 + lane, sb + Stage<SK>::kTileBytes + part * 256);
}

// A lane's list.  The values are KEYS: the computed value with its four lowest mantissa bits
// replaced by the number (0..15) of the accumulator register it came from, so that ranking a
// candidate is three instructions on floats (insert into the ascending triple with min / med3)
// and its place inside the tile needs no bookkeeping; the TILE a key came from is worked out once
// per tile from which keys are new (resolve_tiles).  What the replaced bits cost in accuracy is
// part of the rounding bound below.
struct Top2 {
  float d1, d2, d3;   // keys of the best, the second best and the third best
  int i1, i2;         // tiles of the best and the second best (-1: none)
  // searches that want the second neighbour too (MODE2) also keep the third best's tile and
  // the key of the fourth: three candidates within the band of the second in one list are
  // then still resolved by k_rescore instead of sending the row to the direct search
  float d4;
  int i3;
};
// "No candidate": what every list starts from.  The squared norms of the padding behind the last
// codeword lie above it (kPadNorm), so padding is never ranked, and both are finite because a key
// made from +inf would be a NaN.
constexpr int kDenseTiles = 12;   // see tile_epilogue_gated1
constexpr float kNone = 3.0e38f;
constexpr float kPadNorm = 3.2e38f;

// Rounding model.  The MFMA chain is a k-ordered float32 fma chain: starting from cn = |c - m|^2
// it adds the 128 products -2 (x - m)_i c_i, one rounding each, every partial sum bounded by
// cn + 2 |x - m| |c|.  With u = 2^-24, and 16 u cn for the rounding of cn itself and 2 u |x-m||c|
// for that of x - m, the computed value differs from the exact one by at most
//     u (144 cn + 258 |x - m| |c|)  <=  u (150 Cn + 270 |x - m| Cr),
// Cn = max_j |c_j - m|^2, Cr = max_j |c_j|.  Turning the value into a key (four mantissa bits
// replaced) moves it by less than 16 units in the last place, 32 u |v| with
// |v| <= Cn + 2 |x - m| Cr, so the keys the lists hold differ from the exact values by at most
//     B := u (182 Cn + 334 |x - m| Cr).
constexpr float kU = 5.9604645e-8f;            // 2^-24
constexpr float kGammaDirect = 129.f * kU;      // relative error bound of a 128-term sum (x-c)^2
__device__ __forceinline__ float error_bound(float xn_centred, float cn_max, float cr2_max) {
  return kU * (182.f * cn_max + 334.f * __builtin_sqrtf(xn_centred * cr2_max));
}
// Rows of D = 128 nslab columns (the wide search): the chain has D terms, the norm is a sum of
// nslab partial sums.  D = 128 gives the constants above.
__device__ __forceinline__ float error_bound_d(float xn_centred, float cn_max, float cr2_max,
                                               int D) {
  const float a = (float)(D + 53 + D / kD), b = (float)(2 * D + 78);
  return kU * (a * cn_max + b * __builtin_sqrtf(xn_centred * cr2_max));
}
__device__ __forceinline__ float gamma_direct_d(int D) { return (float)(D + 1) * kU; }

// Band below which another candidate may really be nearer than the reference candidate with
// computed value v: two error bounds, plus the room the direct float32 evaluation of the two
// squared distances (about R + v each, R = |x-m|^2 + 2 (x-m).m) needs to order them.
// In the search kernel: band(v) = c0 + kBandSlope * v with c0 = 2.01 B + kBandSlope * R.
constexpr float kBandSlope = 4.f * kGammaDirect;
// k_rescore decides with 2 B + 3 gamma s (s = the direct squared distance), which is never more.
constexpr float kFlagSlope = 3.f * kGammaDirect;

// Strict '<' and ascending visiting order keep the lowest index on ties, as scipy's _vq.vq scan
// and numpy's argmin do.
__device__ __forceinline__ void top2_update(float& d1r, float& d2r, int& i1r, int& i2r, float v,
                                            int enc) {
  // locals only: every ?: below has plain values on both arms, so it stays a v_cndmask
  const float d1 = d1r, d2 = d2r;
  const int i1 = i1r, i2 = i2r;
  const bool lt1 = v < d1;
  const bool lt2 = v < d2;
  const int t = lt2 ? enc : i2;
  i2r = lt1 ? i1 : t;
  i1r = lt1 ? enc : i1;
  d2r = __builtin_amdgcn_fmed3f(v, d1, d2);
  d1r = lt1 ? v : d1;
}

// value -> key: the four lowest mantissa bits become the register number r (one v_and_or_b32)
__device__ __forceinline__ float key_of(float v, int r) {
  return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & 0xfffffff0u) | (unsigned)r);
}
// insert x into the ascending triple / quadruple, dropping the largest
__device__ __forceinline__ void insert3(float& k1, float& k2, float& k3, float x) {
  k3 = __builtin_amdgcn_fmed3f(k2, k3, x);
  k2 = __builtin_amdgcn_fmed3f(k1, k2, x);
  // the plain instruction: fminf costs three (the compiler first quiets a signalling NaN in
  // either operand, also when the minimum is written as a median with -inf)
  asm("v_min_f32 %0, %1, %2" : "=v"(k1) : "v"(k1), "v"(x));
}
__device__ __forceinline__ void insert4(float& k1, float& k2, float& k3, float& k4, float x) {
  k4 = __builtin_amdgcn_fmed3f(k3, k4, x);
  insert3(k1, k2, k3, x);
}

// Third smallest of the union of two ascending triples.
__device__ __forceinline__ float third_of_union(float a1, float a2, float a3, float e1, float e2,
                                                float e3) {
  // third smallest of the four values a1 <= a2, e1 <= e2, then the thirds themselves
  const float t4 = __builtin_fminf(__builtin_fmaxf(a2, e1), __builtin_fmaxf(a1, e2));
  return __builtin_fminf(t4, __builtin_fminf(a3, e3));
}

__device__ __forceinline__ bool lex_less(float da, int ia, float db, int ib) {
  return da < db || (da == db && ia < ib);
}

// a and b are each ordered pairs; result = the two smallest of the four by (value, index).
// Index -1 marks "none" (value +inf).
__device__ __forceinline__ void top2_merge_pairs(float& d1, int& i1, float& d2, int& i2, float e1,
                                                 int j1, float e2, int j2) {
  // treat "none" as larger than anything real with the same value
  auto less = [](float da, int ia, float db, int ib) {
    if (ia < 0) return false;
    if (ib < 0) return true;
    return lex_less(da, ia, db, ib);
  };
  float r1, r2;
  int k1, k2;
  if (less(e1, j1, d1, i1)) {
    r1 = e1; k1 = j1;
    if (less(e2, j2, d1, i1)) { r2 = e2; k2 = j2; } else { r2 = d1; k2 = i1; }
  } else {
    r1 = d1; k1 = i1;
    if (less(e1, j1, d2, i2)) { r2 = e1; k2 = j1; } else { r2 = d2; k2 = i2; }
  }
  d1 = r1; i1 = k1; d2 = r2; i2 = k2;
}

template <int XT>
__device__ __forceinline__ void compute_tile(const char* tile_base, const char* cn_base,
                                             const float (&xr)[XT][64], f32x16 (&acc)[XT], int i,
                                             int h) {
  // accumulator register g*4+e of lane half h is codeword 8g + 4h + e of the tile
  const f32x4* cnv = reinterpret_cast<const f32x4*>(cn_base + h * 16);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 c4 = cnv[g * 2];
#pragma unroll
    for (int xt = 0; xt < XT; ++xt) {
      acc[xt][g * 4 + 0] = c4[0];
      acc[xt][g * 4 + 1] = c4[1];
      acc[xt][g * 4 + 2] = c4[2];
      acc[xt][g * 4 + 3] = c4[3];
    }
  }
  // row stride 528 B: lanes i and i+1 start one 16-byte bank slot apart, so each 16-lane group of
  // a ds_read_b128 covers all 16 slots of the 256-byte bank row (conflict free); lane half h
  // takes the odd (h = 1) or even 16-byte chunks of the row
  const f32x4* arow = reinterpret_cast<const f32x4*>(tile_base + i * kRowBytes + h * 16);
  f32x4 a = arow[0];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    f32x4 an = a;
    if (q + 1 < 16) an = arow[2 * (q + 1)];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int xt = 0; xt < XT; ++xt) {
        acc[xt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xr[xt][q * 4 + e], acc[xt], 0, 0, 0);
      }
    }
    a = an;
  }
}

__device__ __forceinline__ float min3f(float a, float b, float c) {
  return __builtin_fminf(__builtin_fminf(a, b), c);
}

// Ranking with an early out.  On gfx950 the f32 MFMA runs at the vector-ALU rate, and ranking
// instructions hand-placed between the MFMAs of a chain did not hide behind them (measured: the
// interleaved schedule was 6 % slower than ranking after the chain, and the same loop without any
// ranking reaches 94.8 % of the MFMA peak against 83 % with it).  So ranking is made cheaper by
// doing less of it, in two levels: the minimum of the tile's 16 values (8 v_min3) is compared with
// the lane's gate, and when no lane of the wave passes, the tile is skipped; inside a tile that
// some lane can improve, a register's candidate is ranked only when some lane's value in THAT
// register passes (16 compares and wave-uniform branches instead of 16 updates).
//
// The gate of a lane is `ref + band(ref)`, ref = the running best (MODE2 false: the caller wants
// the nearest codeword only) or the running second best (MODE2 true: the caller wants both).
// Everything below the gate is ranked, so at the end the lane's list holds the three smallest
// values among all candidates that matter: a candidate that was never ranked lay, when it was
// seen, above ref + band(ref) of that moment, and both only shrink afterwards.  For a top-1
// search this means the second and third entries are exact only inside the band of the best --
// the only place k_rescore looks at them -- and a candidate between the band and the running
// second best, which a plain top-2 search would have to rank, is skipped: about half the
// ranking work of the top-2 gate.
// Within a tile the gate of the tile's start is used (stale = larger = ranks a little more).
// After a tile's keys have been inserted: which tiles do the list's keys come from?  The old
// entries keep their order among themselves, so walking the new list from the front, an entry
// is the next unused old one exactly when it equals it (a new key with the very same bits -- the
// same register of another tile, the same value to sixteen units in the last place -- may take
// an old entry's tile and leave its own to the twin behind it: either way the list names two
// different candidates of that value); every other entry is new and belongs to `tile`.
template <bool MODE2>
__device__ __forceinline__ void resolve_tiles(Top2& st, float k1, float k2, float k3, float k4,
                                              int tile) {
  const float o1 = st.d1, o2 = st.d2, o3 = st.d3;
  const int p1 = st.i1, p2 = st.i2, p3 = st.i3;
  const bool e1 = k1 == o1;
  const float a2 = e1 ? o2 : o1;
  const int q2 = e1 ? p2 : p1;
  const bool e2 = k2 == a2;
  st.i1 = e1 ? p1 : tile;
  st.i2 = e2 ? q2 : tile;
  if constexpr (MODE2) {
    const float x3 = e2 ? o3 : o2, y3 = e2 ? o2 : o1;
    const int xq = e2 ? p3 : p2, yq = e2 ? p2 : p1;
    const float a3 = e1 ? x3 : y3;
    const int q3 = e1 ? xq : yq;
    st.i3 = k3 == a3 ? q3 : tile;
    st.d4 = k4;
  }
  st.d1 = k1;
  st.d2 = k2;
  st.d3 = k3;
}

// dense: the tile is one of the first of its scan, where nearly every register holds a candidate
// some lane must rank: all sixteen are inserted without asking (straight-line code: sixteen
// compare-and-branch pairs cost more than the inserts they would skip).
// SHORT: the geometry of short scans, where most tiles are dense ones; there the dense tiles
// skip the tile-minimum test too (in the long-scan geometry that changed nothing but the code
// layout, for the worse).
template <bool MODE2, bool SHORT>
__device__ __forceinline__ void tile_epilogue_gated1(const f32x16& a, Top2& st, float& gate,
                                                     float c0, int tile, float one_plus_slope,
                                                     bool dense = false) {
  if (SHORT && dense) {
    // no test at all: the tile minimum would pass in some lane anyway
    float k1 = st.d1, k2 = st.d2, k3 = st.d3, k4 = st.d4;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float x = key_of(a[r], r);
      if constexpr (MODE2) insert4(k1, k2, k3, k4, x);
      else insert3(k1, k2, k3, x);
    }
    resolve_tiles<MODE2>(st, k1, k2, k3, k4, tile);
    gate = __builtin_fmaf(MODE2 ? k2 : k1, one_plus_slope, c0);
    return;
  }
  // the minima of five triples of registers (and the sixteenth register) on the way to the
  // tile minimum: a tile that passes is usually passed by one candidate of one lane, so the
  // per-register tests are asked only inside the triples whose minimum passes (six tests and
  // three more, instead of sixteen)
  const float t0 = min3f(a[0], a[1], a[2]), t1 = min3f(a[3], a[4], a[5]), t2 = min3f(a[6], a[7], a[8]);
  const float t3 = min3f(a[9], a[10], a[11]), t4 = min3f(a[12], a[13], a[14]);
  const float m = min3f(min3f(t0, t1, t2), min3f(t3, t4, a[15]), __builtin_inff());
  if (__builtin_amdgcn_ballot_w64(m < gate) != 0) {
    float k1 = st.d1, k2 = st.d2, k3 = st.d3, k4 = st.d4;
    const float g = gate;
    if (!SHORT && dense) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = key_of(a[r], r);
        if constexpr (MODE2) insert4(k1, k2, k3, k4, x);
        else insert3(k1, k2, k3, x);
      }
    } else {
      const float tmin[5] = {t0, t1, t2, t3, t4};
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        if (q < 5 && __builtin_amdgcn_ballot_w64(tmin[q < 5 ? q : 0] < g) == 0) continue;
#pragma unroll
        for (int r = 3 * q; r < (q < 5 ? 3 * q + 3 : 16); ++r)
          if (__builtin_expect(__builtin_amdgcn_ballot_w64(a[r] < g) != 0, 0)) {
            // lanes that did not pass: needless, never wrong
            const float x = key_of(a[r], r);
            if constexpr (MODE2) insert4(k1, k2, k3, k4, x);
            else insert3(k1, k2, k3, x);
          }
      }
    }
    resolve_tiles<MODE2>(st, k1, k2, k3, k4, tile);
    gate = __builtin_fmaf(MODE2 ? k2 : k1, one_plus_slope, c0);
  }
}

template <int XT, bool MODE2, bool SHORT>
__device__ __forceinline__ void tile_epilogue_gated(const f32x16 (&acc)[XT], Top2 (&st)[XT],
                                                    float (&gate)[XT], const float (&c0)[XT],
                                                    int tile, bool dense) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt)
    tile_epilogue_gated1<MODE2, SHORT>(acc[xt], st[xt], gate[xt], c0[xt], tile, 1.f + kBandSlope,
                                       dense);
}

// Every candidate ranked (IC_AMD_VARIANT=0: the timing baseline of the gate).
__device__ __forceinline__ void tile_epilogue_all1(const f32x16& a, Top2& st, int tile) {
  float k1 = st.d1, k2 = st.d2, k3 = st.d3, k4 = st.d4;
#pragma unroll
  for (int r = 0; r < 16; ++r) insert4(k1, k2, k3, k4, key_of(a[r], r));
  resolve_tiles<true>(st, k1, k2, k3, k4, tile);
}

template <int XT>
__device__ __forceinline__ void tile_epilogue_all(const f32x16 (&acc)[XT], Top2 (&st)[XT],
                                                  int tile) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) tile_epilogue_all1(acc[xt], st[xt], tile);
}

// (tile, key) -> codeword index: register g * 4 + e of lane half h is codeword 8 g + 4 h + e of
// the tile
// (an index at or beyond K can only come from non-finite input, whose keys are NaNs: none)
__device__ __forceinline__ int decode_candidate(int tile, float key, int h, int K) {
  if (tile < 0) return -1;
  const int r = (int)(__builtin_bit_cast(unsigned, key) & 15u);
  const int idx = tile * kTileK + 8 * (r >> 2) + 4 * h + (r & 3);
  return idx < K ? idx : -1;
}

// Slot of row r's record in list l = 2 * (piece of its row tile) + (lane half).  The two lanes
// that hold a descriptor rank disjoint halves of every tile's codewords and each writes its own
// record: kept apart, the halves are two more places where k_rescore can find the INDEX of a
// candidate inside the error band, which quarters the rows that need the second look.  Lists 0
// and 1 exist for every row ([2][N] records); only the rows from split_row0 on (the last, partly
// filled round of row tiles) can have further lists, stored behind as
// [2 max_pieces - 2][N - split_row0].
__device__ __forceinline__ int64_t part_slot(int l, int64_t r, int64_t N, int64_t split_row0) {
  return l < 2 ? (int64_t)l * N + r : 2 * N + (int64_t)(l - 2) * (N - split_row0) + (r - split_row0);
}

// AssignPlan::tail_order as a kernel argument
struct TailOrder {
  int on;
  uint16_t run[kMaxTailOrder];
};

#ifdef IC_TRACE
// Timeline instrumentation of the search kernel (tools/wgtrace.py; built as a separate library by
// `make trace`, never part of libic_hip.so): per workgroup the 100 MHz wall clock at entry, at the
// first barrier, when its first rows had landed and were centred (a workgroup that goes on to
// another row tile overwrites these two with the time before / after that change), at exit, and
// where it ran.
constexpr int kTraceWgs = 1 << 16;
__device__ unsigned long long ic_trace_buf[kTraceWgs][8];
#define IC_TRACE_PUT(slot, value)                                                         \
  do {                                                                                    \
    if (threadIdx.x == 0 && blockIdx.x < kTraceWgs) ic_trace_buf[blockIdx.x][slot] = (value); \
  } while (0)
#define IC_TRACE_NOW(slot) IC_TRACE_PUT(slot, wall_clock64())
#else
#define IC_TRACE_PUT(slot, value) do {} while (0)
#define IC_TRACE_NOW(slot) do {} while (0)
#endif

// aux: the codebook's auxiliary block (see AuxLayout in ic_internal.h): mean vector, the two
// maxima of the rounding model, then the centred squared norms (+inf in the padding).
// pb: where the lists go (rec / d3 for every search, i3 / d4 only with MODE2).
//
// A workgroup walks `left` consecutive units of the line of (row tile, stage) pairs, starting at
// (tile, st): the first full_tiles workgroups one whole row tile each, the others a run of the
// tail.  The codebook stages stream through the two LDS buffers without regard to row tiles (the
// stage after the last one is stage 0 again); when the walk crosses into the next row tile the
// waves store their lists, load the new rows into the same registers and go on.
template <int XT, int NW, int SK, bool GATED, bool MODE2>
__global__ __launch_bounds__(NW * 64, 4) void k_top2_partial(
    const float* __restrict__ X, int64_t N, const float* __restrict__ C, int K,
    const float* __restrict__ aux, int stages_total, int tail_span, int full_tiles,
    int64_t tail_units, int tail_quota, int tail_long, const float* __restrict__ thr, PartBufs pb,
    float* __restrict__ rowband, uint32_t* __restrict__ zero_word,
    const int32_t* __restrict__ perm, TailOrder order) {
  static_assert(XT == 1, "one 32-descriptor tile per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;  // descriptor (B column / result column) and codeword row (A row)
  const int h = lane >> 5;  // which half of the 128 components this lane multiplies
  const float* __restrict__ cnp = aux + kAuxNorms;
  // the search is the first kernel of a call: it clears the flagged-row counter k_rescore adds to
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
  IC_TRACE_NOW(0);
  IC_TRACE_PUT(3, ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) |
                      (unsigned)__builtin_amdgcn_s_getreg(63492));
  IC_TRACE_PUT(4, 0ull);
  IC_TRACE_PUT(5, 0ull);
  IC_TRACE_PUT(6, (unsigned long long)clock64());

  // where this workgroup's walk begins, how many units it has, and which piece of its first row
  // tile that is (a row tile entered at stage 0 is always piece 0)
  // (rows are counted in 32 bits: the entry points refuse n >= 2^31, and 64-bit row numbers
  // would keep loop-invariant register pairs alive across the sweep)
  int tile = blockIdx.x;
  int st = 0, left = stages_total, piece = 0;
  if ((int)blockIdx.x >= full_tiles) {
    // (tail_span = stages_total: runs may cross into the next row tile; larger: the runs are
    // aligned to the row tiles and the places behind a tile's last stage are padding)
    // which run of the tail this workgroup walks (see AssignPlan::tail_order)
    const int b = (int)blockIdx.x - full_tiles;
    const int run = order.on ? (int)order.run[b < kMaxTailOrder ? b : 0] : b;
    // (the first tail_long runs have tail_quota units, the others one less)
    const TailGeom geom{tail_span, tail_quota, NW * 32, tail_long};
    const int64_t u = tail_run_start(geom, run);
    const int mine = run < tail_long ? tail_quota : tail_quota - 1;
    const int64_t tt = u / tail_span;
    const int64_t rest = tail_units - u;
    tile = full_tiles + (int)tt;
    st = (int)(u - tt * tail_span);
    left = rest < mine ? (int)rest : mine;
    if (tail_span != stages_total && left > stages_total - st) left = stages_total - st;
    piece = run - tail_first_run(geom, tt);
  }
  const int64_t split_row0 = (int64_t)full_tiles * (NW * 32);
  const int n_rows = (int)N;
  // The codebook mean travels through a slot of LDS private to the wave (512 bytes behind the
  // stage buffers): its load is the first one issued, so it has landed long before the rows, and
  // nothing issued after the stage loads has to be waited for before the first MFMA.
  // It goes global -> LDS directly (lanes 0..31 carry its 512 bytes), through no register.
  float* mu_lds = reinterpret_cast<float*>(smem + 2 * Stage<SK>::kBytes) + wave * kD;
  // (the doubled mean, stored behind the padded norms)
  const float* __restrict__ mean2 = cnp + (K + kStageK - 1) / kStageK * kStageK;
  if (lane < 32) glds16(mean2 + lane * 4, reinterpret_cast<char*>(mu_lds));

  int sti = st;            // codebook stage of the next unit to issue
  int to_issue = left;     // units whose stage has not been issued yet
  StageOffsets<NW, SK> offs;
  stage_offsets<NW, SK>(offs, wave, lane);
  auto issue_next = [&](char* sb) {
    issue_stage<NW, SK>(C, K, cnp, sti, sb, wave, lane, offs);
    sti = sti + 1 == stages_total ? 0 : sti + 1;
    --to_issue;
  };
  const float cn_max = aux[kAuxCnMax], cr2_max = aux[kAuxCr2Max], mean_norm = aux[kAuxMeanNorm];
  int buf = 0;
  bool first = true;
  for (;;) {
    // descriptors of this wave, 64 components per lane
    f32x4 xv[16];
    const int row = tile * (NW * 32) + wave * 32 + j;
    {
      const int r = row < n_rows ? row : n_rows - 1;
      const f32x4* p = reinterpret_cast<const f32x4*>(X + (int64_t)r * kD + h * 4);
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = p[2 * q];
    }
    // loads return in order.  First row tile: rows, then the two stages (at least kMinLoads loads
    // per wave and stage): the rows are complete once at most that many loads as were issued
    // behind them are outstanding, and centring them overlaps the stages' flight.  Later row
    // tiles: the rows' loads are the last ones issued.
    constexpr int kMinLoads = Stage<SK>::kGlds / NW;
    bool two = false;
    if (first) {
      issue_next(smem);
      two = to_issue > 0;
      if (two) {
        issue_next(smem + Stage<SK>::kBytes);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * kMinLoads) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMinLoads) : "memory");
      }
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (first) IC_TRACE_NOW(4);
    // centre the rows on the codebook mean and scale by -2: fma(-2, x, 2 m) rounds once, exactly
    // like x - m, and doubling is exact; on the way this lane's half of |2 (x - m)|^2
    float xr[1][64];
    float c0[1];
    {
      // opaque offset: the mean is read from LDS again for every row tile instead of being kept
      // in 64 registers across the sweep, which the MFMA operands need
      int moff = h * 16;
      asm volatile("" : "+v"(moff));
      const f32x4* mp = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(mu_lds) + moff);
      f32x2 sq = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        centre_chunk(xv[q], mp[2 * q], &xr[0][q * 4], sq);
        // one chunk after the other: scheduled freely, the sixteen chunks' values are all alive
        // at once and a few registers spill to scratch memory
        __builtin_amdgcn_sched_barrier(0);
      }
      float xn4 = sq.x + sq.y;
      xn4 += __shfl_xor(xn4, 32);
      const float xn = 0.25f * xn4;
      // the band's constant part needs R = |x - m|^2 + 2 (x - m).m (what separates the computed
      // value from the squared distance); (x - m).m <= |x - m| |m| spares the second sum per
      // component, and a larger band only ranks a little more
      const float bound = error_bound(xn, cn_max, cr2_max);
      const float r_up = xn + 2.000001f * __builtin_amdgcn_sqrtf(xn) * 1.000001f * mean_norm;
      c0[0] = 2.01f * bound + kBandSlope * r_up;
    }
    Top2 best[1];
    float gate[1];
    {
      // with a caller-supplied bound only candidates below it are ranked at all (index -1 =
      // none); the bound then stands for every candidate that was never looked at
      float start = kNone;
      if (thr) start = thr[row < n_rows ? row : n_rows - 1];
      best[0].d1 = start;
      best[0].d2 = start;
      best[0].d3 = start;
      best[0].d4 = start;
      best[0].i1 = -1;
      best[0].i2 = -1;
      best[0].i3 = -1;
      gate[0] = start;
    }

    if (first) IC_TRACE_NOW(5);
    if (first) {
      // the first stage has landed once at most the second one's loads are outstanding
      if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMinLoads) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      first = false;
      IC_TRACE_NOW(1);
    } else {
      IC_TRACE_NOW(5);
    }

    // this row tile's stages [st, st + item)
    const int item = stages_total - st < left ? stages_total - st : left;
    f32x16 acc[1];
    for (int i = 0; i < item; ++i, ++st) {
      char* sb = smem + buf * Stage<SK>::kBytes;
#pragma unroll
      for (int t = 0; t < Stage<SK>::kTiles; ++t) {
        // tiles that lie wholly past the end of the codebook (padding of the last stage) are skipped
        const bool live = (st * Stage<SK>::kTiles + t) * kTileK < K;
        if (live)
          compute_tile<1>(sb + t * kTileK * kRowBytes, sb + Stage<SK>::kTileBytes + t * kTileK * 4,
                          xr, acc, j, h);
        if (t == Stage<SK>::kTiles - 1) {
          // every wave is done reading this buffer once it passes the barrier; the stage issued
          // one iteration ago has landed in the other buffer by then
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if (to_issue > 0) issue_next(sb);
        }
        if (live) {
          const int tile_no = st * Stage<SK>::kTiles + t;
          // the first kDenseTiles tiles of a scan are ranked without per-register tests
          const bool dense = i * Stage<SK>::kTiles + t < kDenseTiles;
          if constexpr (GATED) tile_epilogue_gated<1, MODE2, SK == 64>(acc, best, gate, c0, tile_no, dense);
          else tile_epilogue_all<1>(acc, best, tile_no);
        }
      }
      buf ^= 1;
    }
    left -= item;

    // (tile, key) -> codeword index, and store this lane's list
    {
      const Top2& b = best[0];
      if (row < n_rows) {
        // C is the codebook in scan order when `perm` is given (scan position -> codeword
        // index): the lists leave the kernel in codeword indices, so that everything that
        // follows -- merging, direct evaluation, ties by (distance, index) -- never sees positions
        auto named = [&](int pos) { return (perm && pos >= 0) ? perm[pos] : pos; };
        ic_top2_rec rec;
        rec.d1 = b.d1; rec.i1 = named(decode_candidate(b.i1, b.d1, h, K));
        rec.d2 = b.d2; rec.i2 = named(decode_candidate(b.i2, b.d2, h, K));
        const int64_t slot = part_slot(2 * piece + h, row, N, split_row0);
        pb.rec[slot] = rec;
        pb.d3[slot] = b.d3;
        if constexpr (MODE2) {
          pb.i3[slot] = named(decode_candidate(b.i3, b.d3, h, K));
          pb.d4[slot] = b.d4;
        }
        // the constant part of the row's band, for k_rescore's shortcut (same for every list)
        if (rowband && piece == 0 && h == 0) rowband[row] = c0[0];
      }
    }
    if (left <= 0) break;
    // on into the next row tile
    IC_TRACE_NOW(4);
    ++tile;
    st = 0;
    piece = 0;
  }
  IC_TRACE_NOW(2);
  IC_TRACE_PUT(7, (unsigned long long)clock64());
}

#ifdef IC_TRACE
extern "C" int ic_trace_read(void* host_dst, size_t bytes) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(ic_trace_buf), bytes, 0,
                                  hipMemcpyDeviceToHost);
}
#endif

// ---------------------------------------------------------------------------------------------
// Codebooks of at most 256 words (k-means hashing tables, small vocabularies): the padded image of
// the WHOLE codebook is 135 KB and stays in LDS, in the layout of two 128-word stages, for the
// life of a workgroup.  One workgroup of 16 waves per CU loads it once; after that single barrier
// every wave is on its own: it takes tiles of 32 rows (tile = it * 16 G + slot, G workgroups),
// scans the codebook out of LDS, writes its lists and goes on.  There are no stage loads and no
// barriers inside the walk, so the four waves of a SIMD drift apart and one wave's ranking and
// row loads fall into the others' multiplications.  The rows of a wave's NEXT tile are loaded
// during the last codeword tile of the current one: each 16-byte chunk of the row registers is
// overwritten by the load of the next rows' chunk right behind the four multiplications that
// read it last, so the loads fly while the rest of the tile multiplies and is ranked.
// Slots are numbered CU first, then SIMD, then wave within the SIMD (wave w of a workgroup sits
// on SIMD w % 4), so the tiles of the last, partly filled round spread over all CUs and SIMDs.
// Every tile of such a scan is one of its first: all candidates are inserted without tests.
template <bool MODE2>
__global__ __launch_bounds__(1024, 4) void k_top2_resident(
    const float* __restrict__ X, int64_t N, const float* __restrict__ C, int K,
    const float* __restrict__ aux, const float* __restrict__ thr, PartBufs pb,
    float* __restrict__ rowband, uint32_t* __restrict__ zero_word) {
  constexpr int NW = 16, SK = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  const float* __restrict__ cnp = aux + kAuxNorms;
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;
  const int n_rows = (int)N;
  const int n_tiles = (n_rows + 31) >> 5;          // tiles of 32 rows
  const int G = (int)gridDim.x;
  const int stride = NW * G;
  int wt = (wave >> 2) * (4 * G) + (wave & 3) * G + (int)blockIdx.x;

  float* mu_lds = reinterpret_cast<float*>(smem + 2 * Stage<SK>::kBytes) + wave * kD;
  const float* __restrict__ mean2 = cnp + (K + kStageK - 1) / kStageK * kStageK;
  if (lane < 32) glds16(mean2 + lane * 4, reinterpret_cast<char*>(mu_lds));
  {
    StageOffsets<NW, SK> offs;
    stage_offsets<NW, SK>(offs, wave, lane);
    issue_stage<NW, SK>(C, K, cnp, 0, smem, wave, lane, offs);
    if (K > SK) issue_stage<NW, SK>(C, K, cnp, 1, smem + Stage<SK>::kBytes, wave, lane, offs);
  }

  // rows of a tile, 64 components per lane: raw as loaded, then centred in place
  float xr[1][64];
  auto rows_of = [&](int t) {
    int r = t * 32 + j;
    r = r < n_rows ? r : n_rows - 1;
    return reinterpret_cast<const f32x4*>(X + (int64_t)r * kD + h * 4);
  };
  {
    const f32x4* p = rows_of(wt < n_tiles ? wt : n_tiles - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const f32x4 v = p[2 * q];
#pragma unroll
      for (int e = 0; e < 4; ++e) xr[0][q * 4 + e] = v[e];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // The four waves of a SIMD leave the barrier together and would rank, wait for LDS and load
  // rows at the same moments; they are started up to 10 us apart instead (the matrix pipe is
  // busy with the earlier ones meanwhile).  Worth 0.3 ... 0.7 points at 1 M rows.  The index is
  // 0..3 within a SIMD whether the hardware deals waves to SIMDs round-robin or in fours.
  for (int i = ((wave + (wave >> 2)) & 3) * 16; i > 0; --i) __builtin_amdgcn_s_sleep(8);

  const float cn_max = aux[kAuxCnMax], cr2_max = aux[kAuxCr2Max], mean_norm = aux[kAuxMeanNorm];
  const int k_tiles = (K + kTileK - 1) / kTileK;
  for (; wt < n_tiles; wt += stride) {
    const int row = wt * 32 + j;
    float c0[1];
    {
      int moff = h * 16;
      asm volatile("" : "+v"(moff));
      const f32x4* mp = reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(mu_lds) + moff);
      f32x2 sq = {0.f, 0.f};
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4 raw = {xr[0][q * 4], xr[0][q * 4 + 1], xr[0][q * 4 + 2], xr[0][q * 4 + 3]};
        centre_chunk(raw, mp[2 * q], &xr[0][q * 4], sq);
        __builtin_amdgcn_sched_barrier(0);
      }
      float xn4 = sq.x + sq.y;
      xn4 += __shfl_xor(xn4, 32);
      const float xn = 0.25f * xn4;
      const float bound = error_bound(xn, cn_max, cr2_max);
      const float r_up = xn + 2.000001f * __builtin_amdgcn_sqrtf(xn) * 1.000001f * mean_norm;
      c0[0] = 2.01f * bound + kBandSlope * r_up;
    }
    Top2 best[1];
    float gate[1];
    {
      float start = kNone;
      if (thr) start = thr[row < n_rows ? row : n_rows - 1];
      best[0].d1 = start;
      best[0].d2 = start;
      best[0].d3 = start;
      best[0].d4 = start;
      best[0].i1 = -1;
      best[0].i2 = -1;
      best[0].i3 = -1;
      gate[0] = start;
    }
    f32x16 acc[1];
    auto tile_at = [&](int t) {
      return smem + (t >> 2) * Stage<SK>::kBytes + (t & 3) * (kTileK * kRowBytes);
    };
    auto norms_at = [&](int t) {
      return smem + (t >> 2) * Stage<SK>::kBytes + Stage<SK>::kTileBytes + (t & 3) * (kTileK * 4);
    };
    for (int t = 0; t + 1 < k_tiles; ++t) {
      compute_tile<1>(tile_at(t), norms_at(t), xr, acc, j, h);
      tile_epilogue_gated<1, MODE2, true>(acc, best, gate, c0, t, true);
    }
    // (a wave without a next tile loads the last rows again: 16 KB out of L2, once)
    const int nxt = wt + stride;
    const f32x4* pn = rows_of(nxt < n_tiles ? nxt : n_tiles - 1);
    if constexpr (MODE2) {
      // (the longer lists of a top-2 search leave no registers for the overlapped reload below:
      // the next rows are loaded once this tile's lists are stored)
      const int t = k_tiles - 1;
      compute_tile<1>(tile_at(t), norms_at(t), xr, acc, j, h);
      tile_epilogue_gated<1, MODE2, true>(acc, best, gate, c0, t, true);
    } else {
      // the last codeword tile, with the next tile's rows loaded behind the multiplications
      const int t = k_tiles - 1;
      const char* tile_base = tile_at(t);
      const f32x4* cnv = reinterpret_cast<const f32x4*>(norms_at(t) + h * 16);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 c4 = cnv[g * 2];
        acc[0][g * 4 + 0] = c4[0];
        acc[0][g * 4 + 1] = c4[1];
        acc[0][g * 4 + 2] = c4[2];
        acc[0][g * 4 + 3] = c4[3];
      }
      const f32x4* arow = reinterpret_cast<const f32x4*>(tile_base + j * kRowBytes + h * 16);
      f32x4 a = arow[0];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        f32x4 an = a;
        if (q + 1 < 16) an = arow[2 * (q + 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xr[0][q * 4 + e], acc[0], 0, 0, 0);
        const f32x4 v = pn[2 * q];
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[0][q * 4 + e] = v[e];
        a = an;
      }
      tile_epilogue_gated<1, MODE2, true>(acc, best, gate, c0, t, true);
    }
    {
      const Top2& b = best[0];
      if (row < n_rows) {
        ic_top2_rec rec;
        rec.d1 = b.d1; rec.i1 = decode_candidate(b.i1, b.d1, h, K);
        rec.d2 = b.d2; rec.i2 = decode_candidate(b.i2, b.d2, h, K);
        const int64_t slot = (int64_t)h * N + row;   // part_slot of a row tile scanned whole
        pb.rec[slot] = rec;
        pb.d3[slot] = b.d3;
        if constexpr (MODE2) {
          pb.i3[slot] = decode_candidate(b.i3, b.d3, h, K);
          pb.d4[slot] = b.d4;
        }
        if (rowband && h == 0) rowband[row] = c0[0];
      }
    }
    if constexpr (MODE2) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const f32x4 v = pn[2 * q];
#pragma unroll
        for (int e = 0; e < 4; ++e) xr[0][q * 4 + e] = v[e];
      }
    }
  }
}

static hipError_t launch_resident(const float* x, int64_t n, const float* c, int64_t k,
                                  const float* aux, const AssignPlan& plan, bool mode2,
                                  const float* thr, const PartBufs& pb, float* rowband,
                                  uint32_t* zero_word, hipStream_t s) {
  const int64_t wave_tiles = (n + 31) / 32;
  // the kernel's geometry must be the plan's: k_rescore reads the lists by it
  if (k < 1 || k > kResidentMaxK || plan.waves != 16 || plan.stage_k != 128 ||
      plan.rows_per_wg != 512 || plan.stages != (k + 127) / 128 || plan.max_pieces != 1 ||
      plan.tail_wgs != 0 || plan.full_tiles != plan.row_tiles ||
      plan.full_tiles * 512 < n || plan.resident_wgs < 1 ||
      (int64_t)plan.resident_wgs * 4 > wave_tiles + 3)
    return hipErrorInvalidValue;
  auto kern = mode2 ? k_top2_resident<true> : k_top2_resident<false>;
  const int lds = 2 * Stage<128>::kBytes + 16 * kD * 4;
  hipLaunchKernelGGL(kern, dim3((unsigned)plan.resident_wgs), dim3(1024), lds, s, x, n, c, (int)k,
                     aux, thr, pb, rowband, zero_word);
  return hipGetLastError();
}

// IC_AMD_VARIANT selects the ranking variant for A/B timing in one build: 0 = rank every
// candidate, anything else (default) = the two-level gate.
static bool gated_variant() {
  static const int variant = [] {
    const char* v = getenv("IC_AMD_VARIANT");
    return v ? atoi(v) : 2;
  }();
  return variant != 0;
}

template <int NW, int SK>
static hipError_t launch_geometry(const float* x, int64_t n, const float* c, int64_t k,
                                  const float* aux, const AssignPlan& plan, bool mode2,
                                  const float* thr, const PartBufs& pb, float* rowband,
                                  uint32_t* zero_word, const int32_t* perm, hipStream_t s) {
  const int64_t stages = (k + SK - 1) / SK;
  const int64_t tail_tiles = plan.row_tiles - plan.full_tiles;
  auto kern = !gated_variant() ? (mode2 ? k_top2_partial<kXT, NW, SK, false, true>
                                        : k_top2_partial<kXT, NW, SK, false, false>)
              : mode2 ? k_top2_partial<kXT, NW, SK, true, true>
                      : k_top2_partial<kXT, NW, SK, true, false>;
  const int lds = 2 * Stage<SK>::kBytes + NW * kD * 4;   // allowed by assign_init() on this device
  const int64_t grid = plan.full_tiles + plan.tail_wgs;
  // the kernel's geometry must be the plan's: it indexes rows and records by it
  if (stages != plan.stages || plan.rows_per_wg != NW * 32 || plan.tail_quota < 1 ||
      plan.tail_span < stages || (plan.tail_span != stages && plan.tail_span % plan.tail_quota != 0))
    return hipErrorInvalidValue;
  if (plan.tail_long >= plan.tail_wgs) {   // every run has tail_quota stages
    if (plan.tail_wgs != (tail_tiles * plan.tail_span + plan.tail_quota - 1) / plan.tail_quota)
      return hipErrorInvalidValue;
  } else if (plan.tail_long < 1 || plan.tail_quota < 2 || plan.tail_span != stages ||
             plan.tail_long * plan.tail_quota +
                     (plan.tail_wgs - plan.tail_long) * (plan.tail_quota - 1) !=
                 tail_tiles * plan.tail_span) {
    return hipErrorInvalidValue;
  }
  if (grid <= 0 || grid > 0x7fffffffLL || plan.full_tiles > 0x7fffffffLL) return hipErrorInvalidValue;
  TailOrder order = {};
  order.on = plan.tail_ordered;
  if (order.on) {
    if (plan.tail_wgs > kMaxTailOrder) return hipErrorInvalidValue;
    for (int64_t b = 0; b < plan.tail_wgs; ++b) {
      if (plan.tail_order[b] >= plan.tail_wgs) return hipErrorInvalidValue;
      order.run[b] = plan.tail_order[b];
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, x, n, c, (int)k, aux,
                     (int)stages, plan.tail_span, (int)plan.full_tiles,
                     tail_tiles * plan.tail_span, plan.tail_quota,
                     (int)(plan.tail_long < 0x7fffffff ? plan.tail_long : 0x7fffffff), thr, pb,
                     rowband, zero_word, perm, order);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The search for rows wider than 128 columns (D = 128 nslab, up to kMaxWideD): projected face
// signatures and other long vectors (vision/facerec/face_indexing.py:141, :158 run vq / kmeans2
// on them).  The geometry is the 8-wave one -- 32 rows per wave, stages of 64 codewords -- but a
// stage now holds ONE 128-column slab of its 64 codewords, and the walk over a chunk of 64
// codewords is a loop over the slabs: the wave loads that slab of its rows (they no longer fit in
// registers for the whole sweep) from a copy that k_centre_rows has already centred on the mean
// and scaled by -2 -- so the loaded registers are the MFMA operand as they are -- and the two
// tiles' accumulators, started from |c - m|^2 at slab 0, carry the MFMA chain from slab to slab; after the last
// slab they hold |c - m|^2 - 2 (x - m).c over all D columns and are ranked as in the 128-column
// kernel.  Rounding bound and bands are those of a D-term chain (error_bound_d).  The row slabs
// are re-read once per chunk of 64 codewords (32 FLOP per byte): this path is for correctness at
// any width first; at D = 2048 and beyond a few hundred thousand rows it is bound by those reads.
__device__ __forceinline__ void acc_from_norms(const char* cn_base, f32x16& acc, int h) {
  const f32x4* cnv = reinterpret_cast<const f32x4*>(cn_base + h * 16);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 c4 = cnv[g * 2];
    acc[g * 4 + 0] = c4[0];
    acc[g * 4 + 1] = c4[1];
    acc[g * 4 + 2] = c4[2];
    acc[g * 4 + 3] = c4[3];
  }
}

__device__ __forceinline__ void mfma_tile(const char* tile_base, const float (&xr)[64], f32x16& acc,
                                          int i, int h) {
  const f32x4* arow = reinterpret_cast<const f32x4*>(tile_base + i * kRowBytes + h * 16);
  f32x4 a = arow[0];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    f32x4 an = a;
    if (q + 1 < 16) an = arow[2 * (q + 1)];
#pragma unroll
    for (int e = 0; e < 4; ++e)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], xr[q * 4 + e], acc, 0, 0, 0);
    a = an;
  }
}

// stage = slab `sl` (128 columns) of the 64 codewords of chunk `ch`; rows of C are D floats apart
__device__ __forceinline__ void issue_stage_wide(const float* __restrict__ C, int K, int D,
                                                 const float* __restrict__ cnp, int ch, int sl,
                                                 char* sb, int wave, int lane) {
  constexpr int NW = 8, SK = 64;
  const char* base = reinterpret_cast<const char*>(C) + ((size_t)ch * SK * D + (size_t)sl * kD) * 4;
  const int last = K - 1 - ch * SK;
  const unsigned row_bytes = (unsigned)D * 4u;
  asm volatile("" : "+v"(lane));
#pragma unroll
  for (int i = 0; i < (Stage<SK>::kGlds + NW - 1) / NW; ++i) {
    const int m = wave + NW * i;
    if (m < Stage<SK>::kGlds) {
      const int pos = m * 64 + lane;
      int r = (pos * 1986) >> 16;  // pos / 33, exact for pos < 4224
      int c = pos - r * kRowChunks;
      c = c == kRowChunks - 1 ? 0 : c;
      r = r < last ? r : last;
      const unsigned voff = (unsigned)r * row_bytes + (unsigned)c * 16u;
      glds16(reinterpret_cast<const float*>(base + voff), sb + m * 1024);
    }
  }
  if (wave == NW - 1) glds4(cnp + (size_t)ch * SK + lane, sb + Stage<SK>::kTileBytes);
}

__device__ __forceinline__ float group16_sum(float v);

// out[row] = -2 (x[row] - m), c0[row] = the constant part of the row's band (see k_top2_partial):
// 16 lanes per row, 128 columns after 128 columns.
__global__ __launch_bounds__(256) void k_centre_rows(const float* __restrict__ X, int64_t N, int D,
                                                     const float* __restrict__ aux,
                                                     const float* __restrict__ mean,
                                                     float* __restrict__ out,
                                                     float* __restrict__ c0) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  const float cn_max = aux[kAuxCnMax], cr2_max = aux[kAuxCr2Max];
  const float band_slope = 4.f * gamma_direct_d(D);
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < N; row += stride) {
    float xn = 0.f, xm = 0.f;
    for (int sl = 0; sl < D; sl += kD) {
      const int64_t at = row * D + sl + sub * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float mu = mean[sl + sub * 8 + e];
        const float t = X[at + e] - mu;
        xn = __builtin_fmaf(t, t, xn);
        xm = __builtin_fmaf(t, mu, xm);
        out[at + e] = -2.0f * t;
      }
    }
    xn = group16_sum(xn);
    xm = group16_sum(xm);
    if (sub == 0)
      c0[row] = 2.01f * error_bound_d(xn, cn_max, cr2_max, D) + band_slope * (xn + 2.f * xm);
  }
}

// XC: the rows centred and scaled by k_centre_rows; rowc0 [N]: their band constants.
template <bool GATED, bool MODE2>
__global__ __launch_bounds__(512, 4) void k_top2_wide(
    const float* __restrict__ XC, int64_t N, const float* __restrict__ C, int K,
    const float* __restrict__ aux, int D, int stages_total, int full_tiles, int64_t tail_units,
    int tail_quota, int pieces, const float* __restrict__ thr, PartBufs pb,
    const float* __restrict__ rowc0, uint32_t* __restrict__ zero_word) {
  constexpr int NW = 8, SK = 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;
  const int h = lane >> 5;
  const int nslab = D / kD;
  const float* __restrict__ cnp = aux + kAuxNorms;
  if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0;

  // the walk over (row tile, chunk of 64 codewords) units, as in k_top2_partial
  int tile = blockIdx.x;
  int st = 0, left = stages_total, piece = 0;
  if (pieces > 0) {
    // piece `piece` of row tile `tile`: chunks [piece q, piece q + q); the pieces of a row tile
    // sit at consecutive places of one XCD (workgroups are dealt to the 8 XCDs in turn)
    const int xcd = blockIdx.x & 7, place = blockIdx.x >> 3;
    tile = place / pieces * 8 + xcd;
    piece = place - place / pieces * pieces;
    st = piece * tail_quota;
    left = stages_total - st < tail_quota ? stages_total - st : tail_quota;
    if ((int64_t)tile * (NW * 32) >= N) return;   // the grid is rounded up to 8 row tiles
  } else if ((int)blockIdx.x >= full_tiles) {
    const int64_t u = (int64_t)((int)blockIdx.x - full_tiles) * tail_quota;
    const int64_t tt = u / stages_total;
    const int64_t rest = tail_units - u;
    tile = full_tiles + (int)tt;
    st = (int)(u - tt * stages_total);
    left = rest < tail_quota ? (int)rest : tail_quota;
    piece = ((int)blockIdx.x - full_tiles) - (int)(tt * stages_total / tail_quota);
  }
  const int64_t split_row0 = (int64_t)full_tiles * (NW * 32);

  // LDS stages follow each other as (chunk, slab 0 .. nslab-1), chunk after chunk
  int ich = st, isl = 0;
  int to_issue = left * nslab;
  auto issue_next = [&](char* sb) {
    issue_stage_wide(C, K, D, cnp, ich, isl, sb, wave, lane);
    if (++isl == nslab) {
      isl = 0;
      ich = ich + 1 == stages_total ? 0 : ich + 1;
    }
    --to_issue;
  };
  issue_next(smem);
  if (to_issue > 0) issue_next(smem + Stage<SK>::kBytes);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const float band_slope = 4.f * gamma_direct_d(D);
  int buf = 0;
  for (;;) {
    const int row = tile * (NW * 32) + wave * 32 + j;
    const int r = row < (int)N ? row : (int)N - 1;
    const float* xrow = XC + (int64_t)r * D + h * 4;
    Top2 best[1];
    float gate[1];
    float c0[1] = {rowc0[r]};
    {
      float start = kNone;
      if (thr) start = thr[r];
      best[0].d1 = start;
      best[0].d2 = start;
      best[0].d3 = start;
      best[0].d4 = start;
      best[0].i1 = -1;
      best[0].i2 = -1;
      best[0].i3 = -1;
      gate[0] = start;
    }
    const int item = stages_total - st < left ? stages_total - st : left;
#pragma unroll 1
    for (int i = 0; i < item; ++i, ++st) {
      f32x16 acc0, acc1;
#pragma unroll 1
      for (int sl = 0; sl < nslab; ++sl) {
        char* sb = smem + buf * Stage<SK>::kBytes;
        // this slab of the wave's rows: already -2 (x - m), the MFMA operand as it is
        float xr[64];
        {
          const f32x4* p = reinterpret_cast<const f32x4*>(xrow + sl * kD);
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const f32x4 v = p[2 * q];
            xr[q * 4 + 0] = v[0];
            xr[q * 4 + 1] = v[1];
            xr[q * 4 + 2] = v[2];
            xr[q * 4 + 3] = v[3];
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const bool live = (st * 2 + t) * kTileK < K;
          if (live) {
            f32x16& acc = t == 0 ? acc0 : acc1;
            if (sl == 0) acc_from_norms(sb + Stage<SK>::kTileBytes + t * kTileK * 4, acc, h);
            mfma_tile(sb + t * kTileK * kRowBytes, xr, acc, j, h);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (to_issue > 0) issue_next(sb);
        buf ^= 1;
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if ((st * 2 + t) * kTileK < K) {
          if constexpr (GATED)
            tile_epilogue_gated1<MODE2, true>(t == 0 ? acc0 : acc1, best[0], gate[0], c0[0], st * 2 + t,
                                        1.f + band_slope);
          else
            tile_epilogue_all1(t == 0 ? acc0 : acc1, best[0], st * 2 + t);
        }
      }
    }
    left -= item;
    {
      const Top2& b = best[0];
      if (row < N) {
        ic_top2_rec rec;
        rec.d1 = b.d1; rec.i1 = decode_candidate(b.i1, b.d1, h, K);
        rec.d2 = b.d2; rec.i2 = decode_candidate(b.i2, b.d2, h, K);
        const int64_t slot = part_slot(2 * piece + h, row, N, split_row0);
        pb.rec[slot] = rec;
        pb.d3[slot] = b.d3;
        if constexpr (MODE2) {
          pb.i3[slot] = decode_candidate(b.i3, b.d3, h, K);
          pb.d4[slot] = b.d4;
        }
      }
    }
    if (left <= 0) break;
    ++tile;
    st = 0;
    piece = 0;
  }
}

// xc [n][D] and rowc0 [n] are workspace (launch_top2_partial's `wide_ws`): the centred copy of
// the rows and their band constants.
static hipError_t launch_wide(const float* x, int64_t n, const float* c, int64_t k,
                              const float* aux, const float* mean, int D, const AssignPlan& plan,
                              bool mode2, const float* thr, const PartBufs& pb, float* xc,
                              float* rowc0, uint32_t* zero_word, hipStream_t s) {
  const int64_t stages = (k + 63) / 64;
  const int64_t tail_tiles = plan.row_tiles - plan.full_tiles;
  const int64_t P = plan.wide_pieces, q = plan.tail_quota;
  if (plan.waves != 8 || plan.stage_k != 64 || stages != plan.stages || q < 1 ||
      D <= kD || D % kD != 0 || D > kMaxWideD || !xc || !rowc0)
    return hipErrorInvalidValue;
  int64_t grid;
  if (P > 0) {
    // the kernel and k_rescore index the records by this geometry: P pieces of q chunks per row
    // tile, none of them empty
    if (plan.full_tiles != 0 || plan.tail_span != P * q || plan.max_pieces != P ||
        P * q < stages || (P - 1) * q >= stages || plan.tail_wgs != plan.row_tiles * P)
      return hipErrorInvalidValue;
    grid = (plan.row_tiles + 7) / 8 * 8 * P;
  } else {
    if (plan.tail_span != stages || plan.tail_wgs != (tail_tiles * stages + q - 1) / q)
      return hipErrorInvalidValue;
    grid = plan.full_tiles + plan.tail_wgs;
  }
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  int64_t cgrid = (n + 15) / 16;
  if (cgrid > 16384) cgrid = 16384;
  hipLaunchKernelGGL(k_centre_rows, dim3((unsigned)cgrid), dim3(256), 0, s, x, n, D, aux, mean, xc,
                     rowc0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  auto kern = !gated_variant() ? (mode2 ? k_top2_wide<false, true> : k_top2_wide<false, false>)
              : mode2 ? k_top2_wide<true, true>
                      : k_top2_wide<true, false>;
  const int lds = 2 * Stage<64>::kBytes;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), lds, s, xc, n, c, (int)k, aux, D,
                     (int)stages, (int)plan.full_tiles, tail_tiles * stages, plan.tail_quota, (int)P,
                     thr, pb, rowc0, zero_word);
  return hipGetLastError();
}

static hipError_t allow_lds_wide() {
  const int lds = 2 * Stage<64>::kBytes;
  const void* kerns[4] = {reinterpret_cast<const void*>(k_top2_wide<false, false>),
                          reinterpret_cast<const void*>(k_top2_wide<false, true>),
                          reinterpret_cast<const void*>(k_top2_wide<true, false>),
                          reinterpret_cast<const void*>(k_top2_wide<true, true>)};
  for (const void* kp : kerns) {
    hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t exact_init();

template <int NW, int SK>
static hipError_t allow_lds() {
  const int lds = 2 * Stage<SK>::kBytes + NW * kD * 4;   // two stage buffers + a mean vector per wave
  const void* kerns[4] = {reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, false, false>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, false, true>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, true, false>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, true, true>)};
  for (const void* kp : kerns) {
    hipError_t e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Once per device (from ic_create, with that device current): the search kernels use more dynamic
// LDS than a launch may ask for by default.  Nothing is configured at launch time.
hipError_t assign_init() {
  hipError_t e = allow_lds<8, 64>();
  if (e != hipSuccess) return e;
  e = allow_lds<16, 128>();
  if (e != hipSuccess) return e;
  e = allow_lds_wide();
  if (e != hipSuccess) return e;
  for (const void* kp : {reinterpret_cast<const void*>(k_top2_resident<false>),
                         reinterpret_cast<const void*>(k_top2_resident<true>)}) {
    e = hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * Stage<128>::kBytes + 16 * kD * 4);
    if (e != hipSuccess) return e;
  }
  return exact_init();
}

int64_t assign_part_records(const AssignPlan& p, int64_t n) {
  int64_t split_row0 = p.full_tiles * (int64_t)p.rows_per_wg;
  if (split_row0 > n) split_row0 = n;
  return 2 * n + (int64_t)(2 * p.max_pieces - 2) * (n - split_row0);   // see part_slot
}

hipError_t launch_top2_partial(const float* x, int64_t n, const float* c, int64_t k,
                               const float* aux, const float* mean, int D, const AssignPlan& plan,
                               bool mode2, const float* thr, const PartBufs& pb, float* rowband,
                               float* wide_xc, uint32_t* zero_word, const int32_t* perm,
                               hipStream_t s) {
  if (!pb.rec || !pb.d3 || (mode2 && (!pb.i3 || !pb.d4))) return hipErrorInvalidValue;
  if (D > kD && perm) return hipErrorInvalidValue;   // wide codebooks are scanned as stored
  if (D > kD)   // rowband is always filled for wide rows: it is where the kernel reads the bands
    return launch_wide(x, n, c, k, aux, mean, D, plan, mode2, thr, pb, wide_xc, rowband,
                       zero_word, s);
  if (D != kD || mean != aux) return hipErrorInvalidValue;
  if (plan.resident) {
    if (perm) return hipErrorInvalidValue;   // such codebooks are never reordered
    return launch_resident(x, n, c, k, aux, plan, mode2, thr, pb, rowband, zero_word, s);
  }
  if (plan.waves == 8 && plan.stage_k == 64)
    return launch_geometry<8, 64>(x, n, c, k, aux, plan, mode2, thr, pb, rowband, zero_word, perm, s);
  if (plan.waves == 16 && plan.stage_k == 128)
    return launch_geometry<16, 128>(x, n, c, k, aux, plan, mode2, thr, pb, rowband, zero_word, perm, s);
  return hipErrorInvalidValue;
}

// ---------------------------------------------------------------------------------------------
// 16 lanes per row: 8 components per lane, butterfly over the 16 lanes (every lane ends with the
// same total, fixed order => deterministic).
__device__ __forceinline__ float group16_sum(float v) {
  v += __shfl_xor(v, 8, 16);
  v += __shfl_xor(v, 4, 16);
  v += __shfl_xor(v, 2, 16);
  v += __shfl_xor(v, 1, 16);
  return v;
}

// The direct squared distance sum (x - c)^2 in float32, in ONE summation order shared by
// k_rescore (16 lanes per pair) and k_exact_top2 (one thread per codeword):
//   p_s  = fma chain over components 8 s .. 8 s + 7, from 0            (s = 0 .. 15)
//   b_q  = (p_4q + p_4q+1) + (p_4q+2 + p_4q+3)                         (q = 0 .. 3)
//   sum  = ((b_0 + b_1) + b_2) + b_3
// Float addition commutes bit for bit, so two xor-butterfly steps (partners 1, 2) leave b_q in
// all four lanes of quad q, whatever the lane.  Both kernels returning identical bits for
// identical pairs is what makes the fast path agree with the all-direct search on every row, not
// only on most.
__device__ __forceinline__ float chain8(const f32x4& x0, const f32x4& x1, const f32x4& c0,
                                        const f32x4& c1) {
  float s = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = x0[e] - c0[e];
    s = __builtin_fmaf(t, t, s);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float t = x1[e] - c1[e];
    s = __builtin_fmaf(t, t, s);
  }
  return s;
}

__device__ __forceinline__ float group16_tree(float p) {
  p += __shfl_xor(p, 1, 16);
  p += __shfl_xor(p, 2, 16);
  const float b0 = __shfl(p, 0, 16), b1 = __shfl(p, 4, 16);
  const float b2 = __shfl(p, 8, 16), b3 = __shfl(p, 12, 16);
  return ((b0 + b1) + b2) + b3;
}

// The same for one 128-column slab of a wider row: the slab's four 32-component blocks are added
// to `acc` one after the other, which is the order in which k_exact_top2 adds them.  With
// acc = 0 the result equals group16_tree(p) bit for bit (0 + b0 is b0).
__device__ __forceinline__ float group16_chain(float p, float acc) {
  p += __shfl_xor(p, 1, 16);
  p += __shfl_xor(p, 2, 16);
  const float b0 = __shfl(p, 0, 16), b1 = __shfl(p, 4, 16);
  const float b2 = __shfl(p, 8, 16), b3 = __shfl(p, 12, 16);
  return (((acc + b0) + b1) + b2) + b3;
}

// 32 components of one (row, codeword) pair: xs points at the row's components in LDS, cv holds
// the codeword's (8 float4).
__device__ __forceinline__ float block32(const float* xs, const f32x4 (&cv)[8]) {
  const f32x4* xv = reinterpret_cast<const f32x4*>(xs);
  const float p0 = chain8(xv[0], xv[1], cv[0], cv[1]);
  const float p1 = chain8(xv[2], xv[3], cv[2], cv[3]);
  const float p2 = chain8(xv[4], xv[5], cv[4], cv[5]);
  const float p3 = chain8(xv[6], xv[7], cv[6], cv[7]);
  return (p0 + p1) + (p2 + p3);
}

struct FinalOut {   // outputs of a search (any pointer may be null)
  int32_t* i1;
  float* d1;
  int32_t* i2;
  float* d2;
  float ratio;
  uint8_t* mask;
  ic_top2_rec* rec;
};

// s1 <= s2 by (value, index): squared distances in, Euclidean distances and the ratio test out.
__device__ __forceinline__ void write_final(const FinalOut& o, int64_t row, float s1, int a1,
                                            float s2, int a2) {
  const float e1 = __builtin_sqrtf(s1), e2 = __builtin_sqrtf(s2);
  if (o.i1) o.i1[row] = a1;
  if (o.d1) o.d1[row] = e1;
  if (o.i2) o.i2[row] = a2;
  if (o.d2) o.d2[row] = e2;
  if (o.mask) o.mask[row] = (e1 < e2 * o.ratio) ? 1 : 0;
  if (o.rec) {
    ic_top2_rec r;
    r.d1 = e1; r.i1 = a1; r.d2 = e2; r.i2 = a2;
    o.rec[row] = r;
  }
}

// Workgroup b owns rows [b chunk, (b + 1) chunk): merges the pieces of their row tiles, evaluates
// the two candidates of every row by direct differences, writes the outputs, and decides per row
// whether the pair is certain to hold the answer or the row must be searched again by direct
// differences; such rows are appended to `list` (one atomic increment of *count per flagged row:
// they are rare, their order in the list does not matter -- every row's second look is
// independent and writes only that row's outputs -- and *count was cleared by the search kernel).
// Workgroup 0 also clears the tickets of the second look.
//
// When only the index of the nearest codeword is wanted (rowband != NULL: the constant part of
// each row's band, left by the search kernel), a first pass with one thread per row settles
// every row whose second-best value lies outside the band of the best: no candidate can then be
// nearer than the best by direct evaluation either, so its index is the answer and neither the
// row nor any codeword is read again.  The other rows go through the pass below.
//
// WIDE: rows of D = 128 nslab columns (row stride D).  A row no longer fits a group's registers,
// so every direct evaluation walks the row's slabs again (they stay in cache between the
// candidates of a row), adding slab after slab in the order k_exact_top2 uses.
constexpr int kListBatch = 8;   // lists of a row whose records are fetched together

// MANY: every row is cut into many pieces and a group of 16 lanes merges a row's lists (its own
// instantiation: the common one keeps the registers, and the occupancy, it had without that code)
template <bool WIDE, bool MANY>
__global__ __launch_bounds__(256) void k_rescore(const float* __restrict__ X, int64_t N,
                                                 const float* __restrict__ C,
                                                 const float* __restrict__ aux, int D,
                                                 const float* __restrict__ mean,
                                                 PartBufs pb,
                                                 const float* __restrict__ rowband, TailGeom tail,
                                                 int64_t split_row0, int mode2, int64_t chunk,
                                                 FinalOut out, int32_t* __restrict__ list,
                                                 uint32_t* __restrict__ count,
                                                 uint32_t* __restrict__ tickets) {
  __shared__ int todo[256];
  __shared__ int ntodo;
  const int sub = threadIdx.x & 15;
  const int grp = threadIdx.x >> 4;
  const int64_t row_begin = (int64_t)blockIdx.x * chunk;
  const int64_t row_end = row_begin + chunk < N ? row_begin + chunk : N;
  const f32x4 m0 = *reinterpret_cast<const f32x4*>(mean + sub * 8);
  const f32x4 m1 = *reinterpret_cast<const f32x4*>(mean + sub * 8 + 4);
  const float cn_max = aux[kAuxCnMax], cr2_max = aux[kAuxCr2Max];
  const float band_slope = WIDE ? 4.f * gamma_direct_d(D) : kBandSlope;
  const float flag_slope = WIDE ? 3.f * gamma_direct_d(D) : kFlagSlope;
  if (blockIdx.x == 0)
    for (int t = threadIdx.x; t < kExactGrid; t += 256) tickets[t] = 0;
  // lists of a row: the two lane halves of every piece of its row tile (rows below split_row0
  // were searched whole)
  auto n_lists = [&](int64_t row) {
    return row >= split_row0 ? 2 * tail_pieces(tail, (row - split_row0) / tail.rows_per_wg) : 2;
  };

  // What the lists of a row say once merged: the two best (value, index) pairs; v3, the third
  // smallest value over all lists; u3, the smallest value a list holds without its index (the
  // third of a list, the fourth when the search kept three indices), i.e. the best candidate
  // whose INDEX no record holds.
  struct Head {
    ic_top2_rec b;
    float v3, u3;
  };
  const float* __restrict__ unnamed = mode2 ? pb.d4 : pb.d3;
  auto merge_lists = [&](int64_t row) {
    Head hd;
    hd.b = pb.rec[row];
    hd.v3 = pb.d3[row];
    hd.u3 = unnamed[row];
    const int ns = n_lists(row);
    auto take = [&](const ic_top2_rec& o, float o3, float ou) {
      hd.u3 = __builtin_fminf(hd.u3, ou);
      hd.v3 = third_of_union(hd.b.d1, hd.b.d2, hd.v3, o.d1, o.d2, o3);
      top2_merge_pairs(hd.b.d1, hd.b.i1, hd.b.d2, hd.b.i2, o.d1, o.i1, o.d2, o.i2);
    };
    for (int s = 1; s < ns; ++s) {
      const int64_t slot = part_slot(s, row, N, split_row0);
      take(pb.rec[slot], pb.d3[slot], unnamed[slot]);
    }
    return hd;
  };

  // The same merge by the 16 lanes of a group, for rows cut into many pieces (few rows against a
  // long codebook: 79 pieces and more per row tile): lane `sub` takes the lists sub, sub + 16, ...,
  // eight loads in flight, and the sixteen partial heads are folded by a butterfly -- two or
  // three round trips to memory per row instead of one per list.  Every lane returns the row's
  // head.  (Which of several lists' "none" entries survives depends on the order of merging;
  // their values are the row's start value in every list.)
  auto merge_lists_group = [&](int64_t row) {
    Head hd;
    hd.b.d1 = hd.b.d2 = hd.v3 = hd.u3 = __builtin_inff();
    hd.b.i1 = hd.b.i2 = -1;
    const int ns = n_lists(row);
    auto take = [&](const ic_top2_rec& o, float o3, float ou) {
      hd.u3 = __builtin_fminf(hd.u3, ou);
      hd.v3 = third_of_union(hd.b.d1, hd.b.d2, hd.v3, o.d1, o.d2, o3);
      top2_merge_pairs(hd.b.d1, hd.b.i1, hd.b.d2, hd.b.i2, o.d1, o.i1, o.d2, o.i2);
    };
    for (int s0 = 0; s0 < ns; s0 += 16 * kListBatch) {   // uniform in the group
      ic_top2_rec o[kListBatch];
      float o3[kListBatch], ou[kListBatch];
#pragma unroll
      for (int u = 0; u < kListBatch; ++u) {
        const int l = s0 + 16 * u + sub;
        const int64_t slot = part_slot(l < ns ? l : 0, row, N, split_row0);
        o[u] = pb.rec[slot];
        o3[u] = pb.d3[slot];
        ou[u] = unnamed[slot];
      }
#pragma unroll
      for (int u = 0; u < kListBatch; ++u)
        if (s0 + 16 * u + sub < ns) take(o[u], o3[u], ou[u]);
    }
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) {
      ic_top2_rec o;
      o.d1 = __shfl_xor(hd.b.d1, off);
      o.i1 = __shfl_xor(hd.b.i1, off);
      o.d2 = __shfl_xor(hd.b.d2, off);
      o.i2 = __shfl_xor(hd.b.i2, off);
      const float o3 = __shfl_xor(hd.v3, off), ou = __shfl_xor(hd.u3, off);
      take(o, o3, ou);
    }
    return hd;
  };

  // one row, by the 16 lanes of a group; a row that needs the direct search is put on the list
  auto full_look = [&](int64_t row, const Head& hd) {
    const float* xrow = X + row * (int64_t)(WIDE ? D : kD) + sub * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(xrow + 4);
    const ic_top2_rec b = hd.b;
    const float v3 = hd.v3, u3 = hd.u3;
    const int ns = n_lists(row);
    float xn = group16_sum(chain8(x0, x1, m0, m1));
    if constexpr (WIDE) {
      for (int sl = kD; sl < D; sl += kD)
        xn += group16_sum(chain8(*reinterpret_cast<const f32x4*>(xrow + sl),
                                 *reinterpret_cast<const f32x4*>(xrow + sl + 4),
                                 *reinterpret_cast<const f32x4*>(mean + sl + sub * 8),
                                 *reinterpret_cast<const f32x4*>(mean + sl + sub * 8 + 4)));
    }
    auto direct = [&](int idx) {
      const float* cr = C + (int64_t)idx * (WIDE ? D : kD) + sub * 8;
      if constexpr (!WIDE) {
        return group16_tree(chain8(x0, x1, *reinterpret_cast<const f32x4*>(cr),
                                   *reinterpret_cast<const f32x4*>(cr + 4)));
      } else {
        float tot = group16_chain(chain8(x0, x1, *reinterpret_cast<const f32x4*>(cr),
                                         *reinterpret_cast<const f32x4*>(cr + 4)), 0.f);
        for (int sl = kD; sl < D; sl += kD)
          tot = group16_chain(chain8(*reinterpret_cast<const f32x4*>(xrow + sl),
                                     *reinterpret_cast<const f32x4*>(xrow + sl + 4),
                                     *reinterpret_cast<const f32x4*>(cr + sl),
                                     *reinterpret_cast<const f32x4*>(cr + sl + 4)), tot);
        return tot;
      }
    };
    // both candidates' loads go out together with the row's (an absent candidate reads
    // codeword 0 and is discarded)
    float s1 = direct(b.i1 >= 0 ? b.i1 : 0), s2 = direct(b.i2 >= 0 ? b.i2 : 0);
    if (b.i1 < 0) s1 = __builtin_inff();
    if (b.i2 < 0) s2 = __builtin_inff();
    // Can a candidate outside the pair be the nearest (the second nearest, when both count)?
    // Only one whose computed value lies within `band` of the reference value: two error bounds,
    // plus what the direct evaluation needs to tell it from the reference candidate(s) -- whose
    // direct squared distance is s1 (the computed best) or, when both count, the larger of the two.
    const float ref = mode2 ? b.d2 : b.d1;
    const float band = 2.f * (WIDE ? error_bound_d(xn, cn_max, cr2_max, D)
                                   : error_bound(xn, cn_max, cr2_max)) +
                       flag_slope * (mode2 ? __builtin_fmaxf(s1, s2) : s1);
    int a1 = b.i1, a2 = b.i2;
    if (a2 >= 0 && lex_less(s2, a2, s1, a1)) {
      const float ts = s1; s1 = s2; s2 = ts;
      const int ti = a1; a1 = a2; a2 = ti;
    }
    bool unsure = (v3 - ref) <= band;   // false on NaN / inf - inf
    if (unsure && !((u3 - ref) <= band)) {
      // every candidate inside the band sits in some list with its index (no list holds an
      // unnamed one there): evaluate them directly too and keep the two best of everything
      // evaluated
      for (int s = 0; s < ns; ++s) {
        const int64_t slot = part_slot(s, row, N, split_row0);
        const ic_top2_rec o = pb.rec[slot];
        const int cand[3] = {o.i1, o.i2, mode2 ? pb.i3[slot] : -1};
        const float val[3] = {o.d1, o.d2, mode2 ? pb.d3[slot] : 0.f};
#pragma unroll
        for (int t = 0; t < 3; ++t) {
          const int idx = cand[t];
          if (idx < 0 || idx == a1 || idx == a2 || !((val[t] - ref) <= band)) continue;
          const float sd = direct(idx);
          if (lex_less(sd, idx, s1, a1)) {
            s2 = s1; a2 = a1; s1 = sd; a1 = idx;
          } else if (a2 < 0 || lex_less(sd, idx, s2, a2)) {
            s2 = sd; a2 = idx;
          }
        }
      }
      unsure = false;
    }
    if (sub == 0) {
      write_final(out, row, s1, a1, s2, a2);
      if (unsure) list[atomicAdd(count, 1u)] = (int32_t)row;
    }
  };

  if constexpr (MANY) {
    for (int64_t row = row_begin + grp; row < row_end; row += 16) {
      const Head hd = merge_lists_group(row);
      if (rowband) {
        // the shortcut of the index-only search, see below
        const float band = __builtin_fmaf(hd.b.d1, band_slope, rowband[row]);
        if (hd.b.i1 >= 0 && (hd.b.d2 - hd.b.d1) > band) {   // the same in all 16 lanes
          if (sub == 0) out.i1[row] = hd.b.i1;
          continue;
        }
      }
      full_look(row, hd);
    }
  } else if (rowband == nullptr) {
    // the lists of 256 rows are merged by one thread each (coalesced record reads, one round
    // trip for all of them), then the groups evaluate the rows: a row's chain of dependent loads
    // is one step (row + codewords) instead of two (record, then codewords)
    __shared__ Head heads[256];
    for (int64_t base = row_begin; base < row_end; base += 256) {   // uniform trip count
      if (base + threadIdx.x < row_end) heads[threadIdx.x] = merge_lists(base + threadIdx.x);
      __syncthreads();
      const int here = row_end - base < 256 ? (int)(row_end - base) : 256;
      for (int r = grp; r < here; r += 16) full_look(base + r, heads[r]);
      __syncthreads();
    }
  } else {
    for (int64_t base = row_begin; base < row_end; base += 256) {   // uniform trip count
      if (threadIdx.x == 0) ntodo = 0;
      __syncthreads();
      const int64_t row = base + threadIdx.x;
      if (row < row_end) {
        ic_top2_rec b = pb.rec[row];
        const int ns = n_lists(row);
        for (int s = 1; s < ns; ++s) {
          const ic_top2_rec o = pb.rec[part_slot(s, row, N, split_row0)];
          top2_merge_pairs(b.d1, b.i1, b.d2, b.i2, o.d1, o.i1, o.d2, o.i2);
        }
        // the gate's band of the best: everything outside the pair lies above b.d2 or was
        // never ranked, and what was never ranked lies above this band as well
        const float band = __builtin_fmaf(b.d1, band_slope, rowband[row]);
        if (b.i1 >= 0 && (b.d2 - b.d1) > band) {   // false on NaN
          out.i1[row] = b.i1;
        } else {
          todo[atomicAdd(&ntodo, 1)] = (int)threadIdx.x;   // any order: rows are independent
        }
      }
      __syncthreads();
      const int nt = ntodo;
      for (int t = grp; t < nt; t += 16) full_look(base + todo[t], merge_lists(base + todo[t]));
      __syncthreads();
    }
  }
}

// Rows per workgroup of k_rescore: few, so that many independent gathers are in flight -- one row
// per 16-lane group while that gives no more than 16384 workgroups (a call with a few thousand
// rows is one round trip to memory per step, not four).
int64_t rescore_chunk(int64_t n) {
  int64_t chunk = (n + 16383) / 16384;
  chunk = (chunk + 15) / 16 * 16;
  return chunk < 16 ? 16 : chunk;
}

hipError_t launch_rescore(const float* x, int64_t n, const float* c, const float* aux,
                          const float* mean, int D, const PartBufs& pb, const float* rowband,
                          const AssignPlan& plan, bool mode2, int32_t* i1, float* d1, int32_t* i2,
                          float* d2, float ratio, uint8_t* mask, ic_top2_rec* rec, int32_t* list,
                          uint32_t* count, uint32_t* tickets, hipStream_t s) {
  // the shortcut path looks at one row per thread: whole blocks of 256 rows per workgroup
  int64_t chunk = rescore_chunk(n);
  if (rowband && chunk < 256) chunk = 256;
  // every row cut into many pieces (few rows, long codebook): the kernel merges a row's lists by
  // a group of 16 lanes, so a workgroup takes 16 rows and they all go on at once
  const bool many = plan.full_tiles == 0 && plan.max_pieces > kListBatch;
  if (many) {
    chunk = (n + 16383) / 16384;
    chunk = (chunk + 15) / 16 * 16;
  }
  const int64_t grid = (n + chunk - 1) / chunk;
  if (grid < 1 || n >= 0x7fffffffLL) return hipErrorInvalidValue;
  // the shortcut answers with an index only
  if (rowband && (mode2 || d1 || !i1)) return hipErrorInvalidValue;
  if (mode2 && (!pb.i3 || !pb.d4)) return hipErrorInvalidValue;
  if (D < kD || D % kD != 0 || D > kMaxWideD) return hipErrorInvalidValue;
  auto kern = many ? (D > kD ? k_rescore<true, true> : k_rescore<false, true>)
                   : (D > kD ? k_rescore<true, false> : k_rescore<false, false>);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), 0, s, x, n, c, aux, D, mean, pb,
                     rowband, TailGeom{plan.tail_span, plan.tail_quota, plan.rows_per_wg, plan.tail_long},
                     plan.full_tiles * (int64_t)plan.rows_per_wg, mode2 ? 1 : 0, chunk,
                     FinalOut{i1, d1, i2, d2, ratio, mask, rec}, list, count, tickets);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Upper bound, per row, on the value the search kernel will compute for the row's nearest
// codeword, from a guess `hint[row]` of that codeword (k-means: the label of the previous
// iteration).  With v the value of the hinted codeword as computed here and B the rounding bound:
// the search kernel's value of that codeword is at most v + 2 B, so the kernel's best is, and
// starting every lane's list at  thr = v + 7 B + band(v)  keeps thr - best above the band the
// flag test of k_rescore uses: a row whose guess was right and that has no near-tie is not
// flagged, and every candidate that could matter still lies below thr and is ranked.
__global__ __launch_bounds__(256) void k_hint_threshold(const float* __restrict__ X, int64_t N,
                                                        const float* __restrict__ C, int64_t K,
                                                        const float* __restrict__ aux,
                                                        const float* __restrict__ mean, int D,
                                                        const int32_t* __restrict__ hint,
                                                        const int32_t* __restrict__ rank,
                                                        float* __restrict__ thr) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  const float cn_max = aux[kAuxCnMax], cr2_max = aux[kAuxCr2Max];
  const float band_slope = 4.f * gamma_direct_d(D);   // kBandSlope when D = 128
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < N; row += stride) {
    const int hraw = hint[row];
    const bool ok = hraw >= 0 && hraw < K;
    const int hh = ok ? hraw : 0;
    const float* xrow = X + row * D + sub * 8;
    const float* crow = C + (int64_t)hh * D + sub * 8;
    float dot = 0.f, xn = 0.f, xm = 0.f;
    for (int sl = 0; sl < D; sl += kD) {
      const float* mu = mean + sl + sub * 8;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = xrow[sl + e] - mu[e];
        dot = __builtin_fmaf(t, crow[sl + e], dot);
        xn = __builtin_fmaf(t, t, xn);
        xm = __builtin_fmaf(t, mu[e], xm);
      }
    }
    dot = group16_sum(dot);
    xn = group16_sum(xn);
    xm = group16_sum(xm);
    if (sub == 0) {
      // (the norms follow the search order when the codebook has one)
      const float v = aux[kAuxNorms + (rank ? rank[hh] : hh)] - 2.f * dot;
      const float bound = error_bound_d(xn, cn_max, cr2_max, D);
      const float sq = __builtin_fmaxf(xn + 2.f * xm + v, 0.f);   // about the squared distance
      thr[row] = ok ? __builtin_fminf(v + 7.f * bound + band_slope * sq + 1e-30f, kNone) : kNone;
    }
  }
}

hipError_t launch_hint_threshold(const float* x, int64_t n, const float* c, int64_t k,
                                 const float* aux, const float* mean, int D, const int32_t* hint,
                                 const int32_t* rank, float* thr, hipStream_t s) {
  int64_t grid = (n + 15) / 16;
  if (grid > 16384) grid = 16384;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_hint_threshold, dim3((unsigned)grid), dim3(256), 0, s, x, n, c, k, aux,
                     mean, D, hint, rank, thr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Direct search: sum (x - c)^2 in float32 for every (row, codeword) pair, no norm expansion and
// therefore no cancellation.  Twice the arithmetic of the matrix-core kernel (a subtraction and an
// FMA per component) plus the reads, four times its run time, on the vector ALU.  It serves as the
// whole-input checker (ic_assign_top2_exact) and as the second look at the rows k_rescore flags.
// 16 rows per group live in LDS (every lane reads the same row element: a broadcast), every thread
// walks codewords t, t+256, ... keeping a top-2 per row, and the 256 partial top-2s are folded by
// (distance, index).
//
// Rows: `list` (with *count entries) names them, or rows 0 .. n_rows-1 when list is NULL.  The
// work is cut into (group of 16 rows, piece of the codebook) items that the workgroups of a
// fixed-size grid walk; the number of pieces per group is chosen ON THE DEVICE from the number of
// rows (exact_pieces): a handful of flagged rows against a long codebook must not be one
// workgroup's job, and many rows need no cutting at all.  With more than one piece the partial
// top-2 of an item (squared distances) goes to `partial` [row][pieces] and the workgroup draws a
// ticket of its group: whoever draws the last one folds the group's pieces by (squared distance,
// index) -- the same values a whole scan would have compared -- and writes the outputs; with one
// piece the outputs are written directly.  Every loop bound is uniform in the workgroup and known
// when the kernel starts; with an empty list every workgroup returns at once.
constexpr int kExactRows = 16;

// Pieces per group and codewords per piece.
__device__ __forceinline__ int exact_pieces(int64_t n_groups, int grid, int max_pieces, int K,
                                            int* k_per) {
  int64_t s = 1;
  if (max_pieces > 1 && n_groups > 0 && n_groups < grid) {
    s = grid / n_groups;
    const int64_t by_len = (K + 255) / 256;   // at least 256 codewords (one per thread) a piece
    s = s < by_len ? s : by_len;
    s = s < max_pieces ? s : max_pieces;
    s = s < 1 ? 1 : s;
  }
  const int per = (int)((K + s - 1) / s);
  *k_per = per;
  return (K + per - 1) / per;
}

__global__ __launch_bounds__(256) void k_exact_top2(const float* __restrict__ X, int64_t n_rows,
                                                    const float* __restrict__ C, int K,
                                                    const int32_t* __restrict__ list,
                                                    const uint32_t* __restrict__ count,
                                                    int max_pieces,
                                                    ic_top2_rec* __restrict__ partial,
                                                    uint32_t* __restrict__ tickets, int D,
                                                    FinalOut fin) {
  // the group's rows, [kExactRows][D] (D = 128 for descriptors, up to kMaxWideD)
  extern __shared__ __attribute__((aligned(16))) float xs[];
  __shared__ uint32_t ticket;
  __shared__ float rd1[4][kExactRows], rd2[4][kExactRows];
  __shared__ int ri1[4][kExactRows], ri2[4][kExactRows];
  __shared__ float fd1[16][kExactRows], fd2[16][kExactRows];   // the fold of a group's pieces
  __shared__ int fi1[16][kExactRows], fi2[16][kExactRows];
  __shared__ int64_t rows[kExactRows];
  const int64_t total = list ? (int64_t)*count : n_rows;
  const int64_t n_groups = (total + kExactRows - 1) / kExactRows;
  int k_per;
  const int pieces = exact_pieces(n_groups, (int)gridDim.x, max_pieces, K, &k_per);
  const int64_t items = n_groups * pieces;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  for (int64_t item = blockIdx.x; item < items; item += gridDim.x) {
    const int64_t g = item / pieces;
    const int sp = (int)(item - g * pieces);
    const int k0 = sp * k_per;
    const int k1 = k0 + k_per < K ? k0 + k_per : K;
    __syncthreads();   // the previous item's LDS is no longer read
    if (threadIdx.x < kExactRows) {
      int64_t t = g * kExactRows + threadIdx.x;
      t = t < total ? t : total - 1;
      rows[threadIdx.x] = list ? (int64_t)list[t] : t;
    }
    __syncthreads();
    const int D4 = D / 4;
    for (int e = threadIdx.x; e < kExactRows * D4; e += 256) {
      const int r = e / D4, q = e % D4;
      reinterpret_cast<f32x4*>(xs + r * D)[q] = reinterpret_cast<const f32x4*>(X + rows[r] * D)[q];
    }
    __syncthreads();
    // rows of this group that exist (the last group of a short list is mostly padding, and a
    // short list is the usual case of the second look): uniform, so the guards below are scalar
    // branches
    const int nr = total - g * kExactRows < kExactRows ? (int)(total - g * kExactRows) : kExactRows;
    float d1[kExactRows], d2[kExactRows];
    int i1[kExactRows], i2[kExactRows];
#pragma unroll
    for (int r = 0; r < kExactRows; ++r) {
      d1[r] = d2[r] = __builtin_inff();
      i1[r] = i2[r] = -1;
    }
    // two copies of the scan: a full group runs without the per-row guards
    auto scan = [&](auto full_tag) {
      constexpr bool kFull = decltype(full_tag)::value;
      for (int j = k0 + (int)threadIdx.x; j < k1; j += 256) {
        const float* crow = C + (int64_t)j * D;
        float acc[kExactRows];
#pragma unroll
        for (int r = 0; r < kExactRows; ++r) acc[r] = 0.f;
        const int nblk = D / 32;
#pragma unroll 1
        for (int blk = 0; blk < nblk; ++blk) {
          f32x4 cv[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) cv[q] = *reinterpret_cast<const f32x4*>(crow + blk * 32 + q * 4);
#pragma unroll
          for (int r = 0; r < kExactRows; ++r)
            if (kFull || r < nr) acc[r] += block32(xs + r * D + blk * 32, cv);
        }
#pragma unroll
        for (int r = 0; r < kExactRows; ++r)
          if (kFull || r < nr) top2_update(d1[r], d2[r], i1[r], i2[r], acc[r], j);
      }
    };
    if (nr == kExactRows) scan(std::true_type{});
    else scan(std::false_type{});
#pragma unroll
    for (int r = 0; r < kExactRows; ++r) {
      if (r >= nr) continue;
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) {
        const float od1 = __shfl_xor(d1[r], off), od2 = __shfl_xor(d2[r], off);
        const int oi1 = __shfl_xor(i1[r], off), oi2 = __shfl_xor(i2[r], off);
        top2_merge_pairs(d1[r], i1[r], d2[r], i2[r], od1, oi1, od2, oi2);
      }
      if (lane == 0) {
        rd1[wave][r] = d1[r]; ri1[wave][r] = i1[r];
        rd2[wave][r] = d2[r]; ri2[wave][r] = i2[r];
      }
    }
    __syncthreads();
    if (threadIdx.x < kExactRows) {
      const int r = threadIdx.x;
      float a1 = rd1[0][r], a2 = rd2[0][r];
      int j1 = ri1[0][r], j2 = ri2[0][r];
      for (int w = 1; w < 4; ++w)
        top2_merge_pairs(a1, j1, a2, j2, rd1[w][r], ri1[w][r], rd2[w][r], ri2[w][r]);
      const int64_t t = g * kExactRows + r;
      if (t < total) {
        if (pieces > 1) {
          ic_top2_rec rec;
          rec.d1 = a1; rec.i1 = j1; rec.d2 = a2; rec.i2 = j2;
          partial[t * pieces + sp] = rec;
        } else {
          write_final(fin, rows[r], a1, j1, a2, j2);
        }
      }
    }
    if (pieces > 1) {
      // the records above must be visible device-wide before the ticket is drawn, and the last
      // drawer must not read the others' records from a stale cache
      __threadfence();
      __syncthreads();
      if (threadIdx.x == 0) ticket = atomicAdd(&tickets[g], 1u);
      __syncthreads();
      if (ticket == (uint32_t)pieces - 1u) {   // uniform in the workgroup
        __threadfence();
        // sixteen threads per row, each folding every sixteenth piece (the records are read past
        // the first-level cache, one round trip each: a single thread walking hundreds of pieces
        // took 0.1 ms), then one thread per row folds the sixteen.  The two smallest by
        // (distance, index) do not depend on the order of folding.
        const int r = threadIdx.x & (kExactRows - 1), part = threadIdx.x >> 4;
        const int64_t t = g * kExactRows + r;
        float b1 = __builtin_inff(), b2 = __builtin_inff();
        int k1 = -1, k2 = -1;
        if (t < total) {
          const volatile ic_top2_rec* pr = partial + t * pieces;
          for (int s = part; s < pieces; s += 16)
            top2_merge_pairs(b1, k1, b2, k2, pr[s].d1, pr[s].i1, pr[s].d2, pr[s].i2);
        }
        fd1[part][r] = b1; fi1[part][r] = k1;
        fd2[part][r] = b2; fi2[part][r] = k2;
        __syncthreads();
        if (threadIdx.x < kExactRows && t < total) {
          for (int q = 1; q < 16; ++q)
            top2_merge_pairs(b1, k1, b2, k2, fd1[q][r], fi1[q][r], fd2[q][r], fi2[q][r]);
          write_final(fin, rows[r], b1, k1, b2, k2);
        }
      }
    }
  }
}

// once per device: the direct search keeps 16 rows of up to kMaxWideD columns in dynamic LDS
hipError_t exact_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_exact_top2),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             kExactRows * kMaxWideD * 4);
}

hipError_t launch_exact_top2(const float* x, int64_t n, const float* c, int64_t k, int D,
                             int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                             uint8_t* mask, hipStream_t s) {
  const int64_t grid = (n + kExactRows - 1) / kExactRows;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  if (D < kD || D % kD != 0 || D > kMaxWideD) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_exact_top2, dim3((unsigned)grid), dim3(256), kExactRows * D * 4, s, x, n, c,
                     (int)k, (const int32_t*)nullptr, (const uint32_t*)nullptr, 1,
                     (ic_top2_rec*)nullptr, (uint32_t*)nullptr, D,
                     FinalOut{i1, d1, i2, d2, ratio, mask, nullptr});
  return hipGetLastError();
}

// The second look at the flagged rows.  Their number is only known on the device, so the grid is
// fixed (kExactGrid workgroups) and the kernel derives the decomposition from *count.  tickets
// [kExactGrid] must be zero (k_rescore clears them).
hipError_t launch_exact_rows(const float* x, int64_t n, const float* c, int64_t k, int D,
                             const int32_t* list, const uint32_t* count, ic_top2_rec* partial,
                             uint32_t* tickets, int32_t* i1, float* d1, int32_t* i2, float* d2,
                             float ratio, uint8_t* mask, ic_top2_rec* rec, hipStream_t s) {
  const FinalOut fin{i1, d1, i2, d2, ratio, mask, rec};
  if (D < kD || D % kD != 0 || D > kMaxWideD) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_exact_top2, dim3(kExactGrid), dim3(256), kExactRows * D * 4, s, x, n, c,
                     (int)k, list, count, kExactGrid, partial, tickets, D, fin);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// dst [n][D] <- src [n][d], zero beyond column d
__global__ __launch_bounds__(256) void k_pad_rows(const float* __restrict__ src, int64_t n, int d,
                                                  int D, float* __restrict__ dst) {
  const int64_t total = n * D;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / D;
    const int cidx = (int)(t % D);
    dst[t] = cidx < d ? src[r * d + cidx] : 0.f;
  }
}

hipError_t launch_pad_rows(const float* src, int64_t n, int d, int D, float* dst, hipStream_t s) {
  int64_t grid = (n * D + 255) / 256;
  if (grid > 65536) grid = 65536;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)grid), dim3(256), 0, s, src, n, d, D, dst);
  return hipGetLastError();
}

// 16 lanes per row, any d; out[row] = sum of squares (fixed order).
__global__ __launch_bounds__(256) void k_row_sqnorm(const float* __restrict__ X, int64_t n, int d,
                                                    int64_t npad, float* __restrict__ out) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < npad; row += stride) {
    float s = 0.f;
    if (row < n) {
      const float* xr = X + row * d;
      for (int cidx = sub; cidx < d; cidx += 16) s = __builtin_fmaf(xr[cidx], xr[cidx], s);
    }
    s = group16_sum(s);
    if (sub == 0) out[row] = row < n ? s : __builtin_inff();
  }
}

hipError_t launch_row_sqnorm(const float* x, int64_t n, int d, float* out, hipStream_t s) {
  int64_t grid = (n + 15) / 16;
  if (grid > 16384) grid = 16384;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)grid), dim3(256), 0, s, x, n, d, n, out);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The codebook's auxiliary block (layout: kAux* in ic_internal.h): the mean vector m, the centred
// squared norms |c_j - m|^2 the search kernel starts its accumulators from, and the two maxima of
// the rounding model.  Any m would do -- distances do not depend on it -- so its own rounding is
// of no concern; the sums run in a fixed order, so the block is reproducible.
constexpr int kAuxSlabMax = 128;

// slab g of rows, block y of 128 columns: column sums of rows [g rows_per_slab, ...) -> psum [slab][D]
__global__ __launch_bounds__(256) void k_aux_colsum(const float* __restrict__ C, int64_t K, int D,
                                                    int64_t rows_per_slab,
                                                    float* __restrict__ psum) {
  __shared__ f32x4 red[8][32];
  const int col4 = threadIdx.x & 31, phase = threadIdx.x >> 5;
  const int c0 = blockIdx.y * kD;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_slab;
  const int64_t r1 = r0 + rows_per_slab < K ? r0 + rows_per_slab : K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int64_t r = r0 + phase; r < r1; r += 8)
    acc += reinterpret_cast<const f32x4*>(C + r * D + c0)[col4];
  red[phase][col4] = acc;
  __syncthreads();
  if (phase == 0) {
    f32x4 t = red[0][col4];
    for (int p = 1; p < 8; ++p) t += red[p][col4];
    reinterpret_cast<f32x4*>(psum + (int64_t)blockIdx.x * D + c0)[col4] = t;
  }
}

// Every workgroup folds the slab sums into the mean (same order everywhere, so the same bits),
// then 16 lanes per codeword: centred and raw squared norm, 128 columns after 128 columns;
// per-workgroup maxima -> pmax.  The mean goes to aux + mean_off.
__global__ __launch_bounds__(256) void k_aux_norms(const float* __restrict__ C, int64_t K, int D,
                                                   int64_t kpad, const float* __restrict__ psum,
                                                   int n_slabs, float* __restrict__ aux,
                                                   float* __restrict__ norms_out,
                                                   int64_t mean_off, float* __restrict__ pmax) {
  extern __shared__ __attribute__((aligned(16))) float mean[];   // [D]
  __shared__ float wmax[2][16];
  for (int col = threadIdx.x; col < D; col += 256) {
    double t = 0.0;
    for (int g = 0; g < n_slabs; ++g) t += (double)psum[(int64_t)g * D + col];
    const float m = (float)(t / (double)K);
    mean[col] = m;
    if (blockIdx.x == 0) {
      aux[mean_off + col] = m;
      if (D == kD) aux[aux_tail_offset(kpad) + col] = 2.f * m;
    }
  }
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // |m|, never below the true norm of the stored vector
    double t = 0.0;
    for (int col = 0; col < D; ++col) t += (double)mean[col] * (double)mean[col];
    aux[kAuxMeanNorm] = (float)__builtin_sqrt(t) * (1.f + 2.f * kU);
  }
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
  float cmax = 0.f, rmax = 0.f;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row = (int64_t)blockIdx.x * 16 + grp; row < kpad; row += stride) {
    float cn = kPadNorm;   // padding behind the last codeword: above every list's start value
    if (row < K) {
      cn = 0.f;
      float rn = 0.f;
      for (int sl = 0; sl < D; sl += kD) {
        const float* cr = C + row * D + sl + sub * 8;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cr);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(cr + 4);
        const f32x4 m0 = *reinterpret_cast<const f32x4*>(&mean[sl + sub * 8]);
        const f32x4 m1 = *reinterpret_cast<const f32x4*>(&mean[sl + sub * 8 + 4]);
        cn += group16_sum(chain8(c0, c1, m0, m1));
        rn += group16_sum(chain8(c0, c1, zero, zero));
      }
      cmax = __builtin_fmaxf(cmax, cn);
      rmax = __builtin_fmaxf(rmax, rn);
    }
    if (sub == 0) norms_out[row] = cn;
  }
  if (sub == 0) {
    wmax[0][grp] = cmax;
    wmax[1][grp] = rmax;
  }
  __syncthreads();
  if (threadIdx.x < 2) {
    float t = 0.f;
    for (int g = 0; g < 16; ++g) t = __builtin_fmaxf(t, wmax[threadIdx.x][g]);
    pmax[(int64_t)blockIdx.x * 2 + threadIdx.x] = t;
  }
}

__global__ __launch_bounds__(64) void k_aux_max(const float* __restrict__ pmax, int n,
                                                float* __restrict__ aux) {
  float c = 0.f, r = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) {
    c = __builtin_fmaxf(c, pmax[2 * i]);
    r = __builtin_fmaxf(r, pmax[2 * i + 1]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    c = __builtin_fmaxf(c, __shfl_xor(c, off));
    r = __builtin_fmaxf(r, __shfl_xor(r, off));
  }
  if (threadIdx.x == 0) {
    aux[kAuxCnMax] = c;
    aux[kAuxCr2Max] = r;
  }
}

// The search order of a codebook: rank of every codeword by (|c_j - m|^2, j), counted -- 64
// codewords per workgroup, every wave counting over a quarter of each 1024-norm tile -- and the
// norms, the permutation both ways and the codewords themselves laid out in that order.  The
// norms are compared as bit patterns: a total order whatever the input holds, so the ranks are
// a permutation even when norms are NaN, and no position of the ordered copy stays unwritten.
__global__ __launch_bounds__(256) void k_order_codebook(const float* __restrict__ norm0, int K,
                                                        int kpad, const float* __restrict__ C,
                                                        float* __restrict__ norms_out,
                                                        int32_t* __restrict__ perm,
                                                        int32_t* __restrict__ rank,
                                                        float* __restrict__ cperm) {
  __shared__ __attribute__((aligned(16))) unsigned tile[1024];
  __shared__ int part[4][64];
  __shared__ int place[64];
  const int jl = threadIdx.x & 63, seg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + jl;
  const unsigned mine = j < K ? __builtin_bit_cast(unsigned, norm0[j]) : 0xffffffffu;
  int cnt = 0;
  for (int base = 0; base < K; base += 1024) {
    __syncthreads();
    for (int t = threadIdx.x; t < 1024; t += 256)
      tile[t] = base + t < K ? __builtin_bit_cast(unsigned, norm0[base + t]) : 0xffffffffu;
    __syncthreads();
    // same address in every lane (a broadcast), four norms per read; the entries behind the
    // last codeword hold the largest key at indices above every j, so they never count
    const uint4* tv = reinterpret_cast<const uint4*>(tile + seg * 256);
    const int i0 = base + seg * 256;
#pragma unroll 4
    for (int q = 0; q < 64; ++q) {
      const uint4 o = tv[q];
      const int i = i0 + 4 * q;
      cnt += (o.x < mine || (o.x == mine && i < j)) ? 1 : 0;
      cnt += (o.y < mine || (o.y == mine && i + 1 < j)) ? 1 : 0;
      cnt += (o.z < mine || (o.z == mine && i + 2 < j)) ? 1 : 0;
      cnt += (o.w < mine || (o.w == mine && i + 3 < j)) ? 1 : 0;
    }
  }
  part[seg][jl] = cnt;
  __syncthreads();
  if (seg == 0) {
    const int r = part[0][jl] + part[1][jl] + part[2][jl] + part[3][jl];
    place[jl] = j < K ? r : -1;
    if (j < K) {
      rank[j] = r;
      perm[r] = j;
      norms_out[r] = norm0[j];
    }
  }
  // behind the last codeword: padding norms, identity entries (never read for a real candidate)
  for (int t = K + blockIdx.x * 256 + threadIdx.x; t < kpad; t += gridDim.x * 256) {
    norms_out[t] = kPadNorm;
    perm[t] = t;
    rank[t] = t;
  }
  __syncthreads();
  const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int rl = grp; rl < 64; rl += 16) {
    const int r = place[rl];
    if (r < 0) continue;
    const f32x4* src = reinterpret_cast<const f32x4*>(C + (int64_t)(blockIdx.x * 64 + rl) * kD);
    f32x4* dst = reinterpret_cast<f32x4*>(cperm + (int64_t)r * kD);
    dst[sub] = src[sub];
    dst[sub + 16] = src[sub + 16];
  }
}

static int aux_norm_grid(int64_t kpad) {
  int64_t g = (kpad + 15) / 16;
  return (int)(g > 512 ? 512 : (g < 1 ? 1 : g));
}

size_t codebook_aux_scratch_bytes(int64_t k, int D) {
  const int64_t kpad = (k + kStageK - 1) / kStageK * kStageK;
  // column sums, per-workgroup maxima, and the norms in codebook order while the search order
  // is being worked out
  return (size_t)kAuxSlabMax * D * 4 + (size_t)512 * 2 * 4 + (aux_ordered(k, D) ? (size_t)kpad * 4 : 0);
}

// c [k][D] with D a multiple of 128 (narrower rows are padded by the caller)
hipError_t launch_codebook_aux(const float* c, int64_t k, int D, float* aux, char* scratch,
                               hipStream_t s) {
  if (D < kD || D % kD != 0 || D > kMaxWideD) return hipErrorInvalidValue;
  const int64_t kpad = (k + kStageK - 1) / kStageK * kStageK;
  int64_t n_slabs = (k + 63) / 64;
  if (n_slabs > kAuxSlabMax) n_slabs = kAuxSlabMax;
  const int64_t rows_per_slab = (k + n_slabs - 1) / n_slabs;
  n_slabs = (k + rows_per_slab - 1) / rows_per_slab;
  float* psum = reinterpret_cast<float*>(scratch);
  float* pmax = psum + (size_t)kAuxSlabMax * D;
  hipLaunchKernelGGL(k_aux_colsum, dim3((unsigned)n_slabs, (unsigned)(D / kD)), dim3(256), 0, s, c,
                     k, D, rows_per_slab, psum);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int grid = aux_norm_grid(kpad);
  const bool ordered = aux_ordered(k, D);
  float* norm0 = ordered ? pmax + 512 * 2 : aux + kAuxNorms;
  hipLaunchKernelGGL(k_aux_norms, dim3((unsigned)grid), dim3(256), D * 4, s, c, k, D, kpad, psum,
                     (int)n_slabs, aux, norm0, aux_mean_offset(kpad, D), pmax);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_aux_max, dim3(1), dim3(64), 0, s, pmax, grid, aux);
  e = hipGetLastError();
  if (e != hipSuccess || !ordered) return e;
  hipLaunchKernelGGL(k_order_codebook, dim3((unsigned)((k + 63) / 64)), dim3(256), 0, s, norm0,
                     (int)k, (int)kpad, c, aux + kAuxNorms,
                     reinterpret_cast<int32_t*>(aux + aux_perm_offset(kpad)),
                     reinterpret_cast<int32_t*>(aux + aux_rank_offset(kpad)),
                     aux + aux_cperm_offset(kpad));
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// One thread per row: fold n_shards records (already exact Euclidean distances) by
// (distance, global index).  The shard bases travel as a kernel argument.
struct ShardBases {
  int64_t b[kMaxShards];
};

// packed != 0: every shard's block holds n + 1 records, the last one carrying the shard's base as
// an int64 in its first eight bytes (what the ranks all-gather, so no base travels separately).
__global__ __launch_bounds__(256) void k_top2_merge(const ic_top2_rec* __restrict__ g, int n_shards,
                                                    int64_t n, ShardBases base, int packed,
                                                    int64_t* __restrict__ i1o, float* __restrict__ d1o,
                                                    int64_t* __restrict__ i2o, float* __restrict__ d2o,
                                                    float ratio, uint8_t* __restrict__ mask) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= n) return;
  float d1 = __builtin_inff(), d2 = __builtin_inff();
  int64_t i1 = -1, i2 = -1;
  auto less = [](float da, int64_t ia, float db, int64_t ib) {
    if (ia < 0) return false;
    if (ib < 0) return true;
    return da < db || (da == db && ia < ib);
  };
  auto push = [&](float d, int64_t i) {
    if (i < 0) return;
    if (less(d, i, d1, i1)) {
      d2 = d1; i2 = i1; d1 = d; i1 = i;
    } else if (less(d, i, d2, i2)) {
      d2 = d; i2 = i;
    }
  };
  const int64_t stride = packed ? n + 1 : n;
  for (int s = 0; s < n_shards; ++s) {
    const ic_top2_rec r = g[(int64_t)s * stride + row];
    const int64_t b = packed ? *reinterpret_cast<const int64_t*>(&g[(int64_t)s * stride + n])
                             : base.b[s];
    push(r.d1, r.i1 < 0 ? -1 : r.i1 + b);
    push(r.d2, r.i2 < 0 ? -1 : r.i2 + b);
  }
  if (i1o) i1o[row] = i1;
  if (d1o) d1o[row] = d1;
  if (i2o) i2o[row] = i2;
  if (d2o) d2o[row] = d2;
  if (mask) mask[row] = (d1 < d2 * ratio) ? 1 : 0;
}

hipError_t launch_top2_merge(const ic_top2_rec* gathered, int n_shards, int64_t n,
                             const int64_t* shard_base_host, int64_t* i1, float* d1, int64_t* i2,
                             float* d2, float ratio, uint8_t* mask, hipStream_t s) {
  if (n_shards < 1 || n_shards > kMaxShards) return hipErrorInvalidValue;
  const int64_t grid = (n + 255) / 256;
  if (grid < 1) return hipSuccess;
  ShardBases sb;
  for (int i = 0; i < kMaxShards; ++i)
    sb.b[i] = (shard_base_host && i < n_shards) ? shard_base_host[i] : 0;
  hipLaunchKernelGGL(k_top2_merge, dim3((unsigned)grid), dim3(256), 0, s, gathered, n_shards, n,
                     sb, shard_base_host ? 0 : 1, i1, d1, i2, d2, ratio, mask);
  return hipGetLastError();
}

}  // namespace ic
