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
//     codebook, followed by (row tile, K-split) pairs for the last, partly filled round; every
//     workgroup writes one (value, index) x 2 record per row, k_rescore merges the splits.
//   * ranking is gated twice (tile minimum, then per register) because on gfx950 the fp32 MFMA
//     and the vector ALU never co-execute: every ranking instruction costs MFMA time.
#include <cstdlib>

#include "ic_internal.h"

namespace ic {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

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

// K-split that finishes `row_tiles` row tiles soonest on `num_cus` workgroup slots.
static int choose_nsplit(int64_t row_tiles, int64_t stages, int num_cus) {
  const double overhead = 2.0;  // x-tile load + pipeline fill, in units of one stage of work
  int best = 1;
  double best_cost = 1e300;
  const int64_t max_ns = stages < 256 ? stages : 256;
  for (int64_t ns = 1; ns <= max_ns; ++ns) {
    const int64_t per = (stages + ns - 1) / ns;
    if ((stages + per - 1) / per != ns) continue;  // same per-split length as a smaller ns
    const int64_t items = row_tiles * ns;
    const int64_t rounds = (items + num_cus - 1) / num_cus;
    const double cost = (double)rounds * ((double)per + overhead) * (1.0 + 0.002 * (double)ns);
    if (cost < best_cost * 0.999) {
      best_cost = cost;
      best = (int)ns;
    }
  }
  return best;
}

// One workgroup occupies a CU.  Row tiles that fill whole rounds of the machine are searched over
// the entire codebook by one workgroup each (the longer a lane's scan, the rarer a tile improves
// its top-2, so ranking gets cheaper, and X is read once); only the row tiles of the last, partly
// filled round are split along K so that their pieces spread over all CUs.
void assign_plan(int64_t n, int64_t k, int num_cus, AssignPlan* p) {
  // IC_AMD_GEOMETRY forces a shape (1 = two 8-wave workgroups per CU, 2 = one of 16; tuning aid)
  static const int geom = [] {
    const char* v = getenv("IC_AMD_GEOMETRY");
    return v ? atoi(v) : 0;
  }();
  const bool small = geom == 1 || (geom != 2 && k < kSmallCodebook);
  p->waves = small ? 8 : 16;
  p->stage_k = small ? 64 : 128;
  p->rows_per_wg = kXT * p->waves * 32;
  const int slots = num_cus * (small ? 2 : 1);  // workgroups resident at once
  const int64_t stages = (k + p->stage_k - 1) / p->stage_k;
  p->row_tiles = (n + p->rows_per_wg - 1) / p->rows_per_wg;
  // IC_AMD_NSPLIT forces one split for every row tile (tuning aid)
  static const int forced = [] {
    const char* v = getenv("IC_AMD_NSPLIT");
    return v ? atoi(v) : 0;
  }();
  int ns;
  if (forced > 0) {
    p->full_tiles = 0;
    ns = forced;
  } else {
    p->full_tiles = p->row_tiles / slots * slots;
    const int64_t rem = p->row_tiles - p->full_tiles;
    ns = rem > 0 ? choose_nsplit(rem, stages, slots) : 1;
  }
  if (ns > stages) ns = (int)stages;
  if (ns < 1) ns = 1;
  p->stages_per_split = (int)((stages + ns - 1) / ns);
  p->nsplit = (int)((stages + p->stages_per_split - 1) / p->stages_per_split);
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
template <int NW, int SK>
__device__ __forceinline__ void issue_stage(const float* __restrict__ C, int K,
                                            const float* __restrict__ cnp, int st, char* sb,
                                            int wave, int lane) {
  // wave-uniform 64-bit base + 32-bit per-lane byte offset (keeps the address math off the VALU)
  const char* base = reinterpret_cast<const char*>(C) + (size_t)st * (SK * kD * 4);
  const int last = K - 1 - st * SK;  // last valid row of this stage (>= 0)
  // opaque copy: the offsets below are recomputed per call (a few VALU ops) instead of being
  // hoisted out of the sweep loop and held in registers the MFMA operands need
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
      const unsigned voff = (unsigned)r * 512u + (unsigned)c * 16u;
      glds16(reinterpret_cast<const float*>(base + voff), sb + m * 1024);
    }
  }
#pragma unroll
  for (int part = 0; part < SK / 64; ++part)
    if (wave == NW - 1 - part)
      glds4(cnp + (size_t)st * SK + part * 64 + lane, sb + Stage<SK>::kTileBytes + part * 256);
}

struct Top2 {
  float d1, d2;
  int i1, i2;
};

struct FinalOutExact {   // outputs of k_exact_top2 (any pointer may be null)
  int32_t* i1;
  float* d1;
  int32_t* i2;
  float* d2;
  float ratio;
  uint8_t* mask;
};

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
// doing less of it: the minimum of the tile's 16 values (8 v_min3) is compared with the running
// second best, and when no lane of the wave can improve its top-2 the update is skipped.
template <int XT>
__device__ __forceinline__ void tile_epilogue_skip(const f32x16 (&acc)[XT], Top2 (&st)[XT],
                                                   int encbase) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    const f32x16& a = acc[xt];
    const float m = min3f(min3f(min3f(a[0], a[1], a[2]), min3f(a[3], a[4], a[5]), min3f(a[6], a[7], a[8])),
                          min3f(min3f(a[9], a[10], a[11]), min3f(a[12], a[13], a[14]), a[15]),
                          __builtin_inff());
    if (__builtin_amdgcn_ballot_w64(m < st[xt].d2) != 0) {
      // rank inside the tile with the register number (an inline constant) as index; -1 / -2
      // stand for "the previous best / second best", resolved once after the 16 updates
      float d1 = st[xt].d1, d2 = st[xt].d2;
      int l1 = -1, l2 = -2;
#pragma unroll
      for (int r = 0; r < 16; ++r) top2_update(d1, d2, l1, l2, a[r], r);
      const int o1 = st[xt].i1, o2 = st[xt].i2;
      st[xt].d1 = d1;
      st[xt].d2 = d2;
      st[xt].i1 = l1 >= 0 ? encbase + l1 : o1;
      st[xt].i2 = l2 >= 0 ? encbase + l2 : (l2 == -1 ? o1 : o2);
    }
  }
}

// The same, one level finer: inside a tile that some lane can improve, a register's candidate is
// ranked only when some lane's value in THAT register beats its running second best (16 compares
// and wave-uniform branches instead of 16 seven-instruction updates; a lane improves on about
// 2/t of the tiles, so past the first few tiles most registers are skipped).
template <int XT>
__device__ __forceinline__ void tile_epilogue_skip2(const f32x16 (&acc)[XT], Top2 (&st)[XT],
                                                    int encbase) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    const f32x16& a = acc[xt];
    const float m = min3f(min3f(min3f(a[0], a[1], a[2]), min3f(a[3], a[4], a[5]), min3f(a[6], a[7], a[8])),
                          min3f(min3f(a[9], a[10], a[11]), min3f(a[12], a[13], a[14]), a[15]),
                          __builtin_inff());
    if (__builtin_amdgcn_ballot_w64(m < st[xt].d2) != 0) {
      float d1 = st[xt].d1, d2 = st[xt].d2;
      int l1 = -1, l2 = -2;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(a[r] < d2) != 0, 0))
          top2_update(d1, d2, l1, l2, a[r], r);
      const int o1 = st[xt].i1, o2 = st[xt].i2;
      st[xt].d1 = d1;
      st[xt].d2 = d2;
      st[xt].i1 = l1 >= 0 ? encbase + l1 : o1;
      st[xt].i2 = l2 >= 0 ? encbase + l2 : (l2 == -1 ? o1 : o2);
    }
  }
}

// The same with a tighter gate.  The two lanes that hold one descriptor (lane and lane + 32) rank
// different halves of every tile's codewords, and each lane's own second best is a loose bound for
// the descriptor: once per stage the pair exchanges its lists, and a candidate is ranked only when
// it is below min(own second best, second best of the pair's four entries).  The pair's value is
// raised by two units in the last place first, so a candidate that ties it is still ranked and the
// final merge by (value, index) sees it.  About half as many candidates get past the gate.
template <int XT>
__device__ __forceinline__ void tile_epilogue_skip3(const f32x16 (&acc)[XT], Top2 (&st)[XT],
                                                    float (&gate)[XT], int encbase) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    const f32x16& a = acc[xt];
    const float m = min3f(min3f(min3f(a[0], a[1], a[2]), min3f(a[3], a[4], a[5]), min3f(a[6], a[7], a[8])),
                          min3f(min3f(a[9], a[10], a[11]), min3f(a[12], a[13], a[14]), a[15]),
                          __builtin_inff());
    if (__builtin_amdgcn_ballot_w64(m < gate[xt]) != 0) {
      float d1 = st[xt].d1, d2 = st[xt].d2, g = gate[xt];
      int l1 = -1, l2 = -2;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(a[r] < g) != 0, 0)) {
          // a lane whose value is below its own second best but not below its gate ranks it too:
          // needless, never wrong
          top2_update(d1, d2, l1, l2, a[r], r);
          g = __builtin_fminf(g, d2);
        }
      const int o1 = st[xt].i1, o2 = st[xt].i2;
      st[xt].d1 = d1;
      st[xt].d2 = d2;
      st[xt].i1 = l1 >= 0 ? encbase + l1 : o1;
      st[xt].i2 = l2 >= 0 ? encbase + l2 : (l2 == -1 ? o1 : o2);
      gate[xt] = g;
    }
  }
}

template <int XT>
__device__ __forceinline__ void sync_gate(const Top2 (&st)[XT], float (&gate)[XT]) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    const float od1 = __shfl_xor(st[xt].d1, 32);
    const float od2 = __shfl_xor(st[xt].d2, 32);
    // second smallest of {d1, d2, od1, od2} given d1 <= d2 and od1 <= od2
    float s2 = __builtin_fminf(__builtin_fmaxf(st[xt].d1, od1), __builtin_fminf(st[xt].d2, od2));
    s2 = __builtin_fmaf(__builtin_fmaxf(__builtin_fabsf(s2), 1e-30f), 0x1p-22f, s2);
    gate[xt] = __builtin_fminf(st[xt].d2, s2);
  }
}

template <int XT>
__device__ __forceinline__ void tile_epilogue(const f32x16 (&acc)[XT], Top2 (&st)[XT], int encbase) {
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
#pragma unroll
    for (int r = 0; r < 16; ++r)
      top2_update(st[xt].d1, st[xt].d2, st[xt].i1, st[xt].i2, acc[xt][r], encbase + r);
  }
}

// Slot of row r's record from K-split sp.  Split 0 of every row comes first ([N] records); only
// the rows from split_row0 on (the last, partly filled round of row tiles) have further splits,
// stored behind as [nsplit - 1][N - split_row0].
__device__ __forceinline__ int64_t part_slot(int sp, int64_t r, int64_t N, int64_t split_row0) {
  return sp == 0 ? r : N + (int64_t)(sp - 1) * (N - split_row0) + (r - split_row0);
}

template <int XT, int NW, int SK, int SKIP>
__global__ __launch_bounds__(NW * 64) void k_top2_partial(
    const float* __restrict__ X, int64_t N, const float* __restrict__ C, int K,
    const float* __restrict__ cnp, int stages_total, int stages_per_split, int full_tiles,
    int split_tiles, const float* __restrict__ thr, ic_top2_rec* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;  // descriptor (B column / result column) and codeword row (A row)
  const int h = lane >> 5;  // which half of the 128 components this lane multiplies

  // the first full_tiles workgroups scan the whole codebook; the rest are (row tile, K-split) pairs
  int rt = blockIdx.x, sp = 0, st0 = 0, st1 = stages_total;
  if ((int)blockIdx.x >= full_tiles) {
    const int idx = (int)blockIdx.x - full_tiles;
    rt = full_tiles + idx % split_tiles;
    sp = idx / split_tiles;
    st0 = sp * stages_per_split;
    st1 = st0 + stages_per_split;
    st1 = st1 < stages_total ? st1 : stages_total;
  }

  // descriptors of this wave, scaled by -2 (exact), 64 components per lane.  Their loads go out
  // first, then the first two stages of the codebook.
  const int64_t row0 = (int64_t)rt * (XT * NW * 32) + (int64_t)wave * (XT * 32);
  f32x4 xv[XT][16];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    int64_t r = row0 + xt * 32 + j;
    r = r < N ? r : N - 1;
    const f32x4* p = reinterpret_cast<const f32x4*>(X + r * kD + h * 4);
#pragma unroll
    for (int q = 0; q < 16; ++q) xv[xt][q] = p[2 * q];
  }
  issue_stage<NW, SK>(C, K, cnp, st0, smem, wave, lane);
  const bool two = st0 + 1 < st1;
  if (two) issue_stage<NW, SK>(C, K, cnp, st0 + 1, smem + Stage<SK>::kBytes, wave, lane);
  float xr[XT][64];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      xr[xt][q * 4 + 0] = -2.0f * xv[xt][q][0];
      xr[xt][q * 4 + 1] = -2.0f * xv[xt][q][1];
      xr[xt][q * 4 + 2] = -2.0f * xv[xt][q][2];
      xr[xt][q * 4 + 3] = -2.0f * xv[xt][q][3];
    }
  }

  Top2 best[XT];
  float gate[XT];
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    // with a caller-supplied bound only candidates below it are ranked at all (index -1 = none)
    float start = __builtin_inff();
    if (thr) {
      int64_t r = row0 + xt * 32 + j;
      start = thr[r < N ? r : N - 1];
    }
    best[xt].d1 = start;
    best[xt].d2 = start;
    best[xt].i1 = -1;
    best[xt].i2 = -1;
    gate[xt] = start;
  }

  // loads return in order: with the second stage in flight (at least 8 loads per wave, issued
  // last) the rows and the first stage are complete once at most 8 loads are outstanding
  constexpr int kMinLoads = Stage<SK>::kGlds / NW;
  if (two) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMinLoads) : "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  f32x16 acc[XT];
  for (int st = st0; st < st1; ++st) {
    char* sb = smem + ((st - st0) & 1) * Stage<SK>::kBytes;
#pragma unroll
    for (int t = 0; t < Stage<SK>::kTiles; ++t) {
      // tiles that lie wholly past the end of the codebook (padding of the last stage) are skipped
      const bool live = (st * Stage<SK>::kTiles + t) * kTileK < K;
      if (live)
        compute_tile<XT>(sb + t * kTileK * kRowBytes, sb + Stage<SK>::kTileBytes + t * kTileK * 4, xr,
                         acc, j, h);
      if (t == Stage<SK>::kTiles - 1) {
        // every wave is done reading this buffer once it passes the barrier; the stage issued one
        // iteration ago has landed in the other buffer by then
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (st + 2 < st1) issue_stage<NW, SK>(C, K, cnp, st + 2, sb, wave, lane);
        if constexpr (SKIP == 3) sync_gate<XT>(best, gate);
      }
      if (live) {
        const int encbase = (st * Stage<SK>::kTiles + t) * 16;
        if constexpr (SKIP == 3) tile_epilogue_skip3<XT>(acc, best, gate, encbase);
        else if constexpr (SKIP == 2) tile_epilogue_skip2<XT>(acc, best, encbase);
        else if constexpr (SKIP == 1) tile_epilogue_skip<XT>(acc, best, encbase);
        else tile_epilogue<XT>(acc, best, encbase);
      }
    }
  }

  // decode (tile, register) -> codeword index, fold the two lane halves, store
#pragma unroll
  for (int xt = 0; xt < XT; ++xt) {
    Top2 b = best[xt];
    auto decode = [h](int enc) {
      if (enc < 0) return -1;
      const int t = enc >> 4, r = enc & 15;
      return t * kTileK + 8 * (r >> 2) + 4 * h + (r & 3);
    };
    b.i1 = decode(b.i1);
    b.i2 = decode(b.i2);
    const float od1 = __shfl_xor(b.d1, 32);
    const float od2 = __shfl_xor(b.d2, 32);
    const int oi1 = __shfl_xor(b.i1, 32);
    const int oi2 = __shfl_xor(b.i2, 32);
    top2_merge_pairs(b.d1, b.i1, b.d2, b.i2, od1, oi1, od2, oi2);
    const int64_t r = row0 + xt * 32 + j;
    if (h == 0 && r < N) {
      ic_top2_rec rec;
      rec.d1 = b.d1; rec.i1 = b.i1; rec.d2 = b.d2; rec.i2 = b.i2;
      part[part_slot(sp, r, N, (int64_t)full_tiles * (XT * NW * 32))] = rec;
    }
  }
}

template <int NW, int SK>
static hipError_t launch_geometry(const float* x, int64_t n, const float* c, int64_t k,
                                  const float* cn_pad, const AssignPlan& plan, const float* thr,
                                  ic_top2_rec* part, hipStream_t s) {
  const int64_t stages = (k + SK - 1) / SK;
  const int64_t split_tiles = plan.row_tiles - plan.full_tiles;
  // IC_AMD_VARIANT selects the ranking variant for A/B timing in one build: 0 = rank every
  // candidate, 1 = skip tiles no lane can improve, 2 (default) = also skip registers, 3 = the
  // same against the bound the two lanes of a descriptor share (fewer ranked candidates, but with
  // four waves per SIMD the exchange costs what it saves: 89.9 vs 89.8 % at 10 000 codewords,
  // 84.3 vs 85.3 % at 1000)
  static const int variant = [] {
    const char* v = getenv("IC_AMD_VARIANT");
    return v ? atoi(v) : 2;
  }();
  auto kern = variant == 0 ? k_top2_partial<kXT, NW, SK, 0>
              : variant == 1 ? k_top2_partial<kXT, NW, SK, 1>
              : variant == 3 ? k_top2_partial<kXT, NW, SK, 3> : k_top2_partial<kXT, NW, SK, 2>;
  const int lds = 2 * Stage<SK>::kBytes;   // allowed by assign_init() on this device
  const int64_t grid = plan.full_tiles + split_tiles * plan.nsplit;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NW * 64), lds, s, x, n, c, (int)k, cn_pad,
                     (int)stages, plan.stages_per_split, (int)plan.full_tiles, (int)split_tiles,
                     thr, part);
  return hipGetLastError();
}

template <int NW, int SK>
static hipError_t allow_lds() {
  const int lds = 2 * Stage<SK>::kBytes;
  const void* kerns[4] = {reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, 0>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, 1>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, 2>),
                          reinterpret_cast<const void*>(k_top2_partial<kXT, NW, SK, 3>)};
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
  return allow_lds<16, 128>();
}

int64_t assign_part_records(const AssignPlan& p, int64_t n) {
  int64_t split_row0 = p.full_tiles * (int64_t)p.rows_per_wg;
  if (split_row0 > n) split_row0 = n;
  return n + (int64_t)(p.nsplit - 1) * (n - split_row0);
}

hipError_t launch_top2_partial(const float* x, int64_t n, const float* c, int64_t k,
                               const float* cn_pad, const AssignPlan& plan, const float* thr,
                               ic_top2_rec* part, hipStream_t s) {
  if (plan.waves == 8 && plan.stage_k == 64)
    return launch_geometry<8, 64>(x, n, c, k, cn_pad, plan, thr, part, s);
  if (plan.waves == 16 && plan.stage_k == 128)
    return launch_geometry<16, 128>(x, n, c, k, cn_pad, plan, thr, part, s);
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

__device__ __forceinline__ float sqdiff8(const f32x4& x0, const f32x4& x1, const float* crow) {
  const f32x4 c0 = *reinterpret_cast<const f32x4*>(crow);
  const f32x4 c1 = *reinterpret_cast<const f32x4*>(crow + 4);
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

__global__ __launch_bounds__(256) void k_rescore(const float* __restrict__ X, int64_t N,
                                                 const float* __restrict__ C,
                                                 const ic_top2_rec* __restrict__ part, int nsplit,
                                                 int64_t split_row0,
                                                 int32_t* __restrict__ i1o, float* __restrict__ d1o,
                                                 int32_t* __restrict__ i2o, float* __restrict__ d2o,
                                                 float ratio, uint8_t* __restrict__ mask,
                                                 ic_top2_rec* __restrict__ reco) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < N; row += stride) {
    ic_top2_rec b = part[row];
    const int ns = row >= split_row0 ? nsplit : 1;   // rows below split_row0 were not split
    for (int s = 1; s < ns; ++s) {
      const ic_top2_rec o = part[part_slot(s, row, N, split_row0)];
      top2_merge_pairs(b.d1, b.i1, b.d2, b.i2, o.d1, o.i1, o.d2, o.i2);
    }
    const float* xrow = X + row * kD + sub * 8;
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(xrow);
    const f32x4 x1 = *reinterpret_cast<const f32x4*>(xrow + 4);
    float s1 = __builtin_inff(), s2 = __builtin_inff();
    if (b.i1 >= 0) s1 = group16_sum(sqdiff8(x0, x1, C + (int64_t)b.i1 * kD + sub * 8));
    if (b.i2 >= 0) s2 = group16_sum(sqdiff8(x0, x1, C + (int64_t)b.i2 * kD + sub * 8));
    int a1 = b.i1, a2 = b.i2;
    if (a2 >= 0 && lex_less(s2, a2, s1, a1)) {
      const float ts = s1; s1 = s2; s2 = ts;
      const int ti = a1; a1 = a2; a2 = ti;
    }
    const float e1 = __builtin_sqrtf(s1), e2 = __builtin_sqrtf(s2);
    if (sub == 0) {
      if (i1o) i1o[row] = a1;
      if (d1o) d1o[row] = e1;
      if (i2o) i2o[row] = a2;
      if (d2o) d2o[row] = e2;
      if (mask) mask[row] = (e1 < e2 * ratio) ? 1 : 0;
      if (reco) {
        ic_top2_rec r;
        r.d1 = e1; r.i1 = a1; r.d2 = e2; r.i2 = a2;
        reco[row] = r;
      }
    }
  }
}

hipError_t launch_rescore(const float* x, int64_t n, const float* c, const ic_top2_rec* part,
                          const AssignPlan& plan, int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                          uint8_t* mask, ic_top2_rec* rec, hipStream_t s) {
  int64_t grid = (n + 15) / 16;
  if (grid > 16384) grid = 16384;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_rescore, dim3((unsigned)grid), dim3(256), 0, s, x, n, c, part, plan.nsplit,
                     plan.full_tiles * (int64_t)plan.rows_per_wg, i1, d1, i2, d2, ratio, mask, rec);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Upper bound, per row, on the value |c|^2 - 2 x.c that the distance kernel will compute for the
// row's nearest codeword, from a guess `hint[row]` of that codeword (k-means: the label of the
// previous iteration).  thr = v_hint + 3 B with B = 129 * 2^-24 * (|c|^2 + 2 |x| |c|), a bound
// on the rounding error of either way of accumulating the 129 terms of v; the true nearest
// codeword has exact v <= exact v_hint, so both it and the hinted codeword rank below thr.
__global__ __launch_bounds__(256) void k_hint_threshold(const float* __restrict__ X, int64_t N,
                                                        const float* __restrict__ C, int64_t K,
                                                        const float* __restrict__ cn,
                                                        const int32_t* __restrict__ hint,
                                                        float* __restrict__ thr) {
  const int sub = threadIdx.x & 15;
  const int64_t stride = (int64_t)gridDim.x * 16;
  for (int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); row < N; row += stride) {
    const int hraw = hint[row];
    const bool ok = hraw >= 0 && hraw < K;
    const int hh = ok ? hraw : 0;
    const float* xrow = X + row * kD + sub * 8;
    const float* crow = C + (int64_t)hh * kD + sub * 8;
    float dot = 0.f, xn = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      dot = __builtin_fmaf(xrow[e], crow[e], dot);
      xn = __builtin_fmaf(xrow[e], xrow[e], xn);
    }
    dot = group16_sum(dot);
    xn = group16_sum(xn);
    if (sub == 0) {
      const float c2 = cn[hh];
      const float v = c2 - 2.f * dot;
      const float band = 7.689e-6f * (c2 + 2.f * __builtin_sqrtf(xn * c2));
      thr[row] = ok ? v + 3.f * band + 1e-30f : __builtin_inff();
    }
  }
}

hipError_t launch_hint_threshold(const float* x, int64_t n, const float* c, int64_t k,
                                 const float* cn_pad, const int32_t* hint, float* thr,
                                 hipStream_t s) {
  int64_t grid = (n + 15) / 16;
  if (grid > 16384) grid = 16384;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_hint_threshold, dim3((unsigned)grid), dim3(256), 0, s, x, n, c, k, cn_pad,
                     hint, thr);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Direct search: sum (x - c)^2 in float32 for every (row, codeword) pair, no norm expansion and
// therefore no cancellation.  Twice the arithmetic of the matrix-core kernel (a subtraction and an FMA per component) plus the reads, four times its run time, on the vector ALU; it
// exists as the on-device checker for sizes no CPU oracle can handle and for callers that want
// distances free of the expansion's rounding.  16 rows per workgroup live in LDS (every lane
// reads the same row element: a broadcast), every thread walks codewords t, t+256, ... keeping a
// top-2 per row, and the 256 partial top-2s are folded by (distance, index).
constexpr int kExactRows = 16;

__global__ __launch_bounds__(256) void k_exact_top2(const float* __restrict__ X, int64_t N,
                                                    const float* __restrict__ C, int K,
                                                    FinalOutExact fin) {
  __shared__ __attribute__((aligned(16))) float xs[kExactRows][kD];
  __shared__ float rd1[4][kExactRows], rd2[4][kExactRows];
  __shared__ int ri1[4][kExactRows], ri2[4][kExactRows];
  const int64_t r0 = (int64_t)blockIdx.x * kExactRows;
  for (int e = threadIdx.x; e < kExactRows * kD; e += 256) {
    const int64_t r = r0 + e / kD;
    xs[e / kD][e % kD] = X[(r < N ? r : N - 1) * kD + e % kD];
  }
  __syncthreads();
  float d1[kExactRows], d2[kExactRows];
  int i1[kExactRows], i2[kExactRows];
#pragma unroll
  for (int r = 0; r < kExactRows; ++r) {
    d1[r] = d2[r] = __builtin_inff();
    i1[r] = i2[r] = -1;
  }
  for (int j = threadIdx.x; j < K; j += 256) {
    const float* crow = C + (int64_t)j * kD;
    float acc[kExactRows];
#pragma unroll
    for (int r = 0; r < kExactRows; ++r) acc[r] = 0.f;
#pragma unroll 1
    for (int ch = 0; ch < kD; ch += 32) {
      f32x4 cv[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) cv[q] = *reinterpret_cast<const f32x4*>(crow + ch + q * 4);
#pragma unroll
      for (int r = 0; r < kExactRows; ++r) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[r][ch + q * 4]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float t = xv[e] - cv[q][e];
            acc[r] = __builtin_fmaf(t, t, acc[r]);
          }
        }
      }
    }
#pragma unroll
    for (int r = 0; r < kExactRows; ++r) top2_update(d1[r], d2[r], i1[r], i2[r], acc[r], j);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int r = 0; r < kExactRows; ++r) {
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
    for (int w = 1; w < 4; ++w) top2_merge_pairs(a1, j1, a2, j2, rd1[w][r], ri1[w][r], rd2[w][r], ri2[w][r]);
    const int64_t row = r0 + r;
    if (row < N) {
      const float e1 = __builtin_sqrtf(a1), e2 = __builtin_sqrtf(a2);
      if (fin.i1) fin.i1[row] = j1;
      if (fin.d1) fin.d1[row] = e1;
      if (fin.i2) fin.i2[row] = j2;
      if (fin.d2) fin.d2[row] = e2;
      if (fin.mask) fin.mask[row] = (e1 < e2 * fin.ratio) ? 1 : 0;
    }
  }
}

hipError_t launch_exact_top2(const float* x, int64_t n, const float* c, int64_t k, int32_t* i1,
                             float* d1, int32_t* i2, float* d2, float ratio, uint8_t* mask,
                             hipStream_t s) {
  const int64_t grid = (n + kExactRows - 1) / kExactRows;
  if (grid <= 0 || grid > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_exact_top2, dim3((unsigned)grid), dim3(256), 0, s, x, n, c, (int)k,
                     FinalOutExact{i1, d1, i2, d2, ratio, mask});
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pad_rows(const float* __restrict__ src, int64_t n, int d,
                                                  float* __restrict__ dst) {
  const int64_t total = n * kD;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * 256) {
    const int64_t r = t / kD;
    const int cidx = (int)(t % kD);
    dst[t] = cidx < d ? src[r * d + cidx] : 0.f;
  }
}

hipError_t launch_pad_rows(const float* src, int64_t n, int d, float* dst, hipStream_t s) {
  int64_t grid = (n * kD + 255) / 256;
  if (grid > 65536) grid = 65536;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_pad_rows, dim3((unsigned)grid), dim3(256), 0, s, src, n, d, dst);
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

__global__ __launch_bounds__(256) void k_copy_pad_inf(const float* __restrict__ src, int64_t k,
                                                      int64_t kpad, float* __restrict__ dst) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < kpad; t += (int64_t)gridDim.x * 256)
    dst[t] = t < k ? src[t] : __builtin_inff();
}

hipError_t launch_codebook_norms(const float* c, int64_t k, int64_t kpad, const float* c_sqnorm,
                                 float* cn_pad, hipStream_t s) {
  if (c_sqnorm) {
    int64_t grid = (kpad + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_copy_pad_inf, dim3((unsigned)grid), dim3(256), 0, s, c_sqnorm, k, kpad,
                       cn_pad);
    return hipGetLastError();
  }
  int64_t grid = (kpad + 15) / 16;
  if (grid > 16384) grid = 16384;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_row_sqnorm, dim3((unsigned)grid), dim3(256), 0, s, c, k, kD, kpad, cn_pad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// One thread per row: fold n_shards records (already exact Euclidean distances) by
// (distance, global index).  The shard bases travel as a kernel argument.
struct ShardBases {
  int64_t b[kMaxShards];
};

__global__ __launch_bounds__(256) void k_top2_merge(const ic_top2_rec* __restrict__ g, int n_shards,
                                                    int64_t n, ShardBases base,
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
  for (int s = 0; s < n_shards; ++s) {
    const ic_top2_rec r = g[(int64_t)s * n + row];
    const int64_t b = base.b[s];
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
  for (int i = 0; i < kMaxShards; ++i) sb.b[i] = i < n_shards ? shard_base_host[i] : 0;
  hipLaunchKernelGGL(k_top2_merge, dim3((unsigned)grid), dim3(256), 0, s, gathered, n_shards, n,
                     sb, i1, d1, i2, d2, ratio, mask);
  return hipGetLastError();
}

}  // namespace ic
