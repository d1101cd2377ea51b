// HBM-bound reductions that follow the assignment: per-image bag-of-words histograms and k-means
// centroid sums.  Both are deterministic: the histogram counts with integer adds in LDS (exact in
// any order), the centroid sums add floating-point values in a fixed order with one writer per
// word between two barriers.
//
// What they replace in the reference (paths relative to the reference tree):
//   np.histogram(code, bins=range(0, K+1))[0] + L1 normalisation   model/features.py:1174-1181
//   scipy _vq.update_cluster_means inside kmeans2                  vision/facerec/face_indexing.py:158
#include "ic_internal.h"

namespace ic {

// ---------------------------------------------------------------------------------------------
// Histogram.  One workgroup of four waves per (image, bin range): the bins live in LDS, every
// thread adds 1 to the bin of each of its codes (ds_add_u32; lanes of one instruction that hold
// the same code are serialised by the hardware, so an image whose descriptors all map to one
// word costs 64 adds per instruction and nothing worse), then the row is normalised and written.
// (An earlier version elected one owner lane per code through tag slots instead of using the
// atomic add: four times the code reads and up to three LDS round trips per 64 codes.)
constexpr int kHistThreads = 256;
constexpr int kHistMaxBins = 32768;  // bins per workgroup: 128 KiB of LDS

__device__ __forceinline__ double block_sum(double v, double* red /*[kHistThreads/64]*/) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kHistThreads / 64; ++w) t += red[w];
  return t;
}

__device__ __forceinline__ double hist_transform(double c, int norm_mode, double power) {
  if (norm_mode != IC_NORM_POWER_L2) return c;
  if (power == 0.5) return sqrt(c);
  return c == 0.0 ? 0.0 : pow(c, power);
}

// scale factor applied to the transformed counts of one row, from the row's reductions
__device__ __forceinline__ double hist_scale(int norm_mode, double sum, double sumsq) {
  if (norm_mode == IC_NORM_L1) return 1.0 / (sum != 0.0 ? sum : 1.0);
  if (norm_mode == IC_NORM_L2 || norm_mode == IC_NORM_POWER_L2) {
    const double nrm = sqrt(sumsq);
    return nrm != 0.0 ? nrm : 1.0;  // divisor in these modes
  }
  return 1.0;
}

__global__ __launch_bounds__(kHistThreads) void k_hist(const int32_t* __restrict__ codes,
                                                       const int64_t* __restrict__ offsets,
                                                       int64_t k, int bins_per_wg, int norm_mode,
                                                       double power, int fused_norm,
                                                       const double* __restrict__ cell_w,
                                                       int64_t cell_k, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned hsm[];
  __shared__ double red[kHistThreads / 64];
  const int tid = threadIdx.x;
  const int64_t img = blockIdx.x;
  const int64_t bin0 = (int64_t)blockIdx.y * bins_per_wg;
  int64_t nb64 = k - bin0;
  if (nb64 > bins_per_wg) nb64 = bins_per_wg;
  const unsigned nb = (unsigned)nb64;
  volatile unsigned* hist = hsm;
  for (unsigned b = tid; b < nb; b += kHistThreads) hist[b] = 0;
  const int64_t beg = offsets[img], end = offsets[img + 1];
  __syncthreads();

  // every thread counts its own codes with an integer add in LDS (exact in any order); four
  // codes per thread are in flight, the next four are fetched before these are counted
  constexpr int kU = 4;
  int in_range = 0;   // this thread's codes that are bins of the row at all (any bin range)
  int32_t nxt[kU];
#pragma unroll
  for (int u = 0; u < kU; ++u) {
    const int64_t i = beg + u * kHistThreads + tid;
    nxt[u] = i < end ? codes[i] : -1;
  }
  for (int64_t base = beg; base < end; base += kU * kHistThreads) {
    int32_t cur[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      cur[u] = nxt[u];
      const int64_t i = base + (kU + u) * kHistThreads + tid;
      nxt[u] = i < end ? codes[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int64_t c = (int64_t)cur[u] - bin0;
      in_range += cur[u] >= 0 && cur[u] < k;
      if (cur[u] >= 0 && c >= 0 && c < (int64_t)nb)
        __hip_atomic_fetch_add(&hsm[(unsigned)c], 1u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();

  double* orow = out + img * k + bin0;
  if (cell_w) {  // spatial pyramid: raw counts scaled by the weight of the bin's cell
    for (unsigned b = tid; b < nb; b += kHistThreads)
      orow[b] = (double)hist[b] * cell_w[(bin0 + b) / cell_k];
    return;
  }
  if (!fused_norm) {  // raw counts; k_hist_normalise finishes rows that span several workgroups
    for (unsigned b = tid; b < nb; b += kHistThreads) orow[b] = (double)hist[b];
    return;
  }
  // Row reductions over the COUNTS (small integers, exact in float64 in any order): the sum of
  // the transformed counts and of their squares.  Most bins of a bag-of-words row are empty, so
  // the transform (a square root or a power) is only taken where a count is not zero.
  double s = 0.0, s2 = 0.0;
  if (norm_mode == IC_NORM_L1) {
    // the sum of the row's counts is the number of its codes: no pass over the bins, and a
    // workgroup that holds only a range of the bins knows the whole row's sum all the same
    s = block_sum((double)in_range, red);
  } else if (norm_mode != IC_NORM_NONE) {
    for (unsigned b = tid; b < nb; b += kHistThreads) {
      const unsigned cnt = hist[b];
      if (cnt != 0) {
        const double y = hist_transform((double)cnt, norm_mode, power);
        s += y;
        s2 += y * y;
      }
    }
    s = block_sum(s, red);
    s2 = block_sum(s2, red);
  }
  const double f = hist_scale(norm_mode, s, s2);
  // two bins per lane and store (16 bytes) when the row allows it
  const bool pair_ok = ((reinterpret_cast<uintptr_t>(orow) & 15) == 0);
  const unsigned nb2 = pair_ok ? (nb & ~1u) : 0u;
  for (unsigned b = 2 * tid; b < nb2; b += 2 * kHistThreads) {
    double y[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const unsigned cnt = hist[b + e];
      y[e] = 0.0;
      if (cnt != 0) {
        const double t = hist_transform((double)cnt, norm_mode, power);
        y[e] = norm_mode == IC_NORM_L1 ? t * f : (norm_mode == IC_NORM_NONE ? t : t / f);
      }
    }
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    f64x2 v = {y[0], y[1]};
    *reinterpret_cast<f64x2*>(orow + b) = v;
  }
  for (unsigned b = nb2 + tid; b < nb; b += kHistThreads) {
    const unsigned cnt = hist[b];
    double y = 0.0;
    if (cnt != 0) {
      const double t = hist_transform((double)cnt, norm_mode, power);
      y = norm_mode == IC_NORM_L1 ? t * f : (norm_mode == IC_NORM_NONE ? t : t / f);
    }
    orow[b] = y;
  }
}

// In-place normalisation of rows of raw counts (rows wider than one workgroup's bin range).
__global__ __launch_bounds__(kHistThreads) void k_hist_normalise(int64_t k, int norm_mode,
                                                                 double power,
                                                                 double* __restrict__ out) {
  __shared__ double red[kHistThreads / 64];
  double* row = out + (int64_t)blockIdx.x * k;
  double s = 0.0, s2 = 0.0;
  for (int64_t b = threadIdx.x; b < k; b += kHistThreads) {
    const double y = hist_transform(row[b], norm_mode, power);
    s += y;
    s2 += y * y;
  }
  s = block_sum(s, red);
  s2 = block_sum(s2, red);
  const double f = hist_scale(norm_mode, s, s2);
  for (int64_t b = threadIdx.x; b < k; b += kHistThreads) {
    const double y = hist_transform(row[b], norm_mode, power);
    row[b] = norm_mode == IC_NORM_L1 ? y * f : y / f;
  }
}

// Once per device (from ic_create, with that device current): the histogram kernel may use the
// dynamic LDS of its widest bin range.
hipError_t reduce_init() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(k_hist),
                             hipFuncAttributeMaxDynamicSharedMemorySize,
                             kHistMaxBins * 4);
}

hipError_t launch_hist(const int32_t* codes, const int64_t* offsets, int64_t n_img, int64_t k,
                       int norm_mode, double power, const double* cell_w, int64_t cell_k,
                       double* out, hipStream_t s) {
  if (n_img <= 0) return hipSuccess;
  int bins_per_wg = k < kHistMaxBins ? (int)k : kHistMaxBins;
  // raw counts and L1 rows need nothing from the other bins of their row (an L1 row's sum is
  // its number of codes), so a few long rows -- one image per call, as the reference's flow
  // issues them -- are cut into bin ranges of their own workgroups instead of one workgroup
  // clearing, scanning and writing 10 000 bins alone (15 -> 7 us)
  const bool row_local = !cell_w && (norm_mode == IC_NORM_NONE || norm_mode == IC_NORM_L1);
  if (row_local && n_img < 64 && k > 1024) {
    const int64_t want = 128 / n_img;                         // workgroups per row
    int64_t per = (k + want - 1) / want;
    per = per < 512 ? 512 : (per + 1) / 2 * 2;                // even: 16-byte stores stay aligned
    if (per < bins_per_wg) bins_per_wg = (int)per;
  }
  const int64_t n_ranges = (k + bins_per_wg - 1) / bins_per_wg;
  if (n_img > 0x7fffffffLL || n_ranges > 65535) return hipErrorInvalidValue;
  const int lds = bins_per_wg * 4;   // see reduce_init()
  const int fused = (n_ranges == 1 || row_local) ? 1 : 0;
  hipLaunchKernelGGL(k_hist, dim3((unsigned)n_img, (unsigned)n_ranges), dim3(kHistThreads), lds, s,
                     codes, offsets, k, bins_per_wg, norm_mode, power, fused, cell_w, cell_k, out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (!fused && norm_mode != IC_NORM_NONE && !cell_w) {
    hipLaunchKernelGGL(k_hist_normalise, dim3((unsigned)n_img), dim3(kHistThreads), 0, s, k,
                       norm_mode, power, out);
    e = hipGetLastError();
  }
  return e;
}

// ---------------------------------------------------------------------------------------------
// Spatial pyramid.  Every keypoint votes once per pyramid level, into the cell it falls in on that
// level; with the cells numbered 0 .. n_cells-1 over all levels this is an ordinary histogram over
// the keys  cell * k + code  with n_levels keys per keypoint, so k_hist does the counting and
// applies the per-cell weight.  Replaces _get_spatial_histogram / compute_spatial_features
// (vision/utils/scene_functions.py:279-373 of the reference).
__global__ __launch_bounds__(256) void k_spatial_keys(const int32_t* __restrict__ codes,
                                                      const int32_t* __restrict__ kp_cells,
                                                      int64_t n_img, int64_t n_kp, int n_levels,
                                                      int64_t k, int32_t* __restrict__ keys,
                                                      int64_t* __restrict__ offsets) {
  const int64_t per_img = n_kp * n_levels;
  const int64_t total = n_img * per_img;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total;
       t += (int64_t)gridDim.x * 256) {
    const int64_t img = t / per_img;
    const int64_t rem = t - img * per_img;
    const int64_t kp = rem / n_levels;
    const int lev = (int)(rem - kp * n_levels);
    const int32_t code = codes[img * n_kp + kp];
    const int32_t cell = kp_cells[(int64_t)lev * n_kp + kp];
    keys[t] = (code >= 0 && code < k && cell >= 0) ? (int32_t)(cell * k + code) : -1;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i <= n_img;
       i += (int64_t)gridDim.x * 256)
    offsets[i] = i * per_img;
}

hipError_t launch_spatial_hist(const int32_t* codes, const int32_t* kp_cells, int64_t n_img,
                               int64_t n_kp, int n_levels, int64_t k, int64_t n_cells,
                               const double* cell_w, int32_t* keys, int64_t* offsets, double* out,
                               hipStream_t s) {
  if (n_img <= 0) return hipSuccess;
  const int64_t total = n_img * n_kp * n_levels;
  int64_t grid = (total + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_spatial_keys, dim3((unsigned)grid), dim3(256), 0, s, codes, kp_cells, n_img,
                     n_kp, n_levels, k, keys, offsets);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return launch_hist(keys, offsets, n_img, n_cells * k, IC_NORM_NONE, 0.5, cell_w, k, out, s);
}

// ---------------------------------------------------------------------------------------------
// Centroid sums, scan-based fallback (used only when the partition table of ic_centroid.hip
// would not fit).  Workgroup (cr, rr) owns clusters [64 cr, 64 cr + 64) over row range rr and keeps
// their float64 sums in LDS.  Wave w of the workgroup owns the clusters with (cluster & 3) == w: it
// scans the range's codes 64 at a time, and for every row of one of its clusters adds the row into
// that cluster's LDS sum (lane l adds components 2l, 2l+1).  Rows are visited in ascending order
// and a cluster has one owner, so the summation order is fixed.  Row ranges are then added in
// ascending order by k_centroid_reduce.
constexpr int kCR = 64;
constexpr int kCentThreads = 256;

__global__ __launch_bounds__(kCentThreads) void k_centroid_partials(
    const float* __restrict__ X, const int32_t* __restrict__ codes, int64_t n, int64_t k, int d,
    int64_t rows_per_range, double* __restrict__ psums, int64_t* __restrict__ pcounts) {
  __shared__ __attribute__((aligned(16))) double sums[kCR][kD];
  __shared__ long long cnt[kCR];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t c0 = (int64_t)blockIdx.x * kCR;
  const int64_t rr = blockIdx.y;
  const int64_t rbeg = rr * rows_per_range;
  int64_t rend = rbeg + rows_per_range;
  if (rend > n) rend = n;

  for (int t = threadIdx.x; t < kCR * kD; t += kCentThreads) (&sums[0][0])[t] = 0.0;
  if (threadIdx.x < kCR) cnt[threadIdx.x] = 0;
  __syncthreads();

  const int c2 = 2 * lane;
  // the scan is latency bound unless several code loads are in flight: kScanUnroll independent
  // 64-code loads are issued before the first is used
  constexpr int kScanUnroll = 8;
  for (int64_t base = rbeg; base < rend; base += 64 * kScanUnroll) {
    int cv[kScanUnroll];
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) {
      const int64_t r = base + u * 64 + lane;
      cv[u] = r < rend ? codes[r] : -1;
    }
#pragma unroll
    for (int u = 0; u < kScanUnroll; ++u) {
      const int64_t c = (int64_t)cv[u] - c0;
      const int lc = (cv[u] >= 0 && c >= 0 && c < kCR && (c & 3) == wave) ? (int)c : -1;
      unsigned long long m = __ballot(lc >= 0);
      while (m) {  // ascending lanes = ascending rows
        const int e = __builtin_ctzll(m);
        m &= m - 1;
        const int cl = __builtin_amdgcn_readlane(lc, e);
        const float* xrow = X + (base + u * 64 + e) * d;
        double a0 = 0.0, a1 = 0.0;
        if (c2 < d) a0 = (double)xrow[c2];
        if (c2 + 1 < d) a1 = (double)xrow[c2 + 1];
        sums[cl][c2] += a0;
        sums[cl][c2 + 1] += a1;
        if (lane == 0) cnt[cl] += 1;
      }
    }
  }
  __syncthreads();

  int64_t ncl = k - c0;
  if (ncl > kCR) ncl = kCR;
  double* ps = psums + (rr * k + c0) * d;
  for (int t = threadIdx.x; t < (int)ncl * d; t += kCentThreads) ps[t] = sums[t / d][t % d];
  if (threadIdx.x < ncl) pcounts[rr * k + c0 + threadIdx.x] = cnt[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_centroid_reduce(const double* __restrict__ psums,
                                                         const int64_t* __restrict__ pcounts,
                                                         int64_t kd, int64_t k, int n_rr,
                                                         double* __restrict__ sums,
                                                         int64_t* __restrict__ counts) {
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < kd; t += (int64_t)gridDim.x * 256) {
    double a = 0.0;
    for (int r = 0; r < n_rr; ++r) a += psums[(int64_t)r * kd + t];
    sums[t] = a;
    if (t < k) {
      int64_t c = 0;
      for (int r = 0; r < n_rr; ++r) c += pcounts[(int64_t)r * k + t];
      counts[t] = c;
    }
  }
}

void centroid_plan_scan(int64_t n, int64_t k, int num_cus, int* n_rr) {
  const int64_t n_cr = (k + kCR - 1) / kCR;
  int64_t rr = (4LL * num_cus + n_cr - 1) / n_cr;
  const int64_t max_rr = (n + 4095) / 4096;
  if (rr > max_rr) rr = max_rr;
  if (rr > 65535) rr = 65535;
  if (rr < 1) rr = 1;
  *n_rr = (int)rr;
}

hipError_t launch_centroid_reduce(const double* part_sums, const int64_t* part_counts, int64_t k,
                                  int d, int n_ranges, double* sums, int64_t* counts,
                                  hipStream_t s) {
  const int64_t kd = k * d;
  int64_t grid = (kd + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_centroid_reduce, dim3((unsigned)grid), dim3(256), 0, s, part_sums,
                     part_counts, kd, k, n_ranges, sums, counts);
  return hipGetLastError();
}

hipError_t launch_centroid_partials_scan(const float* x, const int32_t* codes, int64_t n,
                                         int64_t k, int d, int n_rr, double* part_sums,
                                         int64_t* part_counts, double* sums, int64_t* counts,
                                         hipStream_t s) {
  const int64_t n_cr = (k + kCR - 1) / kCR;
  if (n_cr > 0x7fffffffLL) return hipErrorInvalidValue;
  int64_t rows_per = (n + n_rr - 1) / n_rr;
  rows_per = (rows_per + 63) / 64 * 64;
  if (rows_per < 64) rows_per = 64;
  double* ps = n_rr == 1 ? sums : part_sums;
  int64_t* pc = n_rr == 1 ? counts : part_counts;
  hipLaunchKernelGGL(k_centroid_partials, dim3((unsigned)n_cr, (unsigned)n_rr), dim3(kCentThreads),
                     0, s, x, codes, n, k, d, rows_per, ps, pc);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || n_rr == 1) return e;
  return launch_centroid_reduce(part_sums, part_counts, k, d, n_rr, sums, counts, s);
}

__global__ __launch_bounds__(256) void k_centroid_finalize(const double* __restrict__ sums,
                                                           const int64_t* __restrict__ counts,
                                                           const float* __restrict__ prev,
                                                           int64_t k, int d, float* __restrict__ out,
                                                           uint8_t* __restrict__ has_members) {
  const int64_t kd = k * d;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < kd; t += (int64_t)gridDim.x * 256) {
    const int64_t j = t / d;
    const int64_t c = counts[j];
    out[t] = c != 0 ? (float)(sums[t] / (double)c) : prev[t];
    if (has_members && t % d == 0) has_members[j] = c != 0 ? 1 : 0;
  }
}

hipError_t launch_centroid_finalize(const double* sums, const int64_t* counts, const float* prev_c,
                                    int64_t k, int d, float* new_c, uint8_t* has_members,
                                    hipStream_t s) {
  int64_t grid = (k * d + 255) / 256;
  if (grid > 8192) grid = 8192;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(k_centroid_finalize, dim3((unsigned)grid), dim3(256), 0, s, sums, counts,
                     prev_c, k, d, new_c, has_members);
  return hipGetLastError();
}

}  // namespace ic
