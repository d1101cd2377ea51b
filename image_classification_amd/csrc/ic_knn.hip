// Brute-force k nearest rows in a feature space of any width under a histogram metric
// (Euclidean, L1, histogram intersection, chi-square).  This is the shortlist search over
// bag-of-words histograms (vision/logo_recognition/knn_training.py:24-60 trains it,
// model/classifiers.py:195-197 queries it); none of these metrics except the Euclidean one is a
// matrix product, so the kernels are plain VALU work staged through LDS:
//
//   k_pairdist<METRIC>  64 x 64 tile of the distance matrix per workgroup, 4 x 4 per lane,
//                       64 components per LDS stage (the next one prefetched into registers),
//                       f32 inside a stage, f64 across stages
//   k_topk_rows         one workgroup per search row: k rounds of "smallest (distance, index)
//                       pair above the previous one", so ties resolve to the lowest index
#include "ic_internal.h"

namespace ic {

namespace {

constexpr int kTile = 64;      // rows of either operand per workgroup tile
constexpr int kChunk = 64;     // components per LDS stage
constexpr int kLdsStride = 68; // floats per staged component (64 rows + pad, keeps 16-byte alignment)
constexpr int kPairThreads = 256;
constexpr int kTopkThreads = 256;

template <int METRIC>
__device__ __forceinline__ float pair_term(float a, float b) {
  if (METRIC == IC_METRIC_L1) return fabsf(a - b);
  if (METRIC == IC_METRIC_INTERSECTION) return fminf(a, b);
  if (METRIC == IC_METRIC_CHISQR) {
    // OpenCV's CV_COMP_CHISQR: bins whose first operand is (numerically) zero are skipped
    const float t = a - b;
    return fabsf(a) > 2.2204460492503131e-16f ? t * t / a : 0.f;
  }
  const float t = a - b;
  return t * t;
}

// One staged chunk of one operand as it travels through registers: rows [r0, r0+64) x components
// [d0, d0+kChunk), 16 floats per thread.  Lane -> row, so the transposed store lds[comp][row] is
// conflict free; the loads of a wave touch 64 rows but every cache line they open is used up by
// the wave's following loads.
struct Staged {
  float4 v[kChunk / 16];
};

template <bool VEC>
__device__ __forceinline__ void load_chunk(const float* __restrict__ src, int64_t rows, int d,
                                           int64_t r0, int d0, Staged& st) {
  const int row = threadIdx.x % kTile;
  const int q0 = threadIdx.x / kTile;          // 0..3: which float4 of every group of four
  const int64_t gr = r0 + row;
  const float* p = src + (gr < rows ? gr : 0) * (int64_t)d;
#pragma unroll
  for (int e = 0; e < kChunk / 16; ++e) {
    const int gc = d0 + (e * 4 + q0) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (VEC) {
      if (gr < rows && gc < d) v = *reinterpret_cast<const float4*>(p + gc);   // d % 4 == 0
    } else if (gr < rows) {
      if (gc + 0 < d) v.x = p[gc + 0];
      if (gc + 1 < d) v.y = p[gc + 1];
      if (gc + 2 < d) v.z = p[gc + 2];
      if (gc + 3 < d) v.w = p[gc + 3];
    }
    st.v[e] = v;
  }
}

__device__ __forceinline__ void store_chunk(const Staged& st, float* lds) {
  const int row = threadIdx.x % kTile;
  const int q0 = threadIdx.x / kTile;
#pragma unroll
  for (int e = 0; e < kChunk / 16; ++e) {
    const int col = (e * 4 + q0) * 4;
    lds[(col + 0) * kLdsStride + row] = st.v[e].x;
    lds[(col + 1) * kLdsStride + row] = st.v[e].y;
    lds[(col + 2) * kLdsStride + row] = st.v[e].z;
    lds[(col + 3) * kLdsStride + row] = st.v[e].w;
  }
}

// chi-square divides by the first operand: its reciprocal is taken once per staged element, not
// once per pair (0 where OpenCV's CV_COMP_CHISQR skips the bin)
__device__ __forceinline__ float chisqr_weight(float a) {
  return fabsf(a) > 2.2204460492503131e-16f ? 1.0f / a : 0.f;
}

__device__ __forceinline__ void store_chunk_weights(const Staged& st, float* lds) {
  const int row = threadIdx.x % kTile;
  const int q0 = threadIdx.x / kTile;
#pragma unroll
  for (int e = 0; e < kChunk / 16; ++e) {
    const int col = (e * 4 + q0) * 4;
    lds[(col + 0) * kLdsStride + row] = chisqr_weight(st.v[e].x);
    lds[(col + 1) * kLdsStride + row] = chisqr_weight(st.v[e].y);
    lds[(col + 2) * kLdsStride + row] = chisqr_weight(st.v[e].z);
    lds[(col + 3) * kLdsStride + row] = chisqr_weight(st.v[e].w);
  }
}

template <int METRIC, bool VEC>
__global__ __launch_bounds__(kPairThreads) void k_pairdist(
    const float* __restrict__ x, int64_t m, const float* __restrict__ c, int64_t n, int d,
    double* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float xs[kChunk * kLdsStride];
  __shared__ __attribute__((aligned(16))) float cs[kChunk * kLdsStride];
  constexpr bool kChi = METRIC == IC_METRIC_CHISQR;
  __shared__ __attribute__((aligned(16))) float ws[kChi ? kChunk * kLdsStride : 4];
  const int64_t c0 = (int64_t)blockIdx.x * kTile;
  const int64_t x0 = (int64_t)blockIdx.y * kTile;
  const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;

  double acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;

  // the next chunk is fetched into registers while the current one is being compared
  Staged sx, sc;
  load_chunk<VEC>(x, m, d, x0, 0, sx);
  load_chunk<VEC>(c, n, d, c0, 0, sc);
  for (int d0 = 0; d0 < d; d0 += kChunk) {
    __syncthreads();
    store_chunk(sx, xs);
    store_chunk(sc, cs);
    if (kChi) store_chunk_weights(sx, ws);
    __syncthreads();
    if (d0 + kChunk < d) {
      load_chunk<VEC>(x, m, d, x0, d0 + kChunk, sx);
      load_chunk<VEC>(c, n, d, c0, d0 + kChunk, sc);
    }
    float part[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) part[i][j] = 0.f;
#pragma unroll 8
    for (int kk = 0; kk < kChunk; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&xs[kk * kLdsStride + ty * 4]);
      const float4 b = *reinterpret_cast<const float4*>(&cs[kk * kLdsStride + tx * 4]);
      const float av[4] = {a.x, a.y, a.z, a.w};
      const float bv[4] = {b.x, b.y, b.z, b.w};
      if (kChi) {
        const float4 w = *reinterpret_cast<const float4*>(&ws[kk * kLdsStride + ty * 4]);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float t = av[i] - bv[j];
            part[i][j] = __builtin_fmaf(t * t, wv[i], part[i][j]);
          }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) part[i][j] += pair_term<METRIC>(av[i], bv[j]);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] += (double)part[i][j];
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = x0 + ty * 4 + i;
    if (r >= m) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t col = c0 + tx * 4 + j;
      if (col < n) out[r * n + col] = METRIC == IC_METRIC_EUCLIDEAN ? sqrt(acc[i][j]) : acc[i][j];
    }
  }
}

// Few search rows (the usual call of the logo pipeline searches ONE histogram in the index): the
// tiled kernel above would run a handful of workgroups over the whole width.  Here every wave owns
// two index rows and streams them once, 16 bytes per lane, against up to four search rows at a
// time; the work is bounded by reading the index from HBM.
constexpr int kFewMax = 16;      // search rows up to which this kernel is used (4 per pass)
constexpr int kFewQ = 4;         // search rows per pass
constexpr int kFewR = 2;         // index rows per wave
constexpr int kFewFlush = 8;     // f32 partial sums are folded into float64 every 8 blocks

// NT: the index is far larger than the caches and read once per pass, so its loads are marked
// non-temporal (100 000 x 10 000 floats, one search row: 6.15 -> 6.51 TB/s); smaller indices
// keep the default policy, a second pass or call finds them in the Infinity Cache.
template <int METRIC, bool VEC, int Q, bool NT>
__global__ __launch_bounds__(256) void k_pairdist_few(
    const float* __restrict__ x, int m0, int mq, const float* __restrict__ c, int64_t n, int d,
    double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t r0 = wave * kFewR;
  if (r0 >= n) return;
  constexpr int W = VEC ? 4 : 1;          // floats per lane per block
  const int block = 64 * W;
  double acc[Q][kFewR];
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < kFewR; ++r) acc[q][r] = 0.0;
  const float* crow[kFewR];
#pragma unroll
  for (int r = 0; r < kFewR; ++r) crow[r] = c + (r0 + r < n ? r0 + r : n - 1) * (int64_t)d;
  const float* xrow[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) xrow[q] = x + (int64_t)(m0 + (q < mq ? q : mq - 1)) * d;

  for (int d0 = 0; d0 < d; d0 += block * kFewFlush) {
    float part[Q][kFewR];
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < kFewR; ++r) part[q][r] = 0.f;
#pragma unroll
    for (int u = 0; u < kFewFlush; ++u) {
      const int dd = d0 + u * block + lane * W;
      const bool ok = dd < d;               // with VEC, d % 4 == 0: a lane's four floats are all in or out
      const int di = ok ? dd : 0;
      float cv[kFewR][W], xv[Q][W];
#pragma unroll
      for (int r = 0; r < kFewR; ++r) {
        if (VEC) {
          typedef float f32x4_t __attribute__((ext_vector_type(4)));
          const f32x4_t* src = reinterpret_cast<const f32x4_t*>(crow[r] + di);
          const f32x4_t t = NT ? __builtin_nontemporal_load(src) : *src;
          cv[r][0] = t.x; cv[r][1 % W] = t.y; cv[r][2 % W] = t.z; cv[r][3 % W] = t.w;
        } else {
          cv[r][0] = crow[r][di];
        }
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        if (VEC) {
          const float4 t = *reinterpret_cast<const float4*>(xrow[q] + di);
          xv[q][0] = t.x; xv[q][1 % W] = t.y; xv[q][2 % W] = t.z; xv[q][3 % W] = t.w;
        } else {
          xv[q][0] = xrow[q][di];
        }
      }
#pragma unroll
      for (int q = 0; q < Q; ++q)
#pragma unroll
        for (int r = 0; r < kFewR; ++r) {
          float t = 0.f;
#pragma unroll
          for (int e = 0; e < W; ++e) t += pair_term<METRIC>(xv[q][e], cv[r][e]);
          part[q][r] += ok ? t : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int r = 0; r < kFewR; ++r) acc[q][r] += (double)part[q][r];
  }
#pragma unroll
  for (int q = 0; q < Q; ++q)
#pragma unroll
    for (int r = 0; r < kFewR; ++r) {
      double v = acc[q][r];
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
      if (lane == 0 && q < mq && r0 + r < n)
        out[(int64_t)(m0 + q) * n + r0 + r] = METRIC == IC_METRIC_EUCLIDEAN ? sqrt(v) : v;
    }
}

struct DistIdx {
  double d;
  int32_t i;
};

__device__ __forceinline__ bool pair_less(double d, int32_t i, double e, int32_t j) {
  return d < e || (d == e && i < j);
}

// The k smallest (value, index) pairs of vals[0 .. len), ascending, by one workgroup: k rounds of
// "smallest pair above the previous one".  Candidate j has index idx ? idx[j] : j0 + j.  Rounds
// that find nothing (fewer than k candidates) write +inf / 0x7fffffff.
__device__ __forceinline__ void topk_scan(const double* __restrict__ vals,
                                          const int32_t* __restrict__ idx, int64_t j0, int64_t len,
                                          int k, double* __restrict__ out_d,
                                          int32_t* __restrict__ out_i, bool final_pass) {
  __shared__ double sd[kTopkThreads / 64];
  __shared__ int32_t si[kTopkThreads / 64];
  __shared__ double last_d;
  __shared__ int32_t last_i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double prev_d = -__builtin_inf();
  int32_t prev_i = -1;
  for (int r = 0; r < k; ++r) {
    double best_d = __builtin_inf();
    int32_t best_i = 0x7fffffff;
    for (int64_t j = threadIdx.x; j < len; j += kTopkThreads) {
      const double v = vals[j];
      const int32_t ji = idx ? idx[j] : (int32_t)(j0 + j);
      const bool above = pair_less(prev_d, prev_i, v, ji);
      const bool better = pair_less(v, ji, best_d, best_i);
      if (above && better) {
        best_d = v;
        best_i = ji;
      }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const double od = __shfl_xor(best_d, off);
      const int32_t oi = __shfl_xor(best_i, off);
      if (pair_less(od, oi, best_d, best_i)) {
        best_d = od;
        best_i = oi;
      }
    }
    if (lane == 0) {
      sd[wave] = best_d;
      si[wave] = best_i;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      double bd = sd[0];
      int32_t bi = si[0];
      for (int w = 1; w < kTopkThreads / 64; ++w)
        if (pair_less(sd[w], si[w], bd, bi)) {
          bd = sd[w];
          bi = si[w];
        }
      last_d = bd;
      last_i = bi;
      out_d[r] = bd;
      out_i[r] = (final_pass && bi == 0x7fffffff) ? -1 : bi;
    }
    __syncthreads();
    prev_d = last_d;
    prev_i = last_i;
  }
}

// dist [m][n] -> the k smallest (distance, index) pairs of every row, ascending.
__global__ __launch_bounds__(kTopkThreads) void k_topk_rows(
    const double* __restrict__ dist, int64_t n, int k, double* __restrict__ out_d,
    int32_t* __restrict__ out_i) {
  topk_scan(dist + (int64_t)blockIdx.x * n, nullptr, 0, n, k, out_d + (int64_t)blockIdx.x * k,
            out_i + (int64_t)blockIdx.x * k, true);
}

// Few search rows over a long index: the row is cut into gridDim.y chunks whose local top-k are
// merged by k_topk_merge, so that one row is not left to a single workgroup.
__global__ __launch_bounds__(kTopkThreads) void k_topk_chunks(
    const double* __restrict__ dist, int64_t n, int k, double* __restrict__ part_d,
    int32_t* __restrict__ part_i) {
  const int64_t nch = gridDim.y;
  const int64_t c0 = n * blockIdx.y / nch, c1 = n * (blockIdx.y + 1) / nch;
  const int64_t o = ((int64_t)blockIdx.x * nch + blockIdx.y) * k;
  topk_scan(dist + (int64_t)blockIdx.x * n + c0, nullptr, c0, c1 - c0, k, part_d + o, part_i + o,
            false);
}

__global__ __launch_bounds__(kTopkThreads) void k_topk_merge(
    const double* __restrict__ part_d, const int32_t* __restrict__ part_i, int nch, int k,
    double* __restrict__ out_d, int32_t* __restrict__ out_i) {
  const int64_t o = (int64_t)blockIdx.x * nch * k;
  topk_scan(part_d + o, part_i + o, 0, (int64_t)nch * k, k, out_d + (int64_t)blockIdx.x * k,
            out_i + (int64_t)blockIdx.x * k, true);
}

}  // namespace

template <int METRIC, int Q>
static void launch_few_q(const float* x, int m0, int mq, const float* c, int64_t n, int d,
                         double* out, hipStream_t s) {
  const dim3 grid((unsigned)((n + 4 * kFewR - 1) / (4 * kFewR)));
  // 16-byte loads need rows that start on 16-byte boundaries (a caller may pass a sub-view)
  const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(c) & 15) == 0;
  // (the Infinity Cache holds 256 MB)
  const bool stream = vec && n * (int64_t)d * 4 > (512ll << 20);
  if (stream)
    hipLaunchKernelGGL((k_pairdist_few<METRIC, true, Q, true>), grid, dim3(256), 0, s, x, m0, mq, c, n, d, out);
  else if (vec)
    hipLaunchKernelGGL((k_pairdist_few<METRIC, true, Q, false>), grid, dim3(256), 0, s, x, m0, mq, c, n, d, out);
  else
    hipLaunchKernelGGL((k_pairdist_few<METRIC, false, Q, false>), grid, dim3(256), 0, s, x, m0, mq, c, n, d, out);
}

// up to kFewQ search rows per pass over the index; a single search row gets its own instantiation
// so that it pays for one row's arithmetic only and stays bound by reading the index
template <int METRIC>
static void launch_few(const float* x, int64_t m, const float* c, int64_t n, int d, double* out,
                       hipStream_t s) {
  for (int m0 = 0; m0 < (int)m; m0 += kFewQ) {
    const int mq = (int)m - m0 < kFewQ ? (int)m - m0 : kFewQ;
    if (mq == 1) launch_few_q<METRIC, 1>(x, m0, mq, c, n, d, out, s);
    else if (mq == 2) launch_few_q<METRIC, 2>(x, m0, mq, c, n, d, out, s);
    else launch_few_q<METRIC, kFewQ>(x, m0, mq, c, n, d, out, s);
  }
}

template <int METRIC>
void launch_tiled(const float* x, int64_t m, const float* c, int64_t n, int d, double* out,
                  hipStream_t s) {
  const dim3 grid((unsigned)((n + kTile - 1) / kTile), (unsigned)((m + kTile - 1) / kTile));
  const dim3 block(kPairThreads);
  // 16-byte loads need rows that start on 16-byte boundaries
  const bool vec = d % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(c) & 15) == 0;
  if (vec) hipLaunchKernelGGL((k_pairdist<METRIC, true>), grid, block, 0, s, x, m, c, n, d, out);
  else hipLaunchKernelGGL((k_pairdist<METRIC, false>), grid, block, 0, s, x, m, c, n, d, out);
}

hipError_t launch_pairdist(const float* x, int64_t m, const float* c, int64_t n, int d, int metric,
                           double* out, hipStream_t s) {
  if (m <= kFewMax) {
    switch (metric) {
      case IC_METRIC_EUCLIDEAN: launch_few<IC_METRIC_EUCLIDEAN>(x, m, c, n, d, out, s); break;
      case IC_METRIC_SQEUCLIDEAN: launch_few<IC_METRIC_SQEUCLIDEAN>(x, m, c, n, d, out, s); break;
      case IC_METRIC_L1: launch_few<IC_METRIC_L1>(x, m, c, n, d, out, s); break;
      case IC_METRIC_INTERSECTION: launch_few<IC_METRIC_INTERSECTION>(x, m, c, n, d, out, s); break;
      case IC_METRIC_CHISQR: launch_few<IC_METRIC_CHISQR>(x, m, c, n, d, out, s); break;
      default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
  }
  switch (metric) {
    case IC_METRIC_EUCLIDEAN: launch_tiled<IC_METRIC_EUCLIDEAN>(x, m, c, n, d, out, s); break;
    case IC_METRIC_SQEUCLIDEAN: launch_tiled<IC_METRIC_SQEUCLIDEAN>(x, m, c, n, d, out, s); break;
    case IC_METRIC_L1: launch_tiled<IC_METRIC_L1>(x, m, c, n, d, out, s); break;
    case IC_METRIC_INTERSECTION: launch_tiled<IC_METRIC_INTERSECTION>(x, m, c, n, d, out, s); break;
    case IC_METRIC_CHISQR: launch_tiled<IC_METRIC_CHISQR>(x, m, c, n, d, out, s); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

int topk_chunks(int64_t m, int64_t n, int k) {
  // chunking pays when the rows alone cannot fill the machine and a row is long
  if (m >= 128 || n < 16384) return 1;
  int64_t nch = 512 / m;
  if (nch > n / 4096) nch = n / 4096;
  if (nch > n / (2 * (int64_t)k)) nch = n / (2 * (int64_t)k);
  return nch < 2 ? 1 : (int)nch;
}

hipError_t launch_topk_rows(const double* dist, int64_t m, int64_t n, int k, int nch,
                            double* part_d, int32_t* part_i, double* out_d, int32_t* out_i,
                            hipStream_t s) {
  if (nch <= 1) {
    hipLaunchKernelGGL(k_topk_rows, dim3((unsigned)m), dim3(kTopkThreads), 0, s, dist, n, k, out_d,
                       out_i);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(k_topk_chunks, dim3((unsigned)m, (unsigned)nch), dim3(kTopkThreads), 0, s,
                     dist, n, k, part_d, part_i);
  hipLaunchKernelGGL(k_topk_merge, dim3((unsigned)m), dim3(kTopkThreads), 0, s, part_d, part_i, nch,
                     k, out_d, out_i);
  return hipGetLastError();
}

}  // namespace ic
