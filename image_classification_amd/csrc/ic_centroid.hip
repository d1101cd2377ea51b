// k-means centroid sums at HBM speed: rows are first partitioned by cluster bucket (a stable
// counting sort built from in-LDS bitonic sorts, no atomics), then every bucket's rows are gathered
// once and summed in float64 in ascending row order.
//
// Replaces scipy's _vq.update_cluster_means inside kmeans2 (vision/facerec/face_indexing.py:158 of
// the reference).  Algorithmic traffic: n*d*4 bytes of rows read once + k*d*8 bytes written, plus
// 20 bytes per row of keys / permutation.
//
//   P1 k_cent_sort_count  block of 4096 rows: key = bucket << 12 | local row, bitonic sort in LDS,
//                         sorted keys -> workspace, rows per (block, bucket) -> table
//   P2 k_cent_colscan     per bucket: exclusive prefix over blocks (in place) and bucket total
//      k_cent_bucketscan  exclusive prefix over bucket totals
//   P3 k_cent_scatter     (row, code) of every row -> its slot in bucket order (stable)
//   P4 k_cent_sums        workgroup = (bucket of 32 clusters, range of the bucket's 2048-row
//                         chunks): the chunk's rows grouped by cluster with wave ballots (row
//                         order kept), each wave sums the rows of its clusters, 16 row loads in
//                         flight; float64 sums in LDS
//   P5 k_centroid_reduce  (ic_reduce.hip) folds the ranges in ascending order
// Every word has one writer between two barriers, and the order of every summation is fixed by
// the row order, so results are bit-reproducible.
#include "ic_internal.h"

namespace ic {

constexpr int kBlkRows = 4096;       // rows per sort block (12 bits of local position)
constexpr int kBucketShift = 5;      // 32 clusters per bucket
constexpr int kBucketClusters = 1 << kBucketShift;
constexpr unsigned kNoKey = 0xFFFFFFFFu;

// ascending bitonic sort of buf[0..P), P a power of two >= 2 * THREADS
template <int THREADS>
__device__ __forceinline__ void bitonic_sort_lds(unsigned* buf, int P) {
  for (int size = 2; size <= P; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < P / 2; t += THREADS) {
        const int pos = 2 * t - (t & (stride - 1));
        const unsigned a = buf[pos], b = buf[pos + stride];
        const bool up = (pos & size) == 0;
        if ((a > b) == up) {
          buf[pos] = b;
          buf[pos + stride] = a;
        }
      }
    }
  }
  __syncthreads();
}

// first position in sorted buf[0..n) whose key is >= key
__device__ __forceinline__ int lower_bound_lds(const unsigned* buf, int n, unsigned key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (buf[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__device__ __forceinline__ int pow2_at_least(int v, int floor_) {
  int p = floor_;
  while (p < v) p <<= 1;
  return p;
}

// (a step of the sort is bound by its barrier, not by its compare-exchanges: 1024 threads take two
// each instead of eight)
constexpr int kSortThreads = 1024;

__global__ __launch_bounds__(kSortThreads) void k_cent_sort_count(const int32_t* __restrict__ codes,
                                                         int64_t n, int64_t k, int64_t nb,
                                                         unsigned* __restrict__ keys,
                                                         unsigned* __restrict__ tab) {
  __shared__ unsigned buf[kBlkRows];
  const int64_t base = (int64_t)blockIdx.x * kBlkRows;
  const int64_t left = n - base;
  const int cnt = left < kBlkRows ? (int)left : kBlkRows;
  const int P = pow2_at_least(cnt, 2 * kSortThreads);
  for (int p = threadIdx.x; p < P; p += kSortThreads) {
    unsigned key = kNoKey;
    if (p < cnt) {
      const int32_t c = codes[base + p];
      if (c >= 0 && c < k) key = ((unsigned)(c >> kBucketShift) << 12) | (unsigned)p;
    }
    buf[p] = key;
  }
  bitonic_sort_lds<kSortThreads>(buf, P);
  for (int p = threadIdx.x; p < cnt; p += kSortThreads) {
    const unsigned key = buf[p];
    keys[base + p] = key;
    if (key == kNoKey) continue;
    const unsigned b = key >> 12;
    const bool last = (p == cnt - 1) || (buf[p + 1] >> 12) != b;  // kNoKey >> 12 differs too
    if (last) tab[(int64_t)blockIdx.x * nb + b] = (unsigned)(p + 1 - lower_bound_lds(buf, cnt, b << 12));
  }
}

// Per bucket: exclusive prefix over the blocks (in place) and the bucket's total.  A workgroup
// takes 32 buckets; its eight groups of 32 threads each take an eighth of the blocks, sum it,
// and after the exchange of the eight sums write the running prefix of their part.
__global__ __launch_bounds__(256) void k_cent_colscan(unsigned* __restrict__ tab, int64_t nblk,
                                                      int64_t nb, int64_t* __restrict__ total) {
  __shared__ long long part[8][32];
  const int col = threadIdx.x & 31, g = threadIdx.x >> 5;
  const int64_t b = (int64_t)blockIdx.x * 32 + col;
  const int64_t blk0 = nblk * g / 8, blk1 = nblk * (g + 1) / 8;
  long long sum = 0;
  if (b < nb)
    for (int64_t blk = blk0; blk < blk1; ++blk) sum += tab[blk * nb + b];
  part[g][col] = sum;
  __syncthreads();
  if (b >= nb) return;
  long long run = 0;
  for (int gg = 0; gg < g; ++gg) run += part[gg][col];
  for (int64_t blk = blk0; blk < blk1; ++blk) {
    const unsigned c = tab[blk * nb + b];
    tab[blk * nb + b] = (unsigned)run;  // rows of this bucket in earlier blocks (< 2^31 rows)
    run += c;
  }
  if (g == 7) total[b] = run;
}

// start[b] = sum of total[0..b), start[nb] = number of valid rows; one workgroup, in place capable
__global__ __launch_bounds__(1024) void k_cent_bucketscan(const int64_t* __restrict__ total,
                                                          int64_t nb, int64_t* __restrict__ start) {
  __shared__ long long part[1024];
  __shared__ long long carry_s;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < nb; base += 1024) {
    const int64_t i = base + threadIdx.x;
    const long long v = i < nb ? total[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
      const long long add = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += add;
      __syncthreads();
    }
    const long long carry = carry_s;
    if (i < nb) start[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry_s = carry + part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) start[nb] = carry_s;
}

struct RowCode {
  int32_t row;
  int32_t code;
};

__global__ __launch_bounds__(256) void k_cent_scatter(const int32_t* __restrict__ codes, int64_t n,
                                                      int64_t nb, const unsigned* __restrict__ keys,
                                                      const unsigned* __restrict__ tab,
                                                      const int64_t* __restrict__ start,
                                                      RowCode* __restrict__ perm) {
  __shared__ unsigned buf[kBlkRows];
  const int64_t base = (int64_t)blockIdx.x * kBlkRows;
  const int64_t left = n - base;
  const int cnt = left < kBlkRows ? (int)left : kBlkRows;
  for (int p = threadIdx.x; p < cnt; p += 256) buf[p] = keys[base + p];
  __syncthreads();
  for (int p = threadIdx.x; p < cnt; p += 256) {
    const unsigned key = buf[p];
    if (key == kNoKey) continue;
    const unsigned b = key >> 12;
    const int rank = p - lower_bound_lds(buf, cnt, b << 12);
    const int64_t row = base + (key & 4095u);
    RowCode rc;
    rc.row = (int32_t)row;
    rc.code = codes[row];
    perm[start[b] + tab[(int64_t)blockIdx.x * nb + b] + rank] = rc;
  }
}

// rows of a bucket a workgroup of k_cent_sums stages at a time: with 2048 the kernel's LDS
// (32 KB of sums + 14 KB of rows, positions and cluster numbers) lets three workgroups share a CU
constexpr int kSumChunk = 2048;
constexpr int kSumThreads = 512;
constexpr int kSumWaves = kSumThreads / 64;
constexpr int kRowsInFlight = 16;

// Workgroup = (bucket of 32 clusters, range of the bucket's chunks, slab of 128 columns).  A
// chunk's (row, cluster) pairs arrive in partition order, i.e. ascending rows inside the bucket.
// They are grouped by cluster without a sort: wave w owns the clusters lc with lc % 8 == w and, for
// each of them, walks the chunk's cluster numbers 64 at a time -- a ballot marks its cluster's
// positions, the set bits below a lane give the lane's place -- first to count, then (after an
// exclusive scan over the 32 counts) to write the positions of the cluster's rows, in row order,
// into its stretch of `list`.  The same wave then sums that stretch: 16 row loads in flight,
// float64 accumulators, ascending rows.  (A bitonic sort of the chunk by (cluster, position) did
// the grouping before: 55 barrier-bound steps for a 610-row chunk, 0.249 against 0.228 ms at
// 1.25 M rows x 65 536 clusters.  Walking a wave's four clusters as ONE run of 16-row batches,
// with the accumulators handed over wherever the cluster changes, saves three of eight batches
// and measured 0.257 ms: the hand-over inside the unrolled adds costs more than the batches.)
template <bool PAIR>  // PAIR: d is even, a lane loads its two components as one 8-byte word
__global__ __launch_bounds__(kSumThreads) void k_cent_sums(
    const float* __restrict__ X, int d, int64_t k, const RowCode* __restrict__ perm,
    const int64_t* __restrict__ start, int n_ranges, double* __restrict__ psums,
    int64_t* __restrict__ pcounts) {
  __shared__ __attribute__((aligned(16))) double sums[kBucketClusters][kD];
  __shared__ long long cnt[kBucketClusters];
  __shared__ int rowbuf[kSumChunk];
  __shared__ unsigned short list[kSumChunk];
  __shared__ unsigned char code5[kSumChunk];
  __shared__ int ccount[kBucketClusters], cstart[kBucketClusters];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t b = blockIdx.x;
  const int64_t r = blockIdx.y;
  const int64_t c0 = b << kBucketShift;
  // rows wider than 128 columns: workgroup z sums columns [128 z, 128 z + 128) (the grouping is
  // done again by each, which costs little next to the row gathers)
  const int col0 = (int)blockIdx.z * kD;
  const int dw = d - col0 < kD ? d - col0 : kD;   // columns of this slab

  for (int t = threadIdx.x; t < kBucketClusters * kD; t += kSumThreads) (&sums[0][0])[t] = 0.0;
  if (threadIdx.x < kBucketClusters) cnt[threadIdx.x] = 0;

  const int64_t s0 = start[b], s1 = start[b + 1];
  const int64_t chunks = (s1 - s0 + kSumChunk - 1) / kSumChunk;
  const int64_t ch0 = chunks * r / n_ranges, ch1 = chunks * (r + 1) / n_ranges;
  const int c2 = 2 * lane;
  const unsigned long long below = (1ull << lane) - 1ull;   // the lanes before this one
  constexpr int kMine = kBucketClusters / kSumWaves;         // clusters per wave

  for (int64_t ch = ch0; ch < ch1; ++ch) {
    const int64_t base = s0 + ch * kSumChunk;
    const int64_t left = s1 - base;
    const int n_here = left < kSumChunk ? (int)left : kSumChunk;
    __syncthreads();  // previous chunk fully consumed (and the zeroing above finished)
    for (int p = threadIdx.x; p < n_here; p += kSumThreads) {
      const RowCode rc = perm[base + p];
      rowbuf[p] = rc.row;
      code5[p] = (unsigned char)(rc.code - (int32_t)c0);
    }
    __syncthreads();
    // rows per cluster of this wave
    int n_of[kMine];
#pragma unroll
    for (int q = 0; q < kMine; ++q) n_of[q] = 0;
    for (int p0 = 0; p0 < n_here; p0 += 64) {
      const int mine = p0 + lane < n_here ? (int)code5[p0 + lane] : -1;
#pragma unroll
      for (int q = 0; q < kMine; ++q)
        n_of[q] += __builtin_popcountll(__builtin_amdgcn_ballot_w64(mine == wave + q * kSumWaves));
    }
    if (lane == 0) {
#pragma unroll
      for (int q = 0; q < kMine; ++q) ccount[wave + q * kSumWaves] = n_of[q];
    }
    __syncthreads();
    if (threadIdx.x < kBucketClusters) {
      int at = 0;
      for (int u = 0; u < (int)threadIdx.x; ++u) at += ccount[u];
      cstart[threadIdx.x] = at;
    }
    __syncthreads();
    // positions of each cluster's rows, in row order
    int fill[kMine];
#pragma unroll
    for (int q = 0; q < kMine; ++q) fill[q] = cstart[wave + q * kSumWaves];
    for (int p0 = 0; p0 < n_here; p0 += 64) {
      const int mine = p0 + lane < n_here ? (int)code5[p0 + lane] : -1;
#pragma unroll
      for (int q = 0; q < kMine; ++q) {
        const bool hit = mine == wave + q * kSumWaves;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        if (hit) list[fill[q] + __builtin_popcountll(m & below)] = (unsigned short)(p0 + lane);
        fill[q] += __builtin_popcountll(m);
      }
    }
    // (the stretches of this wave's clusters were written by this wave: no barrier, the
    // hardware keeps a wave's LDS accesses in order)
#pragma unroll 1
    for (int q = 0; q < kMine; ++q) {
      const int lc = wave + q * kSumWaves;
      const int lo = cstart[lc], hi = lo + ccount[lc];
      if (lo == hi) continue;
      double a0 = 0.0, a1 = 0.0;
      // Every load below is unconditional (the tail re-reads the run's last row, lanes past the
      // row's end re-read component 0) and the unwanted values are dropped by selects afterwards:
      // a branch around a load makes the compiler wait for each load before issuing the next.
      const int ca = col0 + (c2 < dw ? c2 : 0);
      const int cb = col0 + (c2 + 1 < dw ? c2 + 1 : 0);
      for (int p = lo; p < hi; p += kRowsInFlight) {
        int rows[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          const int pp = p + u < hi ? p + u : hi - 1;
          rows[u] = rowbuf[list[pp]];
        }
        float v0[kRowsInFlight], v1[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          const float* xrow = X + (int64_t)rows[u] * d;
          if (PAIR) {
            const float2 t = *reinterpret_cast<const float2*>(xrow + ca);
            v0[u] = t.x;
            v1[u] = t.y;
          } else {
            v0[u] = xrow[ca];
            v1[u] = xrow[cb];
          }
        }
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {  // fixed order: ascending rows
          const bool live = p + u < hi;
          a0 += (live && c2 < dw) ? (double)v0[u] : 0.0;
          a1 += (live && c2 + 1 < dw) ? (double)v1[u] : 0.0;
        }
      }
      sums[lc][c2] += a0;
      sums[lc][c2 + 1] += a1;
      if (lane == 0) cnt[lc] += hi - lo;
    }
  }
  __syncthreads();

  int64_t ncl = k - c0;
  if (ncl > kBucketClusters) ncl = kBucketClusters;
  double* ps = psums + (r * k + c0) * d + col0;
  for (int t = threadIdx.x; t < (int)ncl * dw; t += kSumThreads)
    ps[(int64_t)(t / dw) * d + t % dw] = sums[t / dw][t % dw];
  if (blockIdx.z == 0 && threadIdx.x < ncl) pcounts[r * k + c0 + threadIdx.x] = cnt[threadIdx.x];
}

void centroid_plan(int64_t n, int64_t k, int d, int num_cus, CentroidPlan* p) {
  p->nblk = (n + kBlkRows - 1) / kBlkRows;
  if (p->nblk < 1) p->nblk = 1;
  p->nb = (k + kBucketClusters - 1) / kBucketClusters;
  int64_t r = (4LL * num_cus + p->nb - 1) / p->nb;
  const int64_t avg_chunks = (n / p->nb + kSumChunk - 1) / kSumChunk;
  if (r > avg_chunks) r = avg_chunks;
  if (r > 65535) r = 65535;
  if (r < 1) r = 1;
  p->n_ranges = (int)r;
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  p->o_keys = off;   off += up((size_t)n * 4 + 16);
  p->o_tab = off;    p->tab_bytes = (size_t)p->nblk * (size_t)p->nb * 4; off += up(p->tab_bytes);
  p->o_total = off;  off += up((size_t)(p->nb + 1) * 8);
  p->o_start = off;  off += up((size_t)(p->nb + 1) * 8);
  p->o_perm = off;   off += up((size_t)n * 8 + 16);
  p->o_psums = off;  if (r > 1) off += up((size_t)r * (size_t)k * d * 8);
  p->o_pcounts = off; if (r > 1) off += up((size_t)r * (size_t)k * 8);
  p->bytes = off;
  // the (block, bucket) table grows with n * k; beyond 2 GiB the scan-based kernel is used instead
  // (a bucket number must also stay below 2^20 so that it cannot look like the padding key)
  p->use_scan = p->tab_bytes > ((size_t)2 << 30) || n > 0x7fffffffLL || p->nb >= (1 << 20);
}

hipError_t launch_centroid_partials(const float* x, const int32_t* codes, int64_t n, int64_t k,
                                    int d, const CentroidPlan& p, char* ws, double* sums,
                                    int64_t* counts, hipStream_t s) {
  unsigned* keys = reinterpret_cast<unsigned*>(ws + p.o_keys);
  unsigned* tab = reinterpret_cast<unsigned*>(ws + p.o_tab);
  int64_t* total = reinterpret_cast<int64_t*>(ws + p.o_total);
  int64_t* start = reinterpret_cast<int64_t*>(ws + p.o_start);
  RowCode* perm = reinterpret_cast<RowCode*>(ws + p.o_perm);
  double* psums = p.n_ranges > 1 ? reinterpret_cast<double*>(ws + p.o_psums) : sums;
  int64_t* pcounts = p.n_ranges > 1 ? reinterpret_cast<int64_t*>(ws + p.o_pcounts) : counts;
  if (p.nblk > 0x7fffffffLL || p.nb > 0x7fffffffLL) return hipErrorInvalidValue;

  hipError_t e = hipMemsetAsync(tab, 0, p.tab_bytes, s);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_cent_sort_count, dim3((unsigned)p.nblk), dim3(kSortThreads), 0, s, codes, n, k, p.nb,
                     keys, tab);
  hipLaunchKernelGGL(k_cent_colscan, dim3((unsigned)((p.nb + 31) / 32)), dim3(256), 0, s, tab,
                     p.nblk, p.nb, total);
  hipLaunchKernelGGL(k_cent_bucketscan, dim3(1), dim3(1024), 0, s, total, p.nb, start);
  hipLaunchKernelGGL(k_cent_scatter, dim3((unsigned)p.nblk), dim3(256), 0, s, codes, n, p.nb, keys,
                     tab, start, perm);
  const unsigned col_slabs = (unsigned)((d + kD - 1) / kD);
  if (d % 2 == 0)
    hipLaunchKernelGGL(k_cent_sums<true>, dim3((unsigned)p.nb, (unsigned)p.n_ranges, col_slabs),
                       dim3(kSumThreads), 0, s, x, d, k, perm, start, p.n_ranges, psums, pcounts);
  else
    hipLaunchKernelGGL(k_cent_sums<false>, dim3((unsigned)p.nb, (unsigned)p.n_ranges, col_slabs),
                       dim3(kSumThreads), 0, s, x, d, k, perm, start, p.n_ranges, psums, pcounts);
  e = hipGetLastError();
  if (e != hipSuccess || p.n_ranges == 1) return e;
  return launch_centroid_reduce(psums, pcounts, k, d, p.n_ranges, sums, counts, s);
}

}  // namespace ic
