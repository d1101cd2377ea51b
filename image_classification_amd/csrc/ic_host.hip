// Host-side row gather of the upload pipeline: the descriptor arrays of many images are copied
// (float64 ones converted) into one page-locked buffer by a few plain threads.  Nothing here
// touches the GPU.  The reference does this step with numpy.concatenate before every search
// (model/features.py:1164); done from Python threads the copies spend more time waiting for the
// interpreter lock than copying.
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "ic_internal.h"

namespace {

struct GatherJob {
  const void* src;
  int64_t elems;   // floats to produce
  int32_t kind;    // 0: float32 source, 1: float64 source
  int64_t dst0;    // first destination element
};

// elements [lo, hi) of the concatenation of all jobs
void gather_range(float* dst, const GatherJob* jobs, int64_t n_jobs, int64_t lo, int64_t hi) {
  // first job that ends beyond lo
  int64_t a = 0, b = n_jobs;
  while (a < b) {
    const int64_t m = (a + b) / 2;
    if (jobs[m].dst0 + jobs[m].elems <= lo) a = m + 1; else b = m;
  }
  for (int64_t j = a; j < n_jobs && jobs[j].dst0 < hi; ++j) {
    const GatherJob& g = jobs[j];
    const int64_t s = std::max<int64_t>(lo, g.dst0) - g.dst0;
    const int64_t e = std::min<int64_t>(hi, g.dst0 + g.elems) - g.dst0;
    if (e <= s) continue;
    if (g.kind == 0) {
      std::memcpy(dst + g.dst0 + s, (const float*)g.src + s, (size_t)(e - s) * sizeof(float));
    } else {
      const double* p = (const double*)g.src;
      float* q = dst + g.dst0;
      for (int64_t i = s; i < e; ++i) q[i] = (float)p[i];
    }
  }
}

}  // namespace

extern "C" int ic_host_gather_f32(float* dst, const void* const* src, const int64_t* elems,
                                  const int32_t* kind, int64_t n_jobs, int threads) {
  if (n_jobs < 0 || threads < 1 || threads > 256) return IC_EINVAL;
  if (n_jobs == 0) return IC_OK;
  if (!dst || !src || !elems) return IC_EINVAL;
  std::vector<GatherJob> jobs((size_t)n_jobs);
  int64_t total = 0;
  for (int64_t j = 0; j < n_jobs; ++j) {
    const int32_t kd = kind ? kind[j] : 0;
    if (elems[j] < 0 || (kd != 0 && kd != 1) || (elems[j] > 0 && !src[j])) return IC_EINVAL;
    jobs[(size_t)j] = GatherJob{src[j], elems[j], kd, total};
    total += elems[j];
  }
  if (total == 0) return IC_OK;
  // equal shares of the output, cut at 64-byte boundaries; small gathers use fewer threads
  const int64_t min_share = 1 << 16;
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(threads, total / min_share));
  const int64_t share = ((total + nt - 1) / nt + 15) / 16 * 16;
  std::vector<std::thread> pool;
  pool.reserve((size_t)nt);
  for (int t = 1; t < nt; ++t) {
    const int64_t lo = std::min<int64_t>(total, t * share), hi = std::min<int64_t>(total, lo + share);
    if (hi <= lo) continue;
    try {
      pool.emplace_back(gather_range, dst, jobs.data(), n_jobs, lo, hi);
    } catch (...) {
      gather_range(dst, jobs.data(), n_jobs, lo, hi);   // no thread to be had: copy it here
    }
  }
  gather_range(dst, jobs.data(), n_jobs, 0, std::min<int64_t>(total, share));
  for (auto& th : pool) th.join();
  return IC_OK;
}
