// Sanitizer build of the library's host-only code: ic_host_gather_f32 under AddressSanitizer and
// UndefinedBehaviorSanitizer, with buffers sized exactly so that any overrun is caught.
// Built and run by tests/test_host_sanitizers.py (g++/hipcc host code only, no GPU).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" int ic_host_gather_f32(float* dst, const void* const* src, const int64_t* elems,
                                  const int32_t* kind, int64_t n_jobs, int threads);

int main() {
  unsigned seed = 12345u;
  auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
  for (int round = 0; round < 40; ++round) {
    const int n_jobs = 1 + (int)(rnd() % 60);
    std::vector<std::vector<float>> f32;
    std::vector<std::vector<double>> f64;
    std::vector<const void*> src;
    std::vector<int64_t> elems;
    std::vector<int32_t> kind;
    int64_t total = 0;
    for (int j = 0; j < n_jobs; ++j) {
      const int64_t e = (rnd() % 5 == 0) ? 0 : (int64_t)(rnd() % 70000);
      const bool wide = rnd() % 3 == 0;
      if (wide) {
        f64.emplace_back((size_t)e);
        for (auto& v : f64.back()) v = (double)(rnd() % 1000) / 7.0;
        src.push_back(f64.back().data());
      } else {
        f32.emplace_back((size_t)e);
        for (auto& v : f32.back()) v = (float)(rnd() % 1000) / 7.0f;
        src.push_back(f32.back().data());
      }
      elems.push_back(e);
      kind.push_back(wide ? 1 : 0);
      total += e;
    }
    // malloc'ed to the exact size: the sanitizer flags the first byte written or read past it
    float* dst = (float*)std::malloc((size_t)(total > 0 ? total : 1) * sizeof(float));
    const int threads = 1 + (int)(rnd() % 16);
    if (ic_host_gather_f32(dst, src.data(), elems.data(), kind.data(), n_jobs, threads) != 0) {
      std::printf("round %d: call failed\n", round);
      return 1;
    }
    int64_t pos = 0;
    size_t a = 0, b = 0;
    for (int j = 0; j < n_jobs; ++j) {
      for (int64_t i = 0; i < elems[j]; ++i) {
        const float want = kind[j] ? (float)f64[b][(size_t)i] : f32[a][(size_t)i];
        if (dst[pos + i] != want) {
          std::printf("round %d job %d element %lld differs\n", round, j, (long long)i);
          return 1;
        }
      }
      pos += elems[j];
      if (kind[j]) ++b; else ++a;
    }
    std::free(dst);
  }
  std::printf("host gather: 40 rounds clean\n");
  return 0;
}
