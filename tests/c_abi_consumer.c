/* A consumer of libic_hip.so written against include/ic_hip.h alone: no Python, no torch.
 *
 * It owns its device buffers through the HIP runtime, searches 3000 synthetic descriptors in a
 * 500-word codebook (ic_assign_top2), builds per-image histograms (ic_hist) and centroid sums
 * (ic_centroid_partials), and checks all three against a float64 brute-force loop written here.
 * Built by __graft_entry__.build(), run by tests/test_gpu_c_abi.py; exit status 0 = all checks hold.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ic_hip.h"

#define N 3000
#define K 500
#define D 128
#define N_IMG 7

#define HIPCHECK(call)                                                              \
  do {                                                                              \
    hipError_t e_ = (call);                                                         \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));             \
      return 2;                                                                     \
    }                                                                               \
  } while (0)

#define ICCHECK(call)                                                               \
  do {                                                                              \
    int rc_ = (call);                                                               \
    if (rc_ != IC_OK) {                                                             \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ic_last_error(NULL));           \
      return 3;                                                                     \
    }                                                                               \
  } while (0)

static uint64_t lcg_state = 12345;
static double lcg(void) { /* uniform in [0, 1) */
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return (double)(lcg_state >> 11) / 9007199254740992.0;
}

int main(void) {
  static float x[N][D], c[K][D];
  for (int j = 0; j < K; ++j)
    for (int t = 0; t < D; ++t) c[j][t] = (float)floor(lcg() * 120.0);
  for (int i = 0; i < N; ++i) { /* a descriptor = some codeword + noise */
    const int j = (int)(lcg() * K);
    for (int t = 0; t < D; ++t) x[i][t] = c[j][t] + (float)((lcg() - 0.5) * 60.0);
  }
  int64_t offsets[N_IMG + 1] = {0, 400, 400, 1000, 1001, 2200, 2999, N}; /* one image is empty */

  void* h = NULL;
  ICCHECK(ic_create(0, &h));
  float *dx, *dc, *dd1, *dd2;
  int32_t *di1, *di2;
  uint8_t* dmask;
  int64_t *doff, *dcounts;
  double *dhist, *dsums;
  HIPCHECK(hipMalloc((void**)&dx, sizeof x));
  HIPCHECK(hipMalloc((void**)&dc, sizeof c));
  HIPCHECK(hipMalloc((void**)&di1, N * 4));
  HIPCHECK(hipMalloc((void**)&di2, N * 4));
  HIPCHECK(hipMalloc((void**)&dd1, N * 4));
  HIPCHECK(hipMalloc((void**)&dd2, N * 4));
  HIPCHECK(hipMalloc((void**)&dmask, N));
  HIPCHECK(hipMalloc((void**)&doff, sizeof offsets));
  HIPCHECK(hipMalloc((void**)&dhist, (size_t)N_IMG * K * 8));
  HIPCHECK(hipMalloc((void**)&dsums, (size_t)K * D * 8));
  HIPCHECK(hipMalloc((void**)&dcounts, (size_t)K * 8));
  HIPCHECK(hipMemcpy(dx, x, sizeof x, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dc, c, sizeof c, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(doff, offsets, sizeof offsets, hipMemcpyHostToDevice));

  hipStream_t s;
  HIPCHECK(hipStreamCreate(&s));
  ICCHECK(ic_assign_top2(h, dx, N, dc, K, D, NULL, di1, dd1, di2, dd2, 0.8f, dmask, s));
  ICCHECK(ic_hist(h, di1, doff, N_IMG, K, IC_NORM_L1, 0.5, dhist, s));
  ICCHECK(ic_centroid_partials(h, dx, di1, N, K, D, dsums, dcounts, s));
  HIPCHECK(hipStreamSynchronize(s));

  static int32_t i1[N], i2[N];
  static float d1[N], d2[N];
  static uint8_t mask[N];
  static double hist[N_IMG][K], sums[K][D];
  static int64_t counts[K];
  HIPCHECK(hipMemcpy(i1, di1, sizeof i1, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(i2, di2, sizeof i2, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(d1, dd1, sizeof d1, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(d2, dd2, sizeof d2, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(mask, dmask, sizeof mask, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(hist, dhist, sizeof hist, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(sums, dsums, sizeof sums, hipMemcpyDeviceToHost));
  HIPCHECK(hipMemcpy(counts, dcounts, sizeof counts, hipMemcpyDeviceToHost));

  /* float64 brute force: two smallest (distance, index) per row, lowest index first on ties */
  int bad = 0;
  for (int i = 0; i < N; ++i) {
    double b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
    int j1 = -1, j2 = -1;
    for (int j = 0; j < K; ++j) {
      double sq = 0;
      for (int t = 0; t < D; ++t) {
        const double df = (double)x[i][t] - (double)c[j][t];
        sq += df * df;
      }
      if (sq < b1) { b3 = b2; b2 = b1; j2 = j1; b1 = sq; j1 = j; }
      else if (sq < b2) { b3 = b2; b2 = sq; j2 = j; }
      else if (sq < b3) { b3 = sq; }
    }
    const double e1 = sqrt(b1), e2 = sqrt(b2), e3 = sqrt(b3);
    /* indices must agree wherever the next candidate is more than 1e-5 (relative) away */
    if (e2 - e1 > 1e-5 * e1 && i1[i] != j1) { ++bad; fprintf(stderr, "row %d: i1 %d != %d\n", i, i1[i], j1); }
    if (e2 - e1 > 1e-5 * e1 && e3 - e2 > 1e-5 * e2 && i2[i] != j2) { ++bad; fprintf(stderr, "row %d: i2 %d != %d\n", i, i2[i], j2); }
    if (i1[i] == j1 && fabs(d1[i] - e1) > 1e-4 * e1 + 1e-6) { ++bad; fprintf(stderr, "row %d: d1 %g != %g\n", i, d1[i], e1); }
    if (i2[i] == j2 && fabs(d2[i] - e2) > 1e-4 * e2 + 1e-6) { ++bad; fprintf(stderr, "row %d: d2 %g != %g\n", i, d2[i], e2); }
    if (fabs(e1 - 0.8 * e2) > 1e-4 * e2 && mask[i] != (e1 < 0.8 * e2)) { ++bad; fprintf(stderr, "row %d: mask\n", i); }
  }
  /* histograms and centroid sums from the codes the device returned: exact rules */
  for (int g = 0; g < N_IMG; ++g) {
    static double ref[K];
    memset(ref, 0, sizeof ref);
    const int64_t cnt = offsets[g + 1] - offsets[g];
    for (int64_t i = offsets[g]; i < offsets[g + 1]; ++i) ref[i1[i]] += 1.0;
    const double scale = 1.0 / (cnt ? (double)cnt : 1.0);
    for (int j = 0; j < K; ++j)
      if (hist[g][j] != ref[j] * scale) { ++bad; fprintf(stderr, "hist[%d][%d]\n", g, j); break; }
  }
  {
    static double ref[K][D];
    static int64_t rc[K];
    memset(ref, 0, sizeof ref);
    memset(rc, 0, sizeof rc);
    for (int i = 0; i < N; ++i) {
      rc[i1[i]]++;
      for (int t = 0; t < D; ++t) ref[i1[i]][t] += (double)x[i][t];
    }
    for (int j = 0; j < K; ++j) {
      if (rc[j] != counts[j]) { ++bad; fprintf(stderr, "counts[%d]\n", j); }
      for (int t = 0; t < D; ++t)
        if (fabs(ref[j][t] - sums[j][t]) > 1e-9 * (fabs(ref[j][t]) + 1.0)) { ++bad; fprintf(stderr, "sums[%d][%d]\n", j, t); break; }
    }
  }
  /* a caller that keeps a codebook: auxiliary block once, then searches that bring it along; and
   * the entry point that allocates nothing -- both must repeat the answer above bit for bit */
  {
    float* daux;
    int32_t* dj;
    float* de;
    static int32_t j1a[N], j1s[N];
    static float e1a[N], e1s[N];
    HIPCHECK(hipMalloc((void**)&daux, (size_t)ic_codebook_aux_floats(K) * 4));
    HIPCHECK(hipMalloc((void**)&dj, N * 4));
    HIPCHECK(hipMalloc((void**)&de, N * 4));
    ICCHECK(ic_codebook_aux(h, dc, K, D, daux, s));
    ICCHECK(ic_assign_top1(h, dx, N, dc, K, D, daux, dj, de, s));
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipMemcpy(j1a, dj, sizeof j1a, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(e1a, de, sizeof e1a, hipMemcpyDeviceToHost));
    ICCHECK(ic_assign_top1_static(h, dx, N, dc, K, D, daux, dj, de, s));
    HIPCHECK(hipStreamSynchronize(s));
    HIPCHECK(hipMemcpy(j1s, dj, sizeof j1s, hipMemcpyDeviceToHost));
    HIPCHECK(hipMemcpy(e1s, de, sizeof e1s, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i)
      if (j1a[i] != i1[i] || j1s[i] != i1[i] || e1a[i] != d1[i] || e1s[i] != d1[i]) {
        ++bad;
        fprintf(stderr, "row %d: aux / static entry differs\n", i);
        break;
      }
    if (ic_assign_top1_static(h, dx, (int64_t)1 << 24, dc, K, D, daux, dj, de, s) != IC_ELIMIT) {
      ++bad;
      fprintf(stderr, "static entry accepted 16M rows\n");
    }
    HIPCHECK(hipFree(daux));
    HIPCHECK(hipFree(dj));
    HIPCHECK(hipFree(de));
  }
  /* error convention */
  if (ic_assign_top1(h, dx, N, dc, K, 4096, NULL, di1, dd1, s) != IC_EINVAL) { ++bad; fprintf(stderr, "d=4096 accepted\n"); }
  if (ic_hist(NULL, di1, doff, N_IMG, K, IC_NORM_L1, 0.5, dhist, s) != IC_EHANDLE) { ++bad; fprintf(stderr, "NULL handle accepted\n"); }

  ICCHECK(ic_destroy(h));
  printf("c_abi_consumer: %d rows x %d words, %d problems\n", N, K, bad);
  return bad ? 1 : 0;
}
