// Post-stage of the logo matcher (model/robust_matching.py:81-95 and :156-175, :291-365), batched
// over all candidate images of a query:
//
//   k_count_matches / k_scan_counts / k_compact_matches
//       ratio-test mask -> per-candidate lists of (query row, candidate row) and of matched
//       keypoint pairs (x_q, y_q, x_im, y_im), in ascending candidate-row order (stable)
//   k_ransac_score     one lane per (candidate, iteration): homography through the four sampled
//                      pairs, inliers among the remaining pairs counted
//   k_ransac_refit     one wave per candidate: first iteration with the most inliers, direct linear
//                      transform over its inliers, mean reprojection error, acceptance, mask
//
// The reference's loop keeps a model only when it has strictly more inliers than the best so far and
// stops when every pair is an inlier, so its result is that of the FIRST iteration that reaches the
// maximum count; scoring all iterations at once and taking the first maximum gives the same answer
// for the same samples.
#include "ic_internal.h"

namespace ic {

namespace {

constexpr int kItersPerWg = 64;   // iterations (= lanes) per workgroup of k_ransac_score
constexpr int kLdsPairs = 1024;   // pairs of a candidate kept in LDS while its iterations run
constexpr double kProjEps = 0.0000000001;  // "epsilon for non zero division", robust_matching.py:279

// ---------------------------------------------------------------------------------------------
// compaction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_count_matches(const uint8_t* __restrict__ mask,
                                                      const int64_t* __restrict__ row_off,
                                                      int64_t* __restrict__ counts) {
  const int64_t c = blockIdx.x;
  const int64_t r0 = row_off[c], r1 = row_off[c + 1];
  int cnt = 0;
  for (int64_t j = r0 + threadIdx.x; j < r1; j += 64) cnt += mask[j] != 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
  if (threadIdx.x == 0) counts[c] = cnt;
}

// out[0] = 0, out[i + 1] = counts[0] + ... + counts[i]: one workgroup of 1024 threads; every
// thread sums a contiguous segment (its loads are independent, so they are all in flight at
// once), the segment totals are scanned by wave shuffles and a 16-entry LDS table, then every
// thread writes the running sums of its segment.
__global__ __launch_bounds__(1024) void k_scan_counts(const int64_t* __restrict__ counts,
                                                      int64_t n, int64_t* __restrict__ out) {
  __shared__ int64_t wave_total[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t seg = (n + 1023) / 1024;
  const int64_t lo = (int64_t)tid * seg;
  const int64_t hi = lo + seg < n ? lo + seg : n;
  int64_t total = 0;
#pragma unroll 8
  for (int64_t i = lo; i < hi; ++i) total += counts[i];
  int64_t incl = total;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int64_t t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wave_total[wave] = incl;
  __syncthreads();
  int64_t run = incl - total;                 // exclusive prefix inside the wave
  for (int w = 0; w < wave; ++w) run += wave_total[w];
  if (tid == 0) out[0] = 0;
  for (int64_t i = lo; i < hi; ++i) {
    run += counts[i];
    out[i + 1] = run;
  }
}

__global__ __launch_bounds__(64) void k_compact_matches(
    const uint8_t* __restrict__ mask, const int32_t* __restrict__ i1,
    const int64_t* __restrict__ row_off, const int64_t* __restrict__ out_off,
    const double* __restrict__ kp_q, int64_t n_q, const double* __restrict__ kp_rows,
    int32_t* __restrict__ match_q, int32_t* __restrict__ match_im, double* __restrict__ pts) {
  const int64_t c = blockIdx.x;
  const int64_t r0 = row_off[c], r1 = row_off[c + 1];
  int64_t pos = out_off[c];
  const int lane = threadIdx.x;
  for (int64_t j0 = r0; j0 < r1; j0 += 64) {
    const int64_t j = j0 + lane;
    const bool m = j < r1 && mask[j] != 0;
    const unsigned long long b = __ballot(m);
    const int before = __popcll(b & ((1ull << lane) - 1ull));
    if (m) {
      const int64_t o = pos + before;
      int32_t q = i1[j];
      match_q[o] = q;
      match_im[o] = (int32_t)(j - r0);
      if (pts) {
        if (q < 0) q = 0;
        if (q >= n_q) q = (int32_t)(n_q - 1);
        pts[o * 4 + 0] = kp_q[(int64_t)q * 2 + 0];
        pts[o * 4 + 1] = kp_q[(int64_t)q * 2 + 1];
        pts[o * 4 + 2] = kp_rows[j * 2 + 0];
        pts[o * 4 + 3] = kp_rows[j * 2 + 1];
      }
    }
    pos += __popcll(b);
  }
}

// ---------------------------------------------------------------------------------------------
// homography helpers (every lane of a wave computes the same model redundantly, in registers)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double tri(double ax, double ay, double bx, double by, double cx,
                                      double cy) {
  return (bx - ax) * (cy - ay) - (cx - ax) * (by - ay);
}

// columns c_i * (x_i, y_i, 1) of the map from the projective basis to four points
__device__ __forceinline__ void basis_map(const double* p /*x0,y0,..,x3,y3*/, double* A) {
  const double c0 = tri(p[6], p[7], p[2], p[3], p[4], p[5]);
  const double c1 = tri(p[0], p[1], p[6], p[7], p[4], p[5]);
  const double c2 = tri(p[0], p[1], p[2], p[3], p[6], p[7]);
  A[0] = c0 * p[0]; A[1] = c1 * p[2]; A[2] = c2 * p[4];
  A[3] = c0 * p[1]; A[4] = c1 * p[3]; A[5] = c2 * p[5];
  A[6] = c0;        A[7] = c1;        A[8] = c2;
}

// The homography that maps four source points onto four destination points, scaled to H[8] = 1.
// Degenerate samples (three collinear points) give non-finite entries; every comparison against
// them is false, so such a model simply has no inliers.
__device__ __forceinline__ void homography4(const double* src, const double* dst, double* H) {
  double A[9], B[9], J[9];
  basis_map(src, A);
  basis_map(dst, B);
  J[0] = A[4] * A[8] - A[5] * A[7];
  J[1] = A[2] * A[7] - A[1] * A[8];
  J[2] = A[1] * A[5] - A[2] * A[4];
  J[3] = A[5] * A[6] - A[3] * A[8];
  J[4] = A[0] * A[8] - A[2] * A[6];
  J[5] = A[2] * A[3] - A[0] * A[5];
  J[6] = A[3] * A[7] - A[4] * A[6];
  J[7] = A[1] * A[6] - A[0] * A[7];
  J[8] = A[0] * A[4] - A[1] * A[3];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c)
      H[r * 3 + c] = B[r * 3] * J[c] + B[r * 3 + 1] * J[3 + c] + B[r * 3 + 2] * J[6 + c];
  const double inv = 1.0 / H[8];
#pragma unroll
  for (int e = 0; e < 9; ++e) H[e] *= inv;
}

// HomographyModel.get_error (model/robust_matching.py:273-281) for one pair
__device__ __forceinline__ double reproj_error(const double* H, double x, double y, double u,
                                               double v) {
  const double w = H[6] * x + H[7] * y + H[8];
  const double den = w + kProjEps;
  const double tx = (H[0] * x + H[1] * y + H[2]) / den;
  const double ty = (H[3] * x + H[4] * y + H[5]) / den;
  const double tz = w / den;
  const double dx = u - tx, dy = v - ty, dz = 1.0 - tz;
  return sqrt(dx * dx + dy * dy + dz * dz);
}

// Squared distance between the projected source point and the destination point, H[8] = 1: the
// error OpenCV's homography estimator thresholds at th^2 (the rule behind the reference's default
// cv2.findHomography(..., cv2.RANSAC, th), model/robust_matching.py:214-216).
__device__ __forceinline__ double cv_sq_error(const double* H, double x, double y, double u,
                                              double v) {
  const double ww = 1.0 / (H[6] * x + H[7] * y + 1.0);
  const double dx = (H[0] * x + H[1] * y + H[2]) * ww - u;
  const double dy = (H[3] * x + H[4] * y + H[5]) * ww - v;
  return dx * dx + dy * dy;
}

// Iteration limit after a model with `good` of n inliers at confidence 0.995 and four-point
// samples, never above `limit` (OpenCV's RANSACUpdateNumIters).
__device__ __forceinline__ int cv_num_iters(int good, int n, int limit) {
  const double w = (double)good / (double)n;
  const double denom = 1.0 - w * w * w * w;
  if (denom < 2.2250738585072014e-308) return 0;
  const double x = log(1.0 - 0.995) / log(denom);
  return x >= (double)limit ? limit : (int)rint(x);
}

__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z += 0x9e3779b97f4a7c15ull;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// four distinct pair indices of iteration `it` of candidate `cand`: the caller's (clamped into
// range), or drawn without replacement from a counter-based generator
__device__ __forceinline__ void sample4(const int32_t* __restrict__ samples, int64_t cand, int it,
                                        int n_iter, int n, uint64_t seed, int* s) {
  if (samples) {
    const int32_t* sp = samples + ((int64_t)cand * n_iter + it) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      int v = sp[k];
      v = v < 0 ? 0 : v;
      s[k] = v >= n ? n - 1 : v;
    }
    return;
  }
  uint64_t state = mix64(seed ^ mix64((uint64_t)cand * 0x100000001b3ull + (uint64_t)it));
  int sorted[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    state = mix64(state);
    int v = (int)(state % (uint64_t)(n - k));
    // skip over the indices already taken (kept sorted), which makes the draw uniform over the rest
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < k && v >= sorted[t]) ++v;
    s[k] = v;
    int ins = v;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int lo = ins < sorted[t] ? ins : sorted[t];
      const int hi = ins < sorted[t] ? sorted[t] : ins;
      sorted[t] = lo;
      ins = hi;
    }
  }
}

__device__ __forceinline__ void load_sample(const double* __restrict__ pts, const int* s,
                                            double* src, double* dst) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double* p = pts + (int64_t)s[k] * 4;
    src[2 * k] = p[0];
    src[2 * k + 1] = p[1];
    dst[2 * k] = p[2];
    dst[2 * k + 1] = p[3];
  }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_ransac_score: n_inl[cand][it] = pairs outside the sample whose reprojection error is < th
// (cv_rules: all pairs whose squared error is <= th^2).
// One LANE per iteration: every lane builds the model of its own sample (no redundant work) and
// walks the candidate's pairs, which all lanes read at the same address (an LDS broadcast).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kItersPerWg) void k_ransac_score(
    const double* __restrict__ pts_all, const int64_t* __restrict__ off, int n_iter,
    const int32_t* __restrict__ samples, uint64_t seed, double th, int min_pairs, int cv_rules,
    int32_t* __restrict__ n_inl) {
  __shared__ double lpts[kLdsPairs * 4];
  const int64_t cand = blockIdx.x;
  const int64_t p0 = off[cand];
  const int n = (int)(off[cand + 1] - p0);
  const int it = (int)blockIdx.y * kItersPerWg + (int)threadIdx.x;
  // a candidate with fewer pairs than the acceptance rule demands can never be accepted: no
  // model is scored for it (the refit kernel rejects it on the same test)
  if (n < min_pairs) {
    if (it < n_iter) n_inl[cand * n_iter + it] = 0;
    return;
  }
  const double* pts = pts_all + p0 * 4;
  if (n <= kLdsPairs) {
    for (int e = threadIdx.x; e < n * 4; e += kItersPerWg) lpts[e] = pts[e];
    pts = lpts;
  }
  __syncthreads();
  if (it >= n_iter) return;
  int s[4];
  sample4(samples, cand, it, n_iter, n, seed, s);
  double src[8], dst[8], H[9];
  load_sample(pts, s, src, dst);
  homography4(src, dst, H);
  int cnt = 0;
  if (cv_rules) {
    // every pair counts, the sample's own included, at squared error <= th^2
    const double th2 = th * th;
    for (int j = 0; j < n; ++j) {
      const double* p = pts + (int64_t)j * 4;
      cnt += cv_sq_error(H, p[0], p[1], p[2], p[3]) <= th2 ? 1 : 0;
    }
  } else {
    for (int j = 0; j < n; ++j) {
      const double* p = pts + (int64_t)j * 4;
      const double e = reproj_error(H, p[0], p[1], p[2], p[3]);
      const bool in_sample = j == s[0] || j == s[1] || j == s[2] || j == s[3];
      cnt += (!in_sample && e < th) ? 1 : 0;
    }
  }
  n_inl[cand * n_iter + it] = cnt;
}

// ---------------------------------------------------------------------------------------------
// k_ransac_refit
// ---------------------------------------------------------------------------------------------
// In-register Cholesky factor of the packed symmetric 9x9 matrix a (row-major lower triangle,
// a[r*(r+1)/2 + c], c <= r), in place; the diagonal slots end up holding the RECIPROCALS of the
// factor's diagonal, so that the solves below multiply instead of divide.
__device__ __forceinline__ void cholesky9(double* a) {
#pragma unroll
  for (int j = 0; j < 9; ++j) {
    double d = a[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= a[j * (j + 1) / 2 + k] * a[j * (j + 1) / 2 + k];
    d = sqrt(d);
    const double inv = 1.0 / d;
    a[j * (j + 1) / 2 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < 9; ++i) {
      double s = a[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= a[i * (i + 1) / 2 + k] * a[j * (j + 1) / 2 + k];
      a[i * (i + 1) / 2 + j] = s * inv;
    }
  }
}

// v <- (L L^T)^-1 v
__device__ __forceinline__ void chol_solve9(const double* L, double* v) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    double s = v[i];
#pragma unroll
    for (int k = 0; k < i; ++k) s -= L[i * (i + 1) / 2 + k] * v[k];
    v[i] = s * L[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = 8; i >= 0; --i) {
    double s = v[i];
#pragma unroll
    for (int k = i + 1; k < 9; ++k) s -= L[k * (k + 1) / 2 + i] * v[k];
    v[i] = s * L[i * (i + 1) / 2 + i];
  }
}

// ---------------------------------------------------------------------------------------------
// The polish cv2.findHomography applies to the refitted model: Levenberg-Marquardt over the first
// eight entries of H (H[8] = 1) on the inliers' reprojection residuals, at most ten steps, with
// the damping schedule of OpenCV's LM solver (gain ratio above 0.75 halves the damping and drops
// it to zero below a floor, below 0.25 multiplies it by a factor between 2 and 10).  One wave:
// the sums over the inliers are wave reductions, the 8x8 solves are done by lane 0 in LDS.
// ---------------------------------------------------------------------------------------------
struct LmShared {
  double A[64];    // J^T J
  double Ap[64];   // damped copy, overwritten by its Cholesky factor
  double v[8];     // J^T r
  double d[8];     // step
  double diag_inv; // largest diagonal entry of A^-1 (only when the damping restarts from zero)
  int solved;
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmax(v, __shfl_xor(v, off));
  return v;
}

// residuals of the inliers at parameters h: S = sum r^2, rmax = max |r|; with `sh` also J^T J
// and J^T r (written to LDS by lane 0, visible after the barrier this function ends with)
__device__ void lm_residuals(const double* pts, const uint8_t* mask, int n, int lane,
                             const double* h, LmShared* sh, double& S, double& rmax) {
  double gg[6] = {0, 0, 0, 0, 0, 0};   // g g^T, g = (a, b, c):  aa ab ac bb bc cc
  double gx[6] = {0, 0, 0, 0, 0, 0};   // xi * g (a, b)^T:       aa ab ba bb ca cb
  double gy[6] = {0, 0, 0, 0, 0, 0};   // yi * g (a, b)^T
  double qq[3] = {0, 0, 0};            // (xi^2 + yi^2) (a, b)(a, b)^T: aa ab bb
  double jv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double s2 = 0, rm = 0;
  for (int j = lane; j < n; j += 64) {
    if (!mask[j]) continue;
    const double* p = pts + (int64_t)j * 4;
    const double x = p[0], y = p[1];
    double ww = h[6] * x + h[7] * y + 1.0;
    ww = fabs(ww) > 2.220446049250313e-16 ? 1.0 / ww : 0.0;
    const double xi = (h[0] * x + h[1] * y + h[2]) * ww;
    const double yi = (h[3] * x + h[4] * y + h[5]) * ww;
    const double r1 = xi - p[2], r2 = yi - p[3];
    s2 += r1 * r1 + r2 * r2;
    rm = fmax(rm, fmax(fabs(r1), fabs(r2)));
    if (sh) {
      const double a = x * ww, b = y * ww, c = ww;
      gg[0] += a * a; gg[1] += a * b; gg[2] += a * c; gg[3] += b * b; gg[4] += b * c; gg[5] += c * c;
      gx[0] += xi * a * a; gx[1] += xi * a * b; gx[2] += xi * b * a; gx[3] += xi * b * b;
      gx[4] += xi * c * a; gx[5] += xi * c * b;
      gy[0] += yi * a * a; gy[1] += yi * a * b; gy[2] += yi * b * a; gy[3] += yi * b * b;
      gy[4] += yi * c * a; gy[5] += yi * c * b;
      const double q = xi * xi + yi * yi;
      qq[0] += q * a * a; qq[1] += q * a * b; qq[2] += q * b * b;
      jv[0] += a * r1; jv[1] += b * r1; jv[2] += c * r1;
      jv[3] += a * r2; jv[4] += b * r2; jv[5] += c * r2;
      const double t = xi * r1 + yi * r2;
      jv[6] -= a * t; jv[7] -= b * t;
    }
  }
  S = wave_sum(s2);
  rmax = wave_max(rm);
  if (!sh) return;
#pragma unroll
  for (int e = 0; e < 6; ++e) {
    gg[e] = wave_sum(gg[e]);
    gx[e] = wave_sum(gx[e]);
    gy[e] = wave_sum(gy[e]);
  }
#pragma unroll
  for (int e = 0; e < 3; ++e) qq[e] = wave_sum(qq[e]);
#pragma unroll
  for (int e = 0; e < 8; ++e) jv[e] = wave_sum(jv[e]);
  __syncthreads();      // every lane is done with the previous A and v
  if (lane == 0) {
    double* A = sh->A;
    for (int e = 0; e < 64; ++e) A[e] = 0.0;
    const double g3[3][3] = {{gg[0], gg[1], gg[2]}, {gg[1], gg[3], gg[4]}, {gg[2], gg[4], gg[5]}};
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        A[r * 8 + c] = g3[r][c];
        A[(3 + r) * 8 + 3 + c] = g3[r][c];
      }
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 2; ++c) {
        A[r * 8 + 6 + c] = A[(6 + c) * 8 + r] = -gx[r * 2 + c];
        A[(3 + r) * 8 + 6 + c] = A[(6 + c) * 8 + 3 + r] = -gy[r * 2 + c];
      }
    A[6 * 8 + 6] = qq[0];
    A[6 * 8 + 7] = A[7 * 8 + 6] = qq[1];
    A[7 * 8 + 7] = qq[2];
    for (int e = 0; e < 8; ++e) sh->v[e] = jv[e];
  }
  __syncthreads();
}

// lane 0: Cholesky factor of the 8x8 matrix M (full storage, lower part used) in place; false
// when a pivot is not positive
__device__ bool chol8(double* M) {
  for (int j = 0; j < 8; ++j) {
    double d = M[j * 8 + j];
    for (int k = 0; k < j; ++k) d -= M[j * 8 + k] * M[j * 8 + k];
    if (!(d > 0.0)) return false;
    d = sqrt(d);
    M[j * 8 + j] = d;
    for (int i = j + 1; i < 8; ++i) {
      double t = M[i * 8 + j];
      for (int k = 0; k < j; ++k) t -= M[i * 8 + k] * M[j * 8 + k];
      M[i * 8 + j] = t / d;
    }
  }
  return true;
}

__device__ void chol8_solve(const double* L, double* b) {
  for (int i = 0; i < 8; ++i) {
    double t = b[i];
    for (int k = 0; k < i; ++k) t -= L[i * 8 + k] * b[k];
    b[i] = t / L[i * 8 + i];
  }
  for (int i = 7; i >= 0; --i) {
    double t = b[i];
    for (int k = i + 1; k < 8; ++k) t -= L[k * 8 + i] * b[k];
    b[i] = t / L[i * 8 + i];
  }
}

__device__ void lm_polish(const double* pts, const uint8_t* mask, int n, int lane, double* H,
                          LmShared* sh) {
  const double eps = 2.220446049250313e-16;
  double x[8], xd[8], D[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] = H[e];
  double S, rmax;
  lm_residuals(pts, mask, n, lane, x, sh, S, rmax);
#pragma unroll
  for (int e = 0; e < 8; ++e) D[e] = sh->A[e * 8 + e];
  double lambda = 1.0, lc = 0.75;
#pragma unroll 1
  for (int iter = 0; iter < 10;) {
    if (lane == 0) {
      for (int e = 0; e < 64; ++e) sh->Ap[e] = sh->A[e];
      for (int e = 0; e < 8; ++e) {
        sh->Ap[e * 8 + e] += lambda * D[e];
        sh->d[e] = sh->v[e];
      }
      sh->solved = chol8(sh->Ap) ? 1 : 0;
      if (sh->solved) chol8_solve(sh->Ap, sh->d);
    }
    __syncthreads();
    if (!sh->solved) break;      // the damped normal equations are not positive definite
    double d[8], dmax = 0, dv = 0, dS = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      d[e] = sh->d[e];
      xd[e] = x[e] - d[e];
      dmax = fmax(dmax, fabs(d[e]));
      dv += d[e] * sh->v[e];
    }
#pragma unroll 1
    for (int r = 0; r < 8; ++r) {
      double ad = 0;
      for (int c = 0; c < 8; ++c) ad += sh->A[r * 8 + c] * d[c];
      dS += d[r] * (2.0 * sh->v[r] - ad);
    }
    double Sd, rdmax;
    lm_residuals(pts, mask, n, lane, xd, nullptr, Sd, rdmax);
    const double R = (S - Sd) / (fabs(dS) > eps ? dS : 1.0);
    if (R > 0.75) {
      lambda *= 0.5;
      if (lambda < lc) lambda = 0.0;
    } else if (R < 0.25) {
      double nu = (Sd - S) / (fabs(dv) > eps ? dv : 1.0) + 2.0;
      nu = fmin(fmax(nu, 2.0), 10.0);
      if (lambda == 0.0) {
        __syncthreads();
        if (lane == 0) {
          for (int e = 0; e < 64; ++e) sh->Ap[e] = sh->A[e];
          double mx = eps;
          if (chol8(sh->Ap)) {
            for (int c = 0; c < 8; ++c) {
              double col[8] = {0, 0, 0, 0, 0, 0, 0, 0};
              col[c] = 1.0;
              chol8_solve(sh->Ap, col);
              mx = fmax(mx, fabs(col[c]));
            }
          }
          sh->diag_inv = mx;
        }
        __syncthreads();
        lambda = lc = 1.0 / sh->diag_inv;
        nu *= 0.5;
      }
      lambda *= nu;
    }
    if (Sd < S) {
      S = Sd;
#pragma unroll
      for (int e = 0; e < 8; ++e) x[e] = xd[e];
      lm_residuals(pts, mask, n, lane, x, sh, S, rmax);
    }
    ++iter;
    if (!(dmax >= eps && rmax >= eps)) break;
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) H[e] = x[e];
  H[8] = 1.0;
}

// CV: OpenCV's rules (adaptive iteration limit, squared-error inliers, LM polish); otherwise the
// reference's own loop.  Two instantiations, so that the plain loop does not carry the polish's
// registers and LDS.
template <bool CV>
__global__ __launch_bounds__(64) void k_ransac_refit(
    const double* __restrict__ pts_all, const int64_t* __restrict__ off, int n_iter,
    const int32_t* __restrict__ samples, uint64_t seed, double th, int min_inliers,
    int flags, const int32_t* __restrict__ n_inl, double* __restrict__ H_out,
    int32_t* __restrict__ n_inliers, double* __restrict__ best_err, int32_t* __restrict__ best_it,
    uint8_t* __restrict__ mask_all) {
  const int64_t cand = blockIdx.x;
  const int64_t p0 = off[cand];
  const int n = (int)(off[cand + 1] - p0);
  const double* pts = pts_all + p0 * 4;
  uint8_t* mask = mask_all + p0;
  const int lane = threadIdx.x;

  const int min_pairs = min_inliers > 4 ? min_inliers : 4;
  bool ok = n >= min_pairs && n_iter > 0;
  const bool accept_mean_err = (flags & IC_RANSAC_MEAN_ERR) != 0;
  constexpr bool cv_rules = CV;
  int bc = -1, bi = 0x7fffffff;
  if (!cv_rules) {
    // first iteration with the largest count
    for (int it = lane; it < n_iter; it += 64) {
      const int c = n_inl[cand * n_iter + it];
      if (c > bc) {
        bc = c;
        bi = it;
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      const int oc = __shfl_xor(bc, o), oi = __shfl_xor(bi, o);
      if (oc > bc || (oc == bc && oi < bi)) {
        bc = oc;
        bi = oi;
      }
    }
  } else if (ok) {
    // OpenCV's sequential loop, 64 iterations at a time.  A model is kept when it has strictly
    // more inliers than every earlier one (and more than three); keeping it lowers the iteration
    // limit to cv_num_iters(its count), and iteration it + 1 runs while it + 1 is below the limit
    // in force after iteration it.  The limit only depends on the best count so far, so a prefix
    // maximum over (count, earliest iteration) gives every lane the state the loop would have
    // after its iteration, and the first lane whose successor would not run ends the search.
    int64_t run_key = -1;
    for (int c0 = 0; c0 < n_iter; c0 += 64) {
      const int it = c0 + lane;
      const int c = it < n_iter ? n_inl[cand * n_iter + it] : 0;
      int64_t key = c > 3 ? (((int64_t)c << 32) | (int64_t)(0x7fffffff - it)) : -1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int64_t other = __shfl_up(key, o);
        if (lane >= o && other > key) key = other;
      }
      if (run_key > key) key = run_key;
      const int best = key >= 0 ? (int)(key >> 32) : 0;
      const int limit = best > 3 ? cv_num_iters(best, n, n_iter) : n_iter;
      const bool last = it < n_iter && it + 1 >= limit;
      const unsigned long long ends = __ballot(last);
      const int src_lane = ends ? __builtin_ctzll(ends) : 63;
      run_key = __shfl(key, src_lane);
      if (ends) break;
    }
    if (run_key >= 0) {
      bc = (int)(run_key >> 32);
      bi = 0x7fffffff - (int)(run_key & 0x7fffffff);
    } else {
      ok = false;          // no sample gave a model with more than three inliers
    }
  }

  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  double err_mean = __builtin_inf();
  int count = 0;
  if (ok) {
    int s[4];
    sample4(samples, cand, bi, n_iter, n, seed, s);
    double src[8], dst[8], H4[9];
    load_sample(pts, s, src, dst);
    homography4(src, dst, H4);
    // inlier flags (the reference's loop: the sample itself always belongs to the consensus
    // set; OpenCV's rule: whatever lies within the threshold) and centroids
    double sx = 0, sy = 0, su = 0, sv = 0, sc = 0;
    const double th2 = th * th;
    for (int j = lane; j < n; j += 64) {
      const double* p = pts + (int64_t)j * 4;
      const bool in_sample = j == s[0] || j == s[1] || j == s[2] || j == s[3];
      const bool in = cv_rules ? cv_sq_error(H4, p[0], p[1], p[2], p[3]) <= th2
                               : in_sample || reproj_error(H4, p[0], p[1], p[2], p[3]) < th;
      mask[j] = in ? 1 : 0;
      const double f = in ? 1.0 : 0.0;
      sx += f * p[0]; sy += f * p[1]; su += f * p[2]; sv += f * p[3]; sc += f;
    }
    sx = wave_sum(sx); sy = wave_sum(sy); su = wave_sum(su); sv = wave_sum(sv); sc = wave_sum(sc);
    count = (int)sc;
    const double cMx = sx / sc, cMy = sy / sc, cmx = su / sc, cmy = sv / sc;
    double ax = 0, ay = 0, au = 0, av = 0;
    for (int j = lane; j < n; j += 64) {
      const double* p = pts + (int64_t)j * 4;
      const double f = mask[j] ? 1.0 : 0.0;
      ax += f * fabs(p[0] - cMx); ay += f * fabs(p[1] - cMy);
      au += f * fabs(p[2] - cmx); av += f * fabs(p[3] - cmy);
    }
    ax = wave_sum(ax); ay = wave_sum(ay); au = wave_sum(au); av = wave_sum(av);
    const double eps = 2.220446049250313e-16;
    if (ax < eps || ay < eps || au < eps || av < eps) ok = false;
    if (ok) {
      const double sMx = sc / ax, sMy = sc / ay, smx = sc / au, smy = sc / av;
      // the four 3x3 moment blocks of L^T L: S0 = sum a a^T, Sx = sum x a a^T, Sy = sum y a a^T,
      // Sr = sum (x^2 + y^2) a a^T with a = (X, Y, 1) the normalised source point and (x, y) the
      // normalised destination point; packed as (XX, XY, X1, YY, Y1, 11)
      double S0[6] = {0, 0, 0, 0, 0, 0}, Sx[6] = {0, 0, 0, 0, 0, 0}, Sy[6] = {0, 0, 0, 0, 0, 0},
             Sr[6] = {0, 0, 0, 0, 0, 0};
      for (int j = lane; j < n; j += 64) {
        const double* p = pts + (int64_t)j * 4;
        const double f = mask[j] ? 1.0 : 0.0;
        const double X = (p[0] - cMx) * sMx, Y = (p[1] - cMy) * sMy;
        const double x = (p[2] - cmx) * smx, y = (p[3] - cmy) * smy;
        const double m[6] = {f * X * X, f * X * Y, f * X, f * Y * Y, f * Y, f};
        const double r2 = x * x + y * y;
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          S0[e] += m[e];
          Sx[e] += x * m[e];
          Sy[e] += y * m[e];
          Sr[e] += r2 * m[e];
        }
      }
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        S0[e] = wave_sum(S0[e]);
        Sx[e] = wave_sum(Sx[e]);
        Sy[e] = wave_sum(Sy[e]);
        Sr[e] = wave_sum(Sr[e]);
      }
      // packed lower triangle of L^T L + shift * I
      const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
      double a[45];
#pragma unroll
      for (int r = 0; r < 9; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) {
          const int br = r / 3, bc2 = c / 3, e = sym[r % 3][c % 3];
          double v;
          if (br == bc2) v = br == 2 ? Sr[e] : S0[e];
          else if (br == 2) v = bc2 == 0 ? -Sx[e] : -Sy[e];
          else v = 0.0;
          a[r * (r + 1) / 2 + c] = v;
        }
      const double trace = 2.0 * (S0[0] + S0[3] + S0[5]) + Sr[0] + Sr[3] + Sr[5];
      const double shift = trace * 1e-12;
#pragma unroll
      for (int r = 0; r < 9; ++r) a[r * (r + 1) / 2 + r] += shift;
      cholesky9(a);
      // inverse iteration towards the eigenvector of the smallest eigenvalue
      double v[9] = {1, 1, 1, 1, 1, 1, 1, 1, 1};
#pragma unroll 1
      for (int rep = 0; rep < 8; ++rep) {
        chol_solve9(a, v);
        double nrm = 0;
#pragma unroll
        for (int e = 0; e < 9; ++e) nrm += v[e] * v[e];
        nrm = 1.0 / sqrt(nrm);
#pragma unroll
        for (int e = 0; e < 9; ++e) v[e] *= nrm;
      }
      // H = invHnorm * H0 * Hnorm2, scaled to H[8] = 1
      double T[9];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        T[c] = v[c] / smx + cmx * v[6 + c];
        T[3 + c] = v[3 + c] / smy + cmy * v[6 + c];
        T[6 + c] = v[6 + c];
      }
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        H[r * 3 + 0] = T[r * 3 + 0] * sMx;
        H[r * 3 + 1] = T[r * 3 + 1] * sMy;
        H[r * 3 + 2] = T[r * 3 + 2] - T[r * 3 + 0] * cMx * sMx - T[r * 3 + 1] * cMy * sMy;
      }
      const double inv = 1.0 / H[8];
#pragma unroll
      for (int e = 0; e < 9; ++e) H[e] *= inv;
      if constexpr (CV) {
        __shared__ LmShared lm_shared;
        bool finite = true;
#pragma unroll
        for (int e = 0; e < 9; ++e) finite = finite && fabs(H[e]) < __builtin_inf();
        if (finite) lm_polish(pts, mask, n, lane, H, &lm_shared);
        else ok = false;
      }
      double es = 0;
      for (int j = lane; j < n; j += 64) {
        const double* p = pts + (int64_t)j * 4;
        const double e = reproj_error(H, p[0], p[1], p[2], p[3]);
        es += mask[j] ? e : 0.0;
      }
      err_mean = wave_sum(es) / sc;
    }
  }
  bool accept = ok && count >= min_inliers;
  if (accept_mean_err) accept = accept && err_mean <= th;
  if (!accept)
    for (int j = lane; j < n; j += 64) mask[j] = 0;
  if (lane < 9) H_out[cand * 9 + lane] = accept ? H[lane] : 0.0;
  if (lane == 0) {
    n_inliers[cand] = accept ? count : 0;
    best_err[cand] = err_mean;
    if (best_it) best_it[cand] = ok ? bi : -1;
  }
}

}  // namespace

hipError_t launch_compact_matches(const uint8_t* mask, const int32_t* i1, const int64_t* row_off,
                                  int64_t n_cand, const double* kp_q, int64_t n_q,
                                  const double* kp_rows, int64_t* counts_ws, int64_t* out_off,
                                  int32_t* match_q, int32_t* match_im, double* pts, hipStream_t s) {
  hipLaunchKernelGGL(k_count_matches, dim3((unsigned)n_cand), dim3(64), 0, s, mask, row_off,
                     counts_ws);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, s, counts_ws, n_cand, out_off);
  hipLaunchKernelGGL(k_compact_matches, dim3((unsigned)n_cand), dim3(64), 0, s, mask, i1, row_off,
                     out_off, kp_q, n_q, kp_rows, match_q, match_im, pts);
  return hipGetLastError();
}

hipError_t launch_ransac(const double* pts, const int64_t* off, int64_t n_cand, int n_iter,
                         const int32_t* samples, uint64_t seed, double th, int min_inliers,
                         int flags, int32_t* n_inl_ws, double* H, int32_t* n_inliers,
                         double* best_err, int32_t* best_it, uint8_t* mask, hipStream_t s) {
  if (n_iter > 0) {
    const dim3 grid((unsigned)n_cand, (unsigned)((n_iter + kItersPerWg - 1) / kItersPerWg));
    hipLaunchKernelGGL(k_ransac_score, grid, dim3(kItersPerWg), 0, s, pts, off, n_iter, samples, seed, th,
                       min_inliers > 4 ? min_inliers : 4, (flags & IC_RANSAC_CV_RULES) != 0, n_inl_ws);
  }
  auto refit = (flags & IC_RANSAC_CV_RULES) ? k_ransac_refit<true> : k_ransac_refit<false>;
  hipLaunchKernelGGL(refit, dim3((unsigned)n_cand), dim3(64), 0, s, pts, off, n_iter,
                     samples, seed, th, min_inliers, flags, n_inl_ws, H, n_inliers,
                     best_err, best_it, mask);
  return hipGetLastError();
}

}  // namespace ic
