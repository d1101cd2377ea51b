// C ABI of libic_hip.so (declared in include/ic_hip.h): handle, workspace, argument checks and the
// launch sequences.  No kernel lives here; the one device symbol is the static scratch buffer of
// the entry point that allocates nothing.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "ic_internal.h"

namespace {

thread_local std::string t_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_err = buf;
  return code;
}

int fail_hip(hipError_t e, const char* what) {
  return fail((int)e, "%s: %s (hipError %d)", what, hipGetErrorString(e), (int)e);
}

constexpr uint64_t kMagic = 0x1c41ffa5d15c0de5ULL;

// Scratch of ic_assign_top1_static: device memory that exists from the moment the code object is
// loaded, one copy per GPU, so that entry point never has to allocate.  In host code the name
// `scratch` is only a handle for hipGetSymbolAddress (ic_create asks once per device and keeps the
// answer in the handle); every launch receives that address as an ordinary argument.
constexpr size_t kStaticScratchBytes = IC_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

struct Ctx {
  uint64_t magic = kMagic;
  int device = 0;
  int num_cus = 256;
  std::mutex mu;
  char* ws = nullptr;
  size_t ws_bytes = 0;
  char* static_scratch = nullptr;  // this device's address of `scratch`
  uint32_t* debug_count = nullptr; // device word that receives each search's flagged-row count
  hipEvent_t done = nullptr;      // recorded after each call's last launch
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  // optional timing of the distance kernel (ic_profile_begin / ic_profile_end)
  bool profiling = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
  size_t ev_used = 0;
};

Ctx* as_ctx(void* h) {
  Ctx* c = static_cast<Ctx*>(h);
  if (!c || c->magic != kMagic) return nullptr;
  return c;
}

size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// RAII: make the handle's device current for the call, restore the caller's afterwards.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess) { ok = false; return; }
    if (prev != dev) {
      err = hipSetDevice(dev);
      if (err != hipSuccess) ok = false;
    } else {
      prev = -1;
    }
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};

// Grow-only workspace.  Only this handle's calls use it, each ordered after the one before
// (order_after_previous / finish_call), so growing waits for the handle's last call -- not for
// the device -- before the old block is released.  ic_reserve_workspace sizes it up front.
int ensure_ws(Ctx* c, size_t bytes) {
  if (bytes <= c->ws_bytes) return IC_OK;
  if (c->have_last) {
    hipError_t e = hipEventSynchronize(c->done);
    if (e != hipSuccess) return fail_hip(e, "hipEventSynchronize before workspace growth");
  }
  if (c->ws) {
    (void)hipFree(c->ws);
    c->ws = nullptr;
    c->ws_bytes = 0;
  }
  const size_t want = align_up(bytes + bytes / 4, 1 << 20);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->ws), want);
  if (e != hipSuccess) {
    c->ws = nullptr;
    return fail(IC_ENOMEM, "workspace allocation of %zu bytes failed: %s", want,
                hipGetErrorString(e));
  }
  c->ws_bytes = want;
  return IC_OK;
}

// The workspace is shared by consecutive calls: a call on another stream than the previous one
// first waits (on the device) for the previous call's work.
int order_after_previous(Ctx* c, hipStream_t s) {
  if (c->have_last && c->last_stream != s) {
    hipError_t e = hipStreamWaitEvent(s, c->done, 0);
    if (e != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
  }
  return IC_OK;
}

int finish_call(Ctx* c, hipStream_t s) {
  hipError_t e = hipEventRecord(c->done, s);
  if (e != hipSuccess) return fail_hip(e, "hipEventRecord");
  c->last_stream = s;
  c->have_last = true;
  return IC_OK;
}

// A call that fails half way (a launch is refused) may already have enqueued kernels that use the
// workspace: leaving through an error return, the call is closed like a finished one, so that
// the next call -- on whatever stream -- is still ordered behind that work.  The error being
// returned is the one the caller sees; a failure of the record itself is not reported over it.
struct CallScope {
  Ctx* c;
  hipStream_t s;
  bool open = true;
  CallScope(Ctx* ctx, hipStream_t stream) : c(ctx), s(stream) {}
  int finish() {
    open = false;
    return finish_call(c, s);
  }
  ~CallScope() {
    if (open && hipEventRecord(c->done, s) == hipSuccess) {
      c->last_stream = s;
      c->have_last = true;
    }
  }
};

// When the handle is profiling: records the start event of a dominant-kernel launch on `s` and
// hands back the stop event to record after it (null otherwise).
int profile_open(Ctx* c, hipStream_t s, hipEvent_t* stop) {
  *stop = nullptr;
  if (!c->profiling) return IC_OK;
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a);
    if (e != hipSuccess) return fail_hip(e, "hipEventCreate");
    e = hipEventCreate(&b);
    if (e != hipSuccess) return fail_hip(e, "hipEventCreate");
    c->ev_pool.emplace_back(a, b);
  }
  hipError_t e = hipEventRecord(c->ev_pool[c->ev_used].first, s);
  if (e != hipSuccess) return fail_hip(e, "hipEventRecord");
  *stop = c->ev_pool[c->ev_used].second;
  c->ev_used++;
  return IC_OK;
}

#define IC_HIP(call, what)                            \
  do {                                                \
    hipError_t e__ = (call);                          \
    if (e__ != hipSuccess) return fail_hip(e__, what); \
  } while (0)

// Shared body of the search entry points.
//
// Launch sequence: [pad rows and codebook to 128 columns] -> [auxiliary block of the codebook,
// unless the caller brought one] -> [hint thresholds] -> matrix-core search (best, second best and
// the value of the third per row and K-split) -> k_rescore (merge, direct evaluation of the pair,
// outputs, list of the rows it cannot decide) -> direct search of the listed rows, overwriting
// their outputs.  The number of flagged rows never reaches the host.
int assign_impl(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                const float* c_aux, int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                uint8_t* mask, ic_top2_rec* rec, void* stream, const char* who,
                const int32_t* hint = nullptr, bool use_static = false) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "%s: invalid handle", who);
  if (n < 0 || n >= 0x7fffffffLL || k < 1 || k > 0x7fffffffLL)
    return fail(IC_EINVAL, "%s: need 0 <= n < 2^31 and 1 <= k < 2^31 (n=%lld k=%lld)", who,
                (long long)n, (long long)k);
  if (d < 1 || d > ic::kMaxWideD)
    return fail(IC_EINVAL, "%s: row width %d not in [1,%d]", who, d, ic::kMaxWideD);
  if (n == 0) return IC_OK;
  if (!x || !c) return fail(IC_EINVAL, "%s: x and c must not be NULL", who);
  if (!i1 && !rec) return fail(IC_EINVAL, "%s: no output buffer given", who);
  // rows are searched in 128-column slabs: D columns after zero-padding
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  const bool pad = d != D;
  // the kernels read rows, codewords and the auxiliary block 16 bytes at a time
  if (!pad && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(c)) & 15))
    return fail(IC_EINVAL, "%s: x and c must be 16-byte aligned when d is a multiple of 128", who);
  if (reinterpret_cast<uintptr_t>(c_aux) & 15)
    return fail(IC_EINVAL, "%s: c_aux must be 16-byte aligned", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool mode2 = i2 || d2 || mask || rec;   // the second neighbour is part of the answer

  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");

  const int64_t kpad = (k + ic::kStageK - 1) / ic::kStageK * ic::kStageK;
  ic::AssignPlan plan;
  ic::assign_plan(n, k, ctx->num_cus, &plan, D);
  const size_t records = (size_t)ic::assign_part_records(plan, n);
  size_t off = 0;
  auto take = [&off](size_t bytes) { const size_t o = off; off += align_up(bytes); return o; };
  const size_t o_aux = c_aux ? 0 : take((size_t)ic::aux_floats(k, kpad, D) * 4);
  const size_t o_auxs = c_aux ? 0 : take(ic::codebook_aux_scratch_bytes(k, D));
  const size_t o_part = take(records * sizeof(ic_top2_rec));
  const size_t o_part3 = take(records * 4);
  const size_t o_part3i = mode2 ? take(records * 4) : 0;
  const size_t o_part4 = mode2 ? take(records * 4) : 0;
  const size_t o_list = take((size_t)n * 4);
  const size_t o_count = take(4);
  const size_t o_tickets = take((size_t)ic::kExactGrid * 4);
  const size_t o_expart = take((size_t)ic::kExactGrid * 16 * sizeof(ic_top2_rec));
  const size_t o_xp = pad ? take((size_t)n * D * 4) : 0;
  const size_t o_cp = pad ? take((size_t)k * D * 4) : 0;
  const size_t o_thr = hint ? take((size_t)n * 4) : 0;
  // index-only nearest-codeword calls: the search kernel leaves each row's band constant, and
  // k_rescore settles most rows from the records alone
  const bool index_only = !mode2 && !d1;
  const bool wide = D > ic::kD;
  const size_t o_band = (index_only || wide) ? take((size_t)n * 4) : 0;
  const size_t o_xc = wide ? take((size_t)n * D * 4) : 0;   // centred copy of wide rows
  int rc;
  char* ws;
  if (use_static) {
    // no allocation, no event, nothing but launches on the caller's stream
    if (off > kStaticScratchBytes)
      return fail(IC_ELIMIT, "%s: n=%lld k=%lld d=%d needs %zu bytes of scratch, the static buffer "
                  "holds %zu; use the allocating entry point", who, (long long)n, (long long)k, d,
                  off, kStaticScratchBytes);
    ws = ctx->static_scratch;
  } else {
    rc = ensure_ws(ctx, off);
    if (rc) return rc;
    rc = order_after_previous(ctx, s);
    if (rc) return rc;
    ws = ctx->ws;
  }
  // (the static entry point uses no handle state: nothing to close)
  CallScope scope(ctx, s);
  if (use_static) scope.open = false;

  const ic::PartBufs pb{reinterpret_cast<ic_top2_rec*>(ws + o_part),
                        reinterpret_cast<float*>(ws + o_part3),
                        mode2 ? reinterpret_cast<int32_t*>(ws + o_part3i) : nullptr,
                        mode2 ? reinterpret_cast<float*>(ws + o_part4) : nullptr};
  const float* xs = x;
  const float* cs = c;
  if (pad) {
    float* xp = reinterpret_cast<float*>(ws + o_xp);
    float* cp = reinterpret_cast<float*>(ws + o_cp);
    IC_HIP(ic::launch_pad_rows(x, n, d, D, xp, s), "pad descriptors");
    IC_HIP(ic::launch_pad_rows(c, k, d, D, cp, s), "pad codebook");
    xs = xp;
    cs = cp;
  }
  const float* aux = c_aux;
  if (!aux) {
    float* a = reinterpret_cast<float*>(ws + o_aux);
    IC_HIP(ic::launch_codebook_aux(cs, k, D, a, ws + o_auxs, s), "codebook auxiliary block");
    aux = a;
  }
  const float* mean = aux + ic::aux_mean_offset(kpad, D);
  // a codebook kept in search order inside its auxiliary block: the matrix-core kernel scans that
  // copy, everything else works on the caller's codebook and translates through perm / rank
  const bool ordered = ic::aux_ordered(k, D);
  const float* c_search = ordered ? aux + ic::aux_cperm_offset(kpad) : cs;
  const int32_t* perm =
      ordered ? reinterpret_cast<const int32_t*>(aux + ic::aux_perm_offset(kpad)) : nullptr;
  const int32_t* rank =
      ordered ? reinterpret_cast<const int32_t*>(aux + ic::aux_rank_offset(kpad)) : nullptr;
  float* thr = nullptr;
  if (hint) {
    thr = reinterpret_cast<float*>(ws + o_thr);
    IC_HIP(ic::launch_hint_threshold(xs, n, cs, k, aux, mean, D, hint, rank, thr, s),
           "hint threshold kernel");
  }
  hipEvent_t ev1 = nullptr;
  if (!use_static) {   // timing events are created on demand, which allocates
    rc = profile_open(ctx, s, &ev1);
    if (rc) return rc;
  }
  float* rowband = (index_only || wide) ? reinterpret_cast<float*>(ws + o_band) : nullptr;
  int32_t* list = reinterpret_cast<int32_t*>(ws + o_list);
  uint32_t* count = reinterpret_cast<uint32_t*>(ws + o_count);
  uint32_t* tickets = reinterpret_cast<uint32_t*>(ws + o_tickets);
  IC_HIP(ic::launch_top2_partial(xs, n, c_search, k, aux, mean, D, plan, mode2, thr, pb, rowband,
                                 wide ? reinterpret_cast<float*>(ws + o_xc) : nullptr, count, perm, s),
         "distance/top-2 kernel");
  if (ev1) IC_HIP(hipEventRecord(ev1, s), "hipEventRecord");
  if (!index_only) rowband = nullptr;   // k_rescore's shortcut answers index-only calls
  IC_HIP(ic::launch_rescore(xs, n, cs, aux, mean, D, pb, rowband, plan, mode2, i1, d1, i2, d2,
                            ratio, mask, rec, list, count, tickets, s),
         "rescore kernel");
  IC_HIP(ic::launch_exact_rows(xs, n, cs, k, D, list, count,
                               reinterpret_cast<ic_top2_rec*>(ws + o_expart), tickets, i1, d1, i2,
                               d2, ratio, mask, rec, s),
         "direct search of the flagged rows");
  if (ctx->debug_count) {   // ic_debug_flagged: how many rows took the second look
    IC_HIP(hipMemcpyAsync(ctx->debug_count, count, 4, hipMemcpyDeviceToDevice, s),
           "hipMemcpyAsync");
  }
  return use_static ? IC_OK : scope.finish();
}

}  // namespace

extern "C" {

int ic_version(void) { return 100; }

const char* ic_last_error(void* /*handle*/) { return t_err.c_str(); }

int ic_create(int device, void** handle) {
  if (!handle) return fail(IC_EINVAL, "ic_create: handle pointer is NULL");
  *handle = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess) return fail_hip(e, "hipGetDeviceCount");
  if (device < 0 || device >= count)
    return fail(IC_EINVAL, "ic_create: device %d not in [0,%d)", device, count);
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return fail_hip(e, "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(IC_EINVAL, "ic_create: device %d is %s; this library holds gfx950 code only",
                device, prop.gcnArchName);
  Ctx* c = new (std::nothrow) Ctx();
  if (!c) return fail(IC_ENOMEM, "ic_create: out of host memory");
  c->device = device;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  DeviceGuard guard(device);
  if (!guard.ok) { delete c; return fail_hip(guard.err, "hipSetDevice"); }
  e = hipEventCreateWithFlags(&c->done, hipEventDisableTiming);
  if (e != hipSuccess) { delete c; return fail_hip(e, "hipEventCreate"); }
  // per-device set-up, with `device` current: kernel attributes, and where this GPU keeps its
  // copy of the static scratch array
  e = ic::assign_init();
  if (e == hipSuccess) e = ic::reduce_init();
  if (e != hipSuccess) {
    (void)hipEventDestroy(c->done);
    delete c;
    return fail_hip(e, "hipFuncSetAttribute");
  }
  void* sp = nullptr;
  e = hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !sp) {
    (void)hipEventDestroy(c->done);
    delete c;
    if (e != hipSuccess) return fail_hip(e, "hipGetSymbolAddress(scratch)");
    return fail(IC_EINVAL, "ic_create: hipGetSymbolAddress(scratch) returned a NULL address");
  }
  c->static_scratch = static_cast<char*>(sp);
  *handle = c;
  return IC_OK;
}

int ic_destroy(void* handle) {
  Ctx* c = as_ctx(handle);
  if (!c) return fail(IC_EHANDLE, "ic_destroy: invalid handle");
  {
    std::lock_guard<std::mutex> lock(c->mu);
    DeviceGuard guard(c->device);
    if (guard.ok) {
      (void)hipDeviceSynchronize();
      if (c->ws) (void)hipFree(c->ws);
      if (c->done) (void)hipEventDestroy(c->done);
      for (auto& pr : c->ev_pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
      }
    }
    c->ws = nullptr;
    c->magic = 0;
  }
  delete c;
  return IC_OK;
}

int ic_reserve_workspace(void* handle, size_t bytes) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_reserve_workspace: invalid handle");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  return ensure_ws(ctx, bytes);
}

int ic_workspace_bytes(void* handle, size_t* bytes) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_workspace_bytes: invalid handle");
  if (!bytes) return fail(IC_EINVAL, "ic_workspace_bytes: bytes is NULL");
  std::lock_guard<std::mutex> lock(ctx->mu);
  *bytes = ctx->ws_bytes;
  return IC_OK;
}

int ic_profile_begin(void* handle) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_profile_begin: invalid handle");
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->profiling = true;
  ctx->ev_used = 0;
  return IC_OK;
}

int ic_profile_end(void* handle, double* kernel_ms_sum, int64_t* launches) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_profile_end: invalid handle");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  double sum = 0.0;
  for (size_t i = 0; i < ctx->ev_used; ++i) {
    IC_HIP(hipEventSynchronize(ctx->ev_pool[i].second), "hipEventSynchronize");
    float ms = 0.f;
    IC_HIP(hipEventElapsedTime(&ms, ctx->ev_pool[i].first, ctx->ev_pool[i].second),
           "hipEventElapsedTime");
    sum += ms;
  }
  if (kernel_ms_sum) *kernel_ms_sum = sum;
  if (launches) *launches = (int64_t)ctx->ev_used;
  ctx->profiling = false;
  ctx->ev_used = 0;
  return IC_OK;
}

int ic_row_sqnorm(void* handle, const float* x, int64_t n, int d, float* out, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_row_sqnorm: invalid handle");
  if (n < 0 || d < 1) return fail(IC_EINVAL, "ic_row_sqnorm: bad shape n=%lld d=%d", (long long)n, d);
  if (n == 0) return IC_OK;
  if (!x || !out) return fail(IC_EINVAL, "ic_row_sqnorm: NULL pointer");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  IC_HIP(ic::launch_row_sqnorm(x, n, d, out, static_cast<hipStream_t>(stream)), "row norms");
  return IC_OK;
}

int64_t ic_codebook_aux_floats(int64_t k) { return ic_codebook_aux_floats_d(k, ic::kD); }

int64_t ic_codebook_aux_floats_d(int64_t k, int d) {
  if (k < 1 || d < 1 || d > ic::kMaxWideD) return 0;
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  return ic::aux_floats(k, (k + ic::kStageK - 1) / ic::kStageK * ic::kStageK, D);
}

int ic_codebook_aux(void* handle, const float* c, int64_t k, int d, float* aux, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_codebook_aux: invalid handle");
  if (k < 1 || k > 0x7fffffffLL || d < 1 || d > ic::kMaxWideD)
    return fail(IC_EINVAL, "ic_codebook_aux: bad shape k=%lld d=%d", (long long)k, d);
  if (!c || !aux) return fail(IC_EINVAL, "ic_codebook_aux: NULL pointer");
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  const bool pad = d != D;
  if ((reinterpret_cast<uintptr_t>(aux) & 15) || (!pad && (reinterpret_cast<uintptr_t>(c) & 15)))
    return fail(IC_EINVAL, "ic_codebook_aux: aux (and c when d is a multiple of 128) must be "
                           "16-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  const size_t o_cp = align_up(ic::codebook_aux_scratch_bytes(k, D));
  int rc = ensure_ws(ctx, o_cp + (pad ? align_up((size_t)k * D * 4) : 0));
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  const float* cs = c;
  if (pad) {
    float* cp = reinterpret_cast<float*>(ctx->ws + o_cp);
    IC_HIP(ic::launch_pad_rows(c, k, d, D, cp, s), "pad codebook");
    cs = cp;
  }
  IC_HIP(ic::launch_codebook_aux(cs, k, D, aux, ctx->ws, s), "codebook auxiliary block");
  return scope.finish();
}

int ic_debug_flagged(void* handle, uint32_t* count) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_debug_flagged: invalid handle");
  std::lock_guard<std::mutex> lock(ctx->mu);
  ctx->debug_count = count;
  return IC_OK;
}

int ic_debug_plan(int64_t n, int64_t k, int d, int num_cus, int64_t* out) {
  if (!out || n < 1 || n >= 0x7fffffffLL || k < 1 || k > 0x7fffffffLL || d < 1 ||
      d > ic::kMaxWideD || num_cus < 1)
    return fail(IC_EINVAL, "ic_debug_plan: bad argument");
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  ic::AssignPlan p;
  ic::assign_plan(n, k, num_cus, &p, D);
  const int64_t v[15] = {p.waves, p.stage_k, p.rows_per_wg, p.stages, p.tail_span, p.row_tiles,
                         p.full_tiles, p.tail_quota, p.max_pieces, p.tail_wgs, p.wide_pieces,
                         ic::assign_part_records(p, n), p.resident, p.resident_wgs,
                         p.tail_long < p.tail_wgs ? p.tail_long : p.tail_wgs};
  for (int i = 0; i < 15; ++i) out[i] = v[i];
  return IC_OK;
}

int ic_debug_plan_order(int64_t n, int64_t k, int d, int num_cus, int32_t* order, int cap,
                        int* count) {
  if (!count || n < 1 || n >= 0x7fffffffLL || k < 1 || k > 0x7fffffffLL || d < 1 ||
      d > ic::kMaxWideD || num_cus < 1 || cap < 0 || (cap > 0 && !order))
    return fail(IC_EINVAL, "ic_debug_plan_order: bad argument");
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  ic::AssignPlan p;
  ic::assign_plan(n, k, num_cus, &p, D);
  *count = 0;
  if (!p.tail_ordered) return IC_OK;
  if (p.tail_wgs > cap) return fail(IC_ELIMIT, "ic_debug_plan_order: %lld entries, room for %d",
                                    (long long)p.tail_wgs, cap);
  for (int64_t b = 0; b < p.tail_wgs; ++b) order[b] = p.tail_order[b];
  *count = (int)p.tail_wgs;
  return IC_OK;
}

int ic_assign_top1(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                   const float* c_aux, int32_t* idx, float* dist, void* stream) {
  if (!idx) return fail(IC_EINVAL, "ic_assign_top1: idx is NULL");
  return assign_impl(handle, x, n, c, k, d, c_aux, idx, dist, nullptr, nullptr, 0.f, nullptr,
                     nullptr, stream, "ic_assign_top1");
}

int ic_assign_top1_static(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                          const float* c_aux, int32_t* idx, float* dist, void* stream) {
  if (!idx) return fail(IC_EINVAL, "ic_assign_top1_static: idx is NULL");
  return assign_impl(handle, x, n, c, k, d, c_aux, idx, dist, nullptr, nullptr, 0.f, nullptr,
                     nullptr, stream, "ic_assign_top1_static", nullptr, /*use_static=*/true);
}

int ic_assign_top1_hint(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                        const float* c_aux, const int32_t* hint, int32_t* idx, float* dist,
                        void* stream) {
  if (!idx) return fail(IC_EINVAL, "ic_assign_top1_hint: idx is NULL");
  if (!hint && n > 0) return fail(IC_EINVAL, "ic_assign_top1_hint: hint is NULL");
  return assign_impl(handle, x, n, c, k, d, c_aux, idx, dist, nullptr, nullptr, 0.f, nullptr,
                     nullptr, stream, "ic_assign_top1_hint", hint);
}

int ic_assign_top2(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                   const float* c_aux, int32_t* i1, float* d1, int32_t* i2, float* d2,
                   float ratio, uint8_t* mask, void* stream) {
  if (!i1) return fail(IC_EINVAL, "ic_assign_top2: i1 is NULL");
  return assign_impl(handle, x, n, c, k, d, c_aux, i1, d1, i2, d2, ratio, mask, nullptr, stream,
                     "ic_assign_top2");
}

int ic_assign_top2_exact(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                         int32_t* i1, float* d1, int32_t* i2, float* d2, float ratio,
                         uint8_t* mask, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_assign_top2_exact: invalid handle");
  if (n < 0 || k < 1 || k > 0x7fffffffLL || d < 1 || d > ic::kMaxWideD)
    return fail(IC_EINVAL, "ic_assign_top2_exact: bad shape n=%lld k=%lld d=%d", (long long)n,
                (long long)k, d);
  if (n == 0) return IC_OK;
  if (!x || !c || !i1) return fail(IC_EINVAL, "ic_assign_top2_exact: NULL pointer");
  const int D = (d + ic::kD - 1) / ic::kD * ic::kD;
  if (d == D && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(c)) & 15))
    return fail(IC_EINVAL, "ic_assign_top2_exact: x and c must be 16-byte aligned when d is a "
                           "multiple of 128");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  const float* xs = x;
  const float* cs = c;
  CallScope scope(ctx, s);
  scope.open = false;   // only the padded copies live in the workspace
  if (d != D) {
    const size_t o_cp = align_up((size_t)n * D * 4);
    int rc = ensure_ws(ctx, o_cp + align_up((size_t)k * D * 4));
    if (rc) return rc;
    rc = order_after_previous(ctx, s);
    if (rc) return rc;
    scope.open = true;
    float* xp = reinterpret_cast<float*>(ctx->ws);
    float* cp = reinterpret_cast<float*>(ctx->ws + o_cp);
    IC_HIP(ic::launch_pad_rows(x, n, d, D, xp, s), "pad descriptors");
    IC_HIP(ic::launch_pad_rows(c, k, d, D, cp, s), "pad codebook");
    xs = xp;
    cs = cp;
  }
  IC_HIP(ic::launch_exact_top2(xs, n, cs, k, D, i1, d1, i2, d2, ratio, mask, s),
         "direct search kernel");
  return d != D ? scope.finish() : IC_OK;
}

int ic_assign_top2_rec(void* handle, const float* x, int64_t n, const float* c, int64_t k, int d,
                       const float* c_aux, ic_top2_rec* rec, void* stream) {
  if (!rec) return fail(IC_EINVAL, "ic_assign_top2_rec: rec is NULL");
  return assign_impl(handle, x, n, c, k, d, c_aux, nullptr, nullptr, nullptr, nullptr, 0.f,
                     nullptr, rec, stream, "ic_assign_top2_rec");
}

int ic_top2_merge(void* handle, const ic_top2_rec* gathered, int n_shards, int64_t n,
                  const int64_t* shard_base, int64_t* i1, float* d1, int64_t* i2, float* d2,
                  float ratio, uint8_t* mask, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_top2_merge: invalid handle");
  if (n < 0 || n_shards < 1 || n_shards > ic::kMaxShards)
    return fail(IC_EINVAL, "ic_top2_merge: need n >= 0 and 1 <= n_shards <= %d", ic::kMaxShards);
  if (n == 0) return IC_OK;
  if (!gathered || !i1) return fail(IC_EINVAL, "ic_top2_merge: NULL pointer");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  IC_HIP(ic::launch_top2_merge(gathered, n_shards, n, shard_base, i1, d1, i2, d2, ratio, mask,
                               static_cast<hipStream_t>(stream)),
         "top-2 merge kernel");
  return IC_OK;
}

int ic_hist(void* handle, const int32_t* codes, const int64_t* offsets, int64_t n_img, int64_t k,
            int norm_mode, double power, double* out, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_hist: invalid handle");
  if (n_img < 0 || k < 1) return fail(IC_EINVAL, "ic_hist: bad shape n_img=%lld k=%lld", (long long)n_img, (long long)k);
  if (norm_mode < IC_NORM_NONE || norm_mode > IC_NORM_POWER_L2)
    return fail(IC_EINVAL, "ic_hist: unknown norm_mode %d", norm_mode);
  if (n_img == 0) return IC_OK;
  if (!offsets || !out) return fail(IC_EINVAL, "ic_hist: NULL pointer");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  IC_HIP(ic::launch_hist(codes, offsets, n_img, k, norm_mode, power, nullptr, 0, out,
                         static_cast<hipStream_t>(stream)),
         "histogram kernel");
  return IC_OK;
}

int ic_spatial_hist(void* handle, const int32_t* codes, const int32_t* kp_cells, int64_t n_img,
                    int64_t n_kp, int n_levels, int64_t k, const double* cell_weights,
                    double* out, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_spatial_hist: invalid handle");
  if (n_img < 0 || n_kp < 0 || k < 1 || n_levels < 1 || n_levels > 8)
    return fail(IC_EINVAL, "ic_spatial_hist: bad shape n_img=%lld n_kp=%lld k=%lld levels=%d",
                (long long)n_img, (long long)n_kp, (long long)k, n_levels);
  int64_t n_cells = 0;
  for (int l = 0, c = 1; l < n_levels; ++l, c *= 4) n_cells += c;
  if (n_cells * k > 0x7fffffffLL)
    return fail(IC_EINVAL, "ic_spatial_hist: %lld cells x %lld words exceeds int32 keys",
                (long long)n_cells, (long long)k);
  if (n_img == 0) return IC_OK;
  if (!out || !cell_weights || (n_kp > 0 && (!codes || !kp_cells)))
    return fail(IC_EINVAL, "ic_spatial_hist: NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  size_t off = 0;
  const size_t o_keys = off; off += align_up((size_t)n_img * (size_t)n_kp * n_levels * 4 + 16);
  const size_t o_off = off;  off += align_up((size_t)(n_img + 1) * 8);
  int rc = ensure_ws(ctx, off);
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  IC_HIP(ic::launch_spatial_hist(codes, kp_cells, n_img, n_kp, n_levels, k, n_cells, cell_weights,
                                 reinterpret_cast<int32_t*>(ctx->ws + o_keys),
                                 reinterpret_cast<int64_t*>(ctx->ws + o_off), out, s),
         "spatial histogram kernels");
  return scope.finish();
}

int ic_centroid_partials(void* handle, const float* x, const int32_t* codes, int64_t n, int64_t k,
                         int d, double* sums, int64_t* counts, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_centroid_partials: invalid handle");
  if (n < 0 || k < 1 || d < 1 || d > ic::kMaxWideD)
    return fail(IC_EINVAL, "ic_centroid_partials: bad shape n=%lld k=%lld d=%d", (long long)n, (long long)k, d);
  if (!sums || !counts || (n > 0 && (!x || !codes)))
    return fail(IC_EINVAL, "ic_centroid_partials: NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  if (n == 0) {
    IC_HIP(hipMemsetAsync(sums, 0, (size_t)k * d * 8, s), "hipMemsetAsync");
    IC_HIP(hipMemsetAsync(counts, 0, (size_t)k * 8, s), "hipMemsetAsync");
    return IC_OK;
  }
  ic::CentroidPlan plan;
  ic::centroid_plan(n, k, d, ctx->num_cus, &plan);
  if (plan.use_scan && d > ic::kD)
    return fail(IC_ELIMIT, "ic_centroid_partials: n=%lld x k=%lld is too large for rows wider than "
                "%d columns", (long long)n, (long long)k, ic::kD);
  if (plan.use_scan) {
    int n_rr = 1;
    ic::centroid_plan_scan(n, k, ctx->num_cus, &n_rr);
    size_t off = 0;
    const size_t o_ps = off; if (n_rr > 1) off += align_up((size_t)n_rr * (size_t)k * d * 8);
    const size_t o_pc = off; if (n_rr > 1) off += align_up((size_t)n_rr * (size_t)k * 8);
    int rc = ensure_ws(ctx, off);
    if (rc) return rc;
    rc = order_after_previous(ctx, s);
    if (rc) return rc;
    CallScope scope(ctx, s);
    IC_HIP(ic::launch_centroid_partials_scan(x, codes, n, k, d, n_rr,
                                             reinterpret_cast<double*>(ctx->ws + o_ps),
                                             reinterpret_cast<int64_t*>(ctx->ws + o_pc), sums,
                                             counts, s),
           "centroid sums kernel (scan)");
    return scope.finish();
  }
  int rc = ensure_ws(ctx, plan.bytes);
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  IC_HIP(ic::launch_centroid_partials(x, codes, n, k, d, plan, ctx->ws, sums, counts, s),
         "centroid sums kernels");
  return scope.finish();
}

int ic_centroid_finalize(void* handle, const double* sums, const int64_t* counts,
                         const float* prev_c, int64_t k, int d, float* new_c,
                         uint8_t* has_members, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_centroid_finalize: invalid handle");
  if (k < 1 || d < 1) return fail(IC_EINVAL, "ic_centroid_finalize: bad shape k=%lld d=%d", (long long)k, d);
  if (!sums || !counts || !prev_c || !new_c) return fail(IC_EINVAL, "ic_centroid_finalize: NULL pointer");
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  IC_HIP(ic::launch_centroid_finalize(sums, counts, prev_c, k, d, new_c, has_members,
                                      static_cast<hipStream_t>(stream)),
         "centroid finalize kernel");
  return IC_OK;
}

int ic_pairwise(void* handle, const float* x, int64_t m, const float* c, int64_t n, int d,
                int metric, double* out, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_pairwise: invalid handle");
  if (m < 0 || n < 0 || d < 1 || metric < IC_METRIC_EUCLIDEAN || metric > IC_METRIC_CHISQR)
    return fail(IC_EINVAL, "ic_pairwise: bad argument m=%lld n=%lld d=%d metric=%d", (long long)m,
                (long long)n, d, metric);
  if (m == 0 || n == 0) return IC_OK;
  if (!x || !c || !out) return fail(IC_EINVAL, "ic_pairwise: NULL pointer");
  if (m > (int64_t)65535 * 64)
    return fail(IC_EINVAL, "ic_pairwise: at most %d search rows per call", 65535 * 64);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  IC_HIP(ic::launch_pairdist(x, m, c, n, d, metric, out, static_cast<hipStream_t>(stream)),
         "pairwise distance kernel");
  return IC_OK;
}

int ic_knn_rows(void* handle, const float* x, int64_t m, const float* c, int64_t n, int d,
                int metric, int k, double* out_dist, int32_t* out_idx, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_knn_rows: invalid handle");
  if (m < 0 || n < 1 || d < 1 || metric < IC_METRIC_EUCLIDEAN || metric > IC_METRIC_CHISQR)
    return fail(IC_EINVAL, "ic_knn_rows: bad argument m=%lld n=%lld d=%d metric=%d", (long long)m,
                (long long)n, d, metric);
  if (k < 1 || k > n)
    return fail(IC_EINVAL, "ic_knn_rows: need 1 <= k <= n, got k=%d n=%lld", k, (long long)n);
  if (n > 0x7fffffffLL) return fail(IC_EINVAL, "ic_knn_rows: index rows exceed int32");
  if (m == 0) return IC_OK;
  if (!x || !c || !out_dist || !out_idx) return fail(IC_EINVAL, "ic_knn_rows: NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  // the distance matrix of one block of search rows lives in the workspace (at most ~1 GiB)
  int64_t rows = ((int64_t)1 << 30) / (n * 8);
  if (rows < 1) rows = 1;
  if (rows > m) rows = m;
  if (rows > 65535 * 64) rows = 65535 * 64;
  const int nch = ic::topk_chunks(rows, n, k);
  const size_t o_pd = align_up((size_t)rows * (size_t)n * 8);
  const size_t o_pi = o_pd + align_up((size_t)rows * nch * k * 8);
  int rc = ensure_ws(ctx, o_pi + align_up((size_t)rows * nch * k * 4));
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  double* dist = reinterpret_cast<double*>(ctx->ws);
  double* part_d = reinterpret_cast<double*>(ctx->ws + o_pd);
  int32_t* part_i = reinterpret_cast<int32_t*>(ctx->ws + o_pi);
  for (int64_t r0 = 0; r0 < m; r0 += rows) {
    const int64_t mr = m - r0 < rows ? m - r0 : rows;
    hipEvent_t ev1 = nullptr;
    rc = profile_open(ctx, s, &ev1);
    if (rc) return rc;
    IC_HIP(ic::launch_pairdist(x + r0 * d, mr, c, n, d, metric, dist, s),
           "pairwise distance kernel");
    if (ev1) IC_HIP(hipEventRecord(ev1, s), "hipEventRecord");
    IC_HIP(ic::launch_topk_rows(dist, mr, n, k, nch, part_d, part_i, out_dist + r0 * k,
                                out_idx + r0 * k, s),
           "row top-k kernels");
  }
  return scope.finish();
}

int ic_compact_matches(void* handle, const uint8_t* mask, const int32_t* i1,
                       const int64_t* row_offsets, int64_t n_cand, const double* kp_query,
                       int64_t n_query, const double* kp_rows, int64_t* out_offsets,
                       int32_t* match_q, int32_t* match_im, double* pts, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_compact_matches: invalid handle");
  if (n_cand < 0 || n_cand > 0x7fffffffLL)
    return fail(IC_EINVAL, "ic_compact_matches: bad candidate count %lld", (long long)n_cand);
  if (!out_offsets) return fail(IC_EINVAL, "ic_compact_matches: out_offsets is NULL");
  if (n_cand > 0 && (!mask || !i1 || !row_offsets || !match_q || !match_im))
    return fail(IC_EINVAL, "ic_compact_matches: NULL pointer");
  if (pts && (!kp_query || !kp_rows || n_query < 1))
    return fail(IC_EINVAL, "ic_compact_matches: pts needs kp_query, kp_rows and n_query >= 1");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  if (n_cand == 0) {
    IC_HIP(hipMemsetAsync(out_offsets, 0, 8, s), "hipMemsetAsync");
    return IC_OK;
  }
  int rc = ensure_ws(ctx, align_up((size_t)n_cand * 8));
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  IC_HIP(ic::launch_compact_matches(mask, i1, row_offsets, n_cand, kp_query, n_query, kp_rows,
                                    reinterpret_cast<int64_t*>(ctx->ws), out_offsets, match_q,
                                    match_im, pts, s),
         "match compaction kernels");
  return scope.finish();
}

int ic_ransac_homography(void* handle, const double* pts, const int64_t* offsets, int64_t n_cand,
                         int n_iter, const int32_t* samples, uint64_t seed, double th,
                         int min_inliers, int flags, double* H, int32_t* n_inliers,
                         double* best_err, int32_t* best_iter, uint8_t* inlier_mask, void* stream) {
  Ctx* ctx = as_ctx(handle);
  if (!ctx) return fail(IC_EHANDLE, "ic_ransac_homography: invalid handle");
  if (n_cand < 0 || n_cand > 0x7fffffffLL || n_iter < 0 || n_iter > 65535 * 16)
    return fail(IC_EINVAL, "ic_ransac_homography: bad size n_cand=%lld n_iter=%d",
                (long long)n_cand, n_iter);
  if (flags & ~(IC_RANSAC_MEAN_ERR | IC_RANSAC_CV_RULES))
    return fail(IC_EINVAL, "ic_ransac_homography: unknown flags %d", flags);
  if (n_cand == 0) return IC_OK;
  if (!pts || !offsets || !H || !n_inliers || !best_err || !inlier_mask)
    return fail(IC_EINVAL, "ic_ransac_homography: NULL pointer");
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(ctx->mu);
  DeviceGuard guard(ctx->device);
  if (!guard.ok) return fail_hip(guard.err, "hipSetDevice");
  int rc = ensure_ws(ctx, align_up((size_t)n_cand * (size_t)(n_iter > 0 ? n_iter : 1) * 4));
  if (rc) return rc;
  rc = order_after_previous(ctx, s);
  if (rc) return rc;
  CallScope scope(ctx, s);
  IC_HIP(ic::launch_ransac(pts, offsets, n_cand, n_iter, samples, seed, th, min_inliers,
                           flags, reinterpret_cast<int32_t*>(ctx->ws), H, n_inliers,
                           best_err, best_iter, inlier_mask, s),
         "homography RANSAC kernels");
  return scope.finish();
}

}  // extern "C"
