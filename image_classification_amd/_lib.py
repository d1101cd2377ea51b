"""ctypes binding of libic_hip.so (C ABI in include/ic_hip.h).

Nothing here touches HIP at import time: the shared library is loaded, and a handle created, on
the first call in a process.  Handles are cached per (pid, device) because the reference forks
its workers after the models are loaded (model/mlflow/parallel_flow.py:50-53) and a HIP context
does not survive a fork; a child simply creates its own handle on first use.

There is no CPU fallback: if the library is missing or there is no gfx950 device, calls raise.
"""
import ctypes
import os
import threading

_LIB_NAME = "libic_hip.so"
_lib = None
_lib_lock = threading.Lock()
_handles = {}  # (pid, device) -> ctypes.c_void_p

IC_NORM_NONE, IC_NORM_L1, IC_NORM_L2, IC_NORM_POWER_L2 = 0, 1, 2, 3
NORM_MODES = {"none": IC_NORM_NONE, "l1": IC_NORM_L1, "l2": IC_NORM_L2,
              "power_l2": IC_NORM_POWER_L2}
# metric names of ic_pairwise / ic_knn_rows (scikit-learn's aliases map to the same number)
METRICS = {"euclidean": 0, "l2": 0, "sqeuclidean": 1, "manhattan": 2, "cityblock": 2, "l1": 2,
           "intersection": 3, "chisqr": 4}

_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int
_c_f = ctypes.c_float
_c_d = ctypes.c_double
_p = ctypes.c_void_p

# name -> argtypes (restype is always int unless listed in _RESTYPES)
_SIGNATURES = {
    "ic_version": [],
    "ic_create": [_c_int, ctypes.POINTER(_p)],
    "ic_destroy": [_p],
    "ic_last_error": [_p],
    "ic_reserve_workspace": [_p, ctypes.c_size_t],
    "ic_workspace_bytes": [_p, ctypes.POINTER(ctypes.c_size_t)],
    "ic_profile_begin": [_p],
    "ic_profile_end": [_p, ctypes.POINTER(_c_d), ctypes.POINTER(_c_i64)],
    "ic_row_sqnorm": [_p, _p, _c_i64, _c_int, _p, _p],
    "ic_codebook_aux_floats": [_c_i64],
    "ic_codebook_aux_floats_d": [_c_i64, _c_int],
    "ic_codebook_aux": [_p, _p, _c_i64, _c_int, _p, _p],
    "ic_debug_flagged": [_p, _p],
    "ic_debug_plan": [_c_i64, _c_i64, _c_int, _c_int, ctypes.POINTER(_c_i64)],
    "ic_debug_plan_order": [_c_i64, _c_i64, _c_int, _c_int, ctypes.POINTER(ctypes.c_int32), _c_int,
                            ctypes.POINTER(_c_int)],
    "ic_assign_top1": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p, _p],
    "ic_assign_top1_static": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p, _p],
    "ic_assign_top1_hint": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p, _p, _p],
    "ic_assign_top2": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p, _p, _p, _c_f, _p, _p],
    "ic_assign_top2_exact": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p, _p, _c_f, _p, _p],
    "ic_assign_top2_rec": [_p, _p, _c_i64, _p, _c_i64, _c_int, _p, _p, _p],
    "ic_top2_merge": [_p, _p, _c_int, _c_i64, ctypes.POINTER(_c_i64), _p, _p, _p, _p, _c_f, _p, _p],
    "ic_hist": [_p, _p, _p, _c_i64, _c_i64, _c_int, _c_d, _p, _p],
    "ic_spatial_hist": [_p, _p, _p, _c_i64, _c_i64, _c_int, _c_i64, _p, _p, _p],
    "ic_centroid_partials": [_p, _p, _p, _c_i64, _c_i64, _c_int, _p, _p, _p],
    "ic_centroid_finalize": [_p, _p, _p, _p, _c_i64, _c_int, _p, _p, _p],
    "ic_pairwise": [_p, _p, _c_i64, _p, _c_i64, _c_int, _c_int, _p, _p],
    "ic_knn_rows": [_p, _p, _c_i64, _p, _c_i64, _c_int, _c_int, _c_int, _p, _p, _p],
    "ic_compact_matches": [_p, _p, _p, _p, _c_i64, _p, _c_i64, _p, _p, _p, _p, _p, _p],
    "ic_ransac_homography": [_p, _p, _p, _c_i64, _c_int, _p, ctypes.c_uint64, _c_d, _c_int, _c_int,
                             _p, _p, _p, _p, _p, _p],
    "ic_host_gather_f32": [_p, _p, _p, _p, _c_i64, _c_int],
}
_RESTYPES = {"ic_last_error": ctypes.c_char_p, "ic_codebook_aux_floats": ctypes.c_int64,
             "ic_codebook_aux_floats_d": ctypes.c_int64}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def library_path():
    # IC_AMD_LIBRARY: another build of the same library (A/B timing of kernel variants)
    return os.environ.get("IC_AMD_LIBRARY") or \
        os.path.join(os.path.dirname(os.path.abspath(__file__)), _LIB_NAME)


def load():
    """Load libic_hip.so and declare every prototype.  Raises if the library was not built."""
    global _lib
    with _lib_lock:
        if _lib is None:
            path = library_path()
            if not os.path.exists(path):
                raise RuntimeError(
                    "%s not found: build it with `make -C image_classification_amd/csrc` "
                    "(or __graft_entry__.build()); there is no CPU fallback" % path)
            lib = ctypes.CDLL(path)
            for name, argtypes in _SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
                fn.argtypes = argtypes
                fn.restype = _RESTYPES.get(name, ctypes.c_int)
            _lib = lib
    return _lib


def last_error():
    msg = load().ic_last_error(None)
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, last_error()))


def handle(device):
    """Per-process, per-device library handle (created lazily, never shared across fork)."""
    key = (os.getpid(), int(device))
    h = _handles.get(key)
    if h is None:
        with _lib_lock:
            h = _handles.get(key)
        if h is None:
            lib = load()
            out = _p()
            check(lib.ic_create(int(device), ctypes.byref(out)), "ic_create(device=%d)" % device)
            with _lib_lock:
                _handles[key] = out
            h = out
    return h


def destroy_handles():
    """Destroy the handles this process created (tests / orderly shutdown)."""
    pid = os.getpid()
    with _lib_lock:
        mine = [k for k in _handles if k[0] == pid]
        hs = [_handles.pop(k) for k in mine]
    for h in hs:
        load().ic_destroy(h)
