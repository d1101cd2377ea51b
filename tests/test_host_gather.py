"""The native host-side gather of the upload pipeline (ic_host_gather_f32): equal to
numpy.concatenate(...).astype(float32), which is what the reference does before every search
(model/features.py:1164).  Runs without a GPU."""
import numpy as np
import pytest

from image_classification_amd import _lib

IC_EINVAL = 10001   # include/ic_hip.h


def _call(dst, parts, threads, kind=True):
    lib = _lib.load()
    src = np.array([p.__array_interface__["data"][0] for p in parts], dtype=np.uintp)
    elems = np.array([p.size for p in parts], dtype=np.int64)
    kinds = np.array([p.dtype == np.float64 for p in parts], dtype=np.int32)
    return lib.ic_host_gather_f32(dst.ctypes.data, src.ctypes.data, elems.ctypes.data,
                                  kinds.ctypes.data if kind else None, len(parts), threads)


@pytest.mark.parametrize("threads", [1, 3, 8, 64])
def test_gather_equals_concatenate(threads):
    rng = np.random.default_rng(threads)
    rows = list(rng.integers(0, 2500, 150)) + [0, 1, 0]
    parts = [rng.standard_normal((int(r), 128)).astype(np.float64 if i % 3 == 0 else np.float32)
             for i, r in enumerate(rows)]
    want = np.concatenate(parts).astype(np.float32)
    dst = np.full(want.shape, -7.0, dtype=np.float32)
    assert _call(dst, parts, threads) == 0
    assert np.array_equal(dst, want)


def test_all_float32_needs_no_kind_array_and_small_inputs_work():
    rng = np.random.default_rng(0)
    parts = [rng.random((3, 5), dtype=np.float32), rng.random((1, 5), dtype=np.float32)]
    dst = np.zeros((4, 5), dtype=np.float32)
    assert _call(dst, parts, 8, kind=False) == 0
    assert np.array_equal(dst, np.concatenate(parts))
    assert _call(dst, [], 8) == 0


def test_bad_arguments_are_refused():
    dst = np.zeros(4, dtype=np.float32)
    part = [np.ones(4, dtype=np.float32)]
    assert _call(dst, part, 0) == IC_EINVAL
    assert _call(dst, part, 1000) == IC_EINVAL
    lib = _lib.load()
    elems = np.array([-1], dtype=np.int64)
    src = np.array([part[0].ctypes.data], dtype=np.uintp)
    assert lib.ic_host_gather_f32(dst.ctypes.data, src.ctypes.data, elems.ctypes.data, None, 1,
                                  2) == IC_EINVAL
    assert lib.ic_host_gather_f32(None, src.ctypes.data, elems.ctypes.data, None, 1,
                                  2) == IC_EINVAL
