"""ic_assign_top1_static -- the entry point that allocates nothing -- against ic_assign_top1.

Runs last (file name) so that a failure here cannot hide other tests; it must also pass alone in a
fresh process, where its first call is the first use of the library on the device."""
import ctypes

import numpy as np
import pytest
import torch

from image_classification_amd import _lib, ops
from image_classification_amd.synthetic import sift_like

pytestmark = pytest.mark.gpu


def _both(x_np, c_np):
    dev = torch.device("cuda", 0)
    x = torch.from_numpy(x_np).to(dev)
    c = torch.from_numpy(c_np).to(dev)
    n = x.shape[0]
    # outputs pre-filled with garbage: every element must be written by the call
    idx_s = torch.full((n,), -7, dtype=torch.int32, device=dev)
    dist_s = torch.full((n,), -1.0, dtype=torch.float32, device=dev)
    _lib.handle(0)   # set-up (loads the code object, whose static scratch is taken from the device)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(dev)[0]
    ops.assign_top1_static(x, c, None, (idx_s, dist_s))
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(dev)[0]
    idx_a, dist_a = ops.assign_top1(x, c)
    torch.cuda.synchronize()
    return (idx_s.cpu().numpy(), dist_s.cpu().numpy(), idx_a.cpu().numpy(), dist_a.cpu().numpy(),
            free0 - free1)


def test_static_entry_equals_allocating_entry_on_two_calls_in_a_row():
    # first call: 70 000 rows x 3000 codewords (whole rounds of row tiles plus a K-split tail)
    i_s, d_s, i_a, d_a, grew = _both(sift_like(70_000, seed=11), sift_like(3000, seed=12))
    assert np.array_equal(i_s, i_a)
    assert np.array_equal(d_s, d_a)                 # bit-equal distances, not merely close
    assert grew == 0                                # no device memory was taken by the call
    # second call, smaller in both dimensions: stale scratch contents must not leak into it
    i_s, d_s, i_a, d_a, grew = _both(sift_like(9001, seed=13), sift_like(701, seed=14))
    assert np.array_equal(i_s, i_a)
    assert np.array_equal(d_s, d_a)
    assert grew == 0
    # and against the float64 oracle, to the rule the other parity tests use
    import oracle
    rd, ri = oracle.topk(sift_like(9001, seed=13), sift_like(701, seed=14), 2)
    must, bad, agree = oracle.check_indices(i_s, ri, rd, 0)
    assert bad == 0 and agree > 0.9999
    assert oracle.check_close(d_s, rd[:, 0]) <= 1e-4


def test_static_entry_with_the_codebook_resident_in_lds():
    # many rows against a small codebook take the kernel that keeps the codebook in LDS
    i_s, d_s, i_a, d_a, grew = _both(sift_like(150_000, seed=17), sift_like(200, seed=18))
    assert np.array_equal(i_s, i_a) and np.array_equal(d_s, d_a) and grew == 0


def test_static_entry_narrow_rows_use_padded_copies_in_the_scratch():
    x = sift_like(5000, d=64, seed=15)
    c = sift_like(300, d=64, seed=16)
    i_s, d_s, i_a, d_a, grew = _both(x, c)
    assert np.array_equal(i_s, i_a) and np.array_equal(d_s, d_a) and grew == 0


def test_static_entry_rejects_what_does_not_fit():
    dev = torch.device("cuda", 0)
    x = torch.zeros((64, 128), dtype=torch.float32, device=dev)
    c = torch.zeros((8, 128), dtype=torch.float32, device=dev)
    idx = torch.full((64,), -7, dtype=torch.int32, device=dev)
    lib = _lib.load()
    # 16 M rows need 256 MB of per-row records: refused before anything is launched, so the short
    # buffers above are never touched
    rc = lib.ic_assign_top1_static(_lib.handle(0), ctypes.c_void_p(x.data_ptr()), 1 << 24,
                                   ctypes.c_void_p(c.data_ptr()), 8, 128, None,
                                   ctypes.c_void_p(idx.data_ptr()), None, None)
    assert rc == 10004                              # IC_ELIMIT
    assert "static buffer" in _lib.last_error()
    torch.cuda.synchronize()
    assert int((idx != -7).sum()) == 0
