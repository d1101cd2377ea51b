"""The C ABI without a GPU: the library loads, exports every symbol include/ic_hip.h declares,
the ctypes prototypes cover exactly those symbols, and nothing silently falls back to the CPU."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from image_classification_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "ic_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ic_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ic_create", "ic_destroy", "ic_last_error", "ic_row_sqnorm", "ic_assign_top1",
                 "ic_assign_top2", "ic_assign_top2_rec", "ic_top2_merge", "ic_hist",
                 "ic_centroid_partials", "ic_centroid_finalize"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.library_path())
    for name in declared_symbols():
        assert hasattr(lib, name), "libic_hip.so does not export %s" % name


def test_ctypes_prototypes_match_header():
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_symbols()
    lib = _lib.load()
    assert lib.ic_version() >= 100
    assert lib.ic_last_error.restype is ctypes.c_char_p


def test_header_cites_the_reference_for_each_search_entry_point():
    text = open(HEADER).read()
    for cite in ("model/features.py:1166", "vision/utils/scene_functions.py:391",
                 "model/classifiers.py:195-197", "model/robust_matching.py:81-84",
                 "model/features.py:1174-1181", "vision/facerec/face_indexing.py:158"):
        assert cite in text


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the behaviour without a GPU")
def test_no_gpu_means_loud_failure_not_cpu_fallback():
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.ic_create(0, ctypes.byref(h))
    assert rc != 0 and not h.value
    assert _lib.last_error() != ""
    assert lib.ic_destroy(None) == 10003          # IC_EHANDLE
    from image_classification_amd import vq
    with pytest.raises(RuntimeError):
        vq(np.zeros((4, 8), dtype=np.float32), np.zeros((2, 8)))


def test_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "image_classification_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M), f
