"""Seeded sweeps over ragged shapes: every kernel of the path against the oracle."""
import numpy as np
import pytest
import torch

import oracle
from image_classification_amd import ops

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_search_hist_centroid_over_random_shapes():
    rng = np.random.default_rng(20240)
    for trial in range(40):
        n = int(rng.integers(1, 3000))
        k = int(rng.integers(1, 700))
        if trial % 5 == 4:
            k = int(rng.integers(2048, 4200))     # the workgroup geometry of large codebooks
        d = int(rng.choice([1, 2, 3, 7, 31, 32, 33, 64, 100, 127, 128]))
        scale = float(rng.choice([0.01, 1.0, 300.0]))
        x = (rng.standard_normal((n, d)) * scale).astype(np.float32)
        c = (rng.standard_normal((k, d)) * scale).astype(np.float32)
        if k > 4 and trial % 3 == 0:
            c[k // 2] = c[0]                      # a duplicated codeword: ties
            x[: min(n, 5)] = c[: min(n, 5)]       # exact hits
        tag = "trial %d: n=%d k=%d d=%d" % (trial, n, k, d)
        i1, d1, i2, d2, mask = (t.cpu().numpy() for t in ops.assign_top2(_dev(x), _dev(c), 0.8))
        rd, ri = oracle.topk(x, c, min(3, k))
        # the strict rule, also for low-dimensional rows that sit far closer to their neighbours
        # than their own norm: such rows are decided by direct differences
        must, bad, _ = oracle.check_indices(i1, ri, rd, 0) if k >= 2 else (0, int((i1 != 0).sum()), 1)
        assert bad == 0, tag
        ok = i1 == ri[:, 0]
        assert oracle.check_close(d1[ok], rd[ok, 0], abs_floor=1e-6 * scale) <= 1e-4, tag
        if k >= 3:
            must, bad, _ = oracle.check_indices(i2, ri, rd, 1)
            assert bad == 0, tag
        elif k == 2:
            assert set(map(tuple, np.sort(np.stack([i1, i2], 1), 1))) == {(0, 1)}, tag
        else:
            assert (i2 == -1).all() and np.isinf(d2).all(), tag
        # histogram of the codes over ragged images (some empty), bit-exact
        cuts = np.sort(rng.integers(0, n + 1, size=int(rng.integers(1, 9))))
        offsets = np.concatenate([[0], cuts, [n]]).astype(np.int64)
        got = ops.hist(_dev(i1.astype(np.int32)), _dev(offsets), k, "l1").cpu().numpy()
        parts = [i1[offsets[j]:offsets[j + 1]] for j in range(len(offsets) - 1)]
        assert np.array_equal(got, oracle.build_histogram(parts, k, "l1")), tag
        # centroid sums / counts of that assignment
        sums, counts = ops.centroid_partials(_dev(x), _dev(i1.astype(np.int32)), k)
        _, _, rs, rc = oracle.update_cluster_means(x, i1.astype(np.int64), k)
        assert np.array_equal(counts.cpu().numpy(), rc), tag
        assert oracle.check_close(sums.cpu().numpy(), rs, rel=1e-12, abs_floor=1e-6 * scale) <= 1e-12, tag


def test_knn_rows_over_random_shapes():
    rng = np.random.default_rng(777)
    for trial in range(25):
        m = int(rng.integers(1, 90))
        n = int(rng.integers(1, 400))
        d = int(rng.integers(1, 600))
        k = int(rng.integers(1, min(n, 9) + 1))
        metric = str(rng.choice(["euclidean", "manhattan", "intersection", "sqeuclidean"]))
        x = rng.random((m, d)).astype(np.float32)
        c = rng.random((n, d)).astype(np.float32)
        dist, idx = ops.knn_rows(_dev(x), _dev(c), k, metric)
        rd, ri = oracle.kneighbors_metric(c, x, k, metric)
        same = idx.cpu().numpy() == ri
        tag = "trial %d: m=%d n=%d d=%d k=%d %s" % (trial, m, n, d, k, metric)
        assert same.mean() > 0.99, tag
        assert oracle.check_close(dist.cpu().numpy()[same], rd[same], abs_floor=1e-9) <= 1e-4, tag


def test_search_against_direct_search_over_random_shapes(monkeypatch):
    # tools/soak.py: both workgroup geometries, K-splits, padded widths, ties, hints, shard merge,
    # each case checked on the device against the search by direct differences
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "soak.py")
    spec = importlib.util.spec_from_file_location("ic_soak", path)
    soak = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(soak)
    monkeypatch.setattr("sys.argv", ["soak.py", "150", "11"])
    soak.main()
