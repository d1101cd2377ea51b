"""GPU parity of the histogram and centroid-sum kernels against the oracle, through the C ABI,
plus size-independent properties at BASELINE config-2 size."""
import numpy as np
import pytest
import torch

import oracle
from image_classification_amd import ops
from image_classification_amd.synthetic import image_offsets, sift_like, sift_like_device

pytestmark = pytest.mark.gpu


def _hist(codes, offsets, k, norm="l1", power=0.5):
    h = ops.hist(torch.from_numpy(np.asarray(codes, dtype=np.int32)).cuda(),
                 torch.from_numpy(np.asarray(offsets, dtype=np.int64)).cuda(), k, norm, power)
    torch.cuda.synchronize()
    return h.cpu().numpy()


def _split(codes, offsets):
    return [codes[offsets[i]:offsets[i + 1]] for i in range(len(offsets) - 1)]


@pytest.mark.parametrize("k", [1, 50, 1000, 10000])
def test_hist_l1_bit_exact_ragged(k):
    rng = np.random.default_rng(k)
    offsets = image_offsets(40, 300, ragged=True, seed=k)
    codes = rng.integers(0, k, size=offsets[-1]).astype(np.int32)
    got = _hist(codes, offsets, k, "l1")
    ref = oracle.build_histogram(_split(codes, offsets), k, "l1")
    assert got.dtype == np.float64 and np.array_equal(got, ref)
    assert np.all(got[1] == 0) and np.all(got[38] == 0)   # the two empty images


@pytest.mark.parametrize("norm", ["none", "l2", "power_l2"])
def test_hist_extension_norms(norm):
    rng = np.random.default_rng(3)
    k = 777
    offsets = image_offsets(25, 500, ragged=True, seed=9)
    codes = rng.integers(0, k, size=offsets[-1]).astype(np.int32)
    got = _hist(codes, offsets, k, norm)
    ref = oracle.build_histogram(_split(codes, offsets), k, norm)
    assert oracle.check_close(got, ref, rel=1e-12, abs_floor=1e-300) <= 1e-12
    got3 = _hist(codes, offsets, k, "power_l2", power=0.3) if norm == "power_l2" else None
    if got3 is not None:
        ref3 = oracle.build_histogram(_split(codes, offsets), k, "power_l2", 0.3)
        assert oracle.check_close(got3, ref3, rel=1e-10, abs_floor=1e-300) <= 1e-10


def test_hist_image_larger_than_one_sort_chunk_and_skewed_codes():
    rng = np.random.default_rng(4)
    k = 300
    sizes = [10000, 4096, 4097, 1, 0, 8191]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    codes = rng.integers(0, k, size=offsets[-1]).astype(np.int32)
    codes[:9000] = 17                       # one run longer than two chunks
    codes[offsets[1]:offsets[2]] = 299      # a whole chunk of one code
    got = _hist(codes, offsets, k, "none")
    ref = oracle.build_histogram(_split(codes, offsets), k, "none")
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("words", [1, 8, 70])
def test_hist_images_whose_codes_fall_on_a_few_words(words):
    # blank frames: every descriptor of an image maps to the same handful of words, all with the
    # same two low bits, so one wave counts them and most lanes of a piece hold equal codes
    rng = np.random.default_rng(words)
    k = 10000
    offsets = image_offsets(9, 2000, ragged=True, seed=5)
    vocab_subset = (rng.choice(k // 4, size=words, replace=False) * 4 + 2).astype(np.int32)
    codes = vocab_subset[rng.integers(0, words, size=offsets[-1])]
    for norm in ("none", "l1"):
        got = _hist(codes, offsets, k, norm)
        ref = oracle.build_histogram(_split(codes, offsets), k, norm)
        assert np.array_equal(got, ref)


@pytest.mark.parametrize("n_img,k", [(1, 10000), (3, 1025), (63, 4001), (2, 40000), (1, 70001)])
def test_hist_few_long_rows_are_cut_into_bin_ranges(n_img, k):
    # one image per call, as the reference's flow issues them: raw and L1 rows are counted by
    # several workgroups, one bin range each (an L1 row's sum is its number of codes, which every
    # range knows); codes outside the vocabulary count nowhere, not even in the sum
    rng = np.random.default_rng(n_img * k)
    offsets = image_offsets(n_img, 1500, ragged=n_img > 2, seed=k)
    codes = rng.integers(0, k, size=offsets[-1]).astype(np.int32)
    codes[::97] = -1
    codes[5::101] = k + 3
    for norm in ("none", "l1", "l2"):
        got = _hist(codes, offsets, k, norm)
        ref = oracle.build_histogram(_split(codes, offsets), k, norm)
        if norm == "l2":
            assert oracle.check_close(got, ref, rel=1e-12, abs_floor=1e-300) <= 1e-12
        else:
            assert np.array_equal(got, ref)


def test_hist_ignores_codes_outside_the_vocabulary():
    codes = np.array([0, 5, -1, 7, 99, 5, 2 ** 31 - 1], dtype=np.int32)
    got = _hist(codes, [0, 7], 8, "l1")
    ref = oracle.build_histogram([codes], 8, "l1")   # np.histogram drops them too
    assert np.array_equal(got, ref)


def test_hist_vocabulary_wider_than_one_workgroup_range():
    rng = np.random.default_rng(6)
    k = 65536
    offsets = image_offsets(12, 2000, ragged=True, seed=2)
    codes = rng.integers(0, k, size=offsets[-1]).astype(np.int32)
    for norm in ("l1", "l2", "power_l2", "none"):
        got = _hist(codes, offsets, k, norm)
        ref = oracle.build_histogram(_split(codes, offsets), k, norm)
        if norm in ("l1", "none"):
            assert np.array_equal(got, ref)
        else:
            assert oracle.check_close(got, ref, rel=1e-12, abs_floor=1e-300) <= 1e-12


def test_centroid_partials_skewed_codes_span_many_sort_chunks():
    # almost every row in one cluster: its run crosses sort-chunk and row-range boundaries
    rng = np.random.default_rng(99)
    n, k, d = 50000, 100, 128
    x = (rng.standard_normal((n, d)) * 50).astype(np.float32)
    codes = np.full(n, 37, dtype=np.int32)
    codes[rng.choice(n, size=300, replace=False)] = rng.integers(0, k, size=300)
    sums, counts = _centroid(x, codes, k)
    _, _, rsums, rcounts = oracle.update_cluster_means(x, codes.astype(np.int64), k)
    assert np.array_equal(counts, rcounts)
    assert oracle.check_close(sums, rsums, rel=1e-12, abs_floor=1e-3) <= 1e-12


def _centroid(x, codes, k):
    s, c = ops.centroid_partials(torch.from_numpy(x).cuda(), torch.from_numpy(codes).cuda(), k)
    torch.cuda.synchronize()
    return s.cpu().numpy(), c.cpu().numpy()


@pytest.mark.parametrize("n,k,d", [(5000, 32, 128), (20000, 1000, 128), (3000, 7, 64),
                                   (100000, 65536, 128), (777, 130, 36), (64, 64, 128),
                                   (60000, 40, 128),   # few buckets: row ranges + reduce pass
                                   (3001, 50, 37),     # odd width: scalar-load variant
                                   (1, 1, 128)])
def test_centroid_partials_vs_oracle(n, k, d):
    rng = np.random.default_rng(n + k)
    x = (rng.standard_normal((n, d)) * 50).astype(np.float32)
    codes = rng.integers(0, k, size=n).astype(np.int32)
    if k > 3:
        codes[codes == 2] = 3                     # cluster 2 is empty
    sums, counts = _centroid(x, codes, k)
    _, _, rsums, rcounts = oracle.update_cluster_means(x, codes.astype(np.int64), k)
    assert np.array_equal(counts, rcounts)
    assert oracle.check_close(sums, rsums, rel=1e-12, abs_floor=1e-3) <= 1e-12
    # deterministic: a second launch gives the same bits
    sums2, counts2 = _centroid(x, codes, k)
    assert np.array_equal(sums, sums2) and np.array_equal(counts, counts2)
    prev = (rng.standard_normal((k, d))).astype(np.float32)
    new_c, has = ops.centroid_finalize(torch.from_numpy(sums).cuda(), torch.from_numpy(counts).cuda(),
                                       torch.from_numpy(prev).cuda())
    new_c, has = new_c.cpu().numpy(), has.cpu().numpy().astype(bool)
    assert np.array_equal(has, rcounts > 0)
    ref_c = prev.copy()
    ref_c[has] = (rsums[has] / rcounts[has][:, None]).astype(np.float32)
    assert np.array_equal(new_c[~has], prev[~has])
    assert oracle.check_close(new_c, ref_c, rel=1e-6, abs_floor=1e-3) <= 1e-6


@pytest.mark.parametrize("k,levels,n_img", [(500, 3, 9), (10000, 3, 4), (64, 4, 3), (7, 1, 2)])
def test_spatial_pyramid_vs_reference_loop(k, levels, n_img):
    from image_classification_amd.scene_functions import (spatial_cell_weights,
                                                          spatial_keypoint_cells)
    rng = np.random.default_rng(k + levels)
    cols = np.arange(7, 800, 15)
    rows = np.arange(7, 400, 15)
    kps = np.asarray([(c, r) for c in cols for r in rows], dtype=np.float64)
    codes = rng.integers(0, k, size=(n_img, len(kps))).astype(np.int32)
    cells = spatial_keypoint_cells(kps, levels, 800, 400)
    w = spatial_cell_weights(levels)
    got = ops.spatial_hist(torch.from_numpy(codes).cuda(), torch.from_numpy(cells).cuda(),
                           torch.from_numpy(w).cuda(), k)
    torch.cuda.synchronize()
    ref = oracle.compute_spatial_features(codes, kps, k, levels, 800, 400)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_config2_full_size_properties():
    """1M x 10k x 128 on one GPU: properties that need no CPU oracle at this size."""
    dev = torch.device("cuda")
    n, k, n_img = 1_000_000, 10_000, 1000
    cb = sift_like_device(k, n_modes=2048, seed=4, device=dev)
    x = sift_like_device(n, n_modes=2048, seed=3, device=dev)
    offsets = torch.from_numpy(image_offsets(n_img, n // n_img)).to(dev)
    idx, dist = ops.assign_top1(x, cb)
    # (1) histogram rows are the per-image bincount of the codes, L1 rows sum to 1
    h = ops.hist(idx, offsets, k, "none")
    ref = torch.stack([torch.bincount(idx[i * 1000:(i + 1) * 1000].long(), minlength=k)
                       for i in range(0, n_img, 97)]).double()
    assert torch.equal(h[::97], ref)
    h1 = ops.hist(idx, offsets, k, "l1")
    assert torch.allclose(h1.sum(1), torch.ones(n_img, dtype=torch.float64, device=dev), rtol=1e-12)
    # (2) idempotence: every codeword is its own nearest codeword at distance 0
    self_idx, self_dist = ops.assign_top1(cb, cb)
    assert torch.equal(self_idx.long(), torch.arange(k, device=dev))
    assert torch.all(self_dist == 0)
    # (3) scaling both sides by 2 keeps the indices and doubles the distances (exact in binary fp)
    idx2, dist2 = ops.assign_top1(x[:200000] * 2, cb * 2)
    assert torch.equal(idx2, idx[:200000]) and torch.equal(dist2, dist[:200000] * 2)
    # (4) the reported distance is the distance to the reported codeword, and no sampled codeword
    #     is closer
    sel = torch.arange(0, n, 1013, device=dev)
    direct = (x[sel].double() - cb[idx[sel].long()].double()).norm(dim=1)
    assert torch.allclose(dist[sel].double(), direct, rtol=1e-5)
    probe = torch.cdist(x[sel].double(), cb[::7].double())
    assert torch.all(probe.min(1).values >= direct * (1 - 1e-6))
    # (5) a row-sharded run equals the unsharded run (the multi-GPU decomposition)
    parts = [ops.assign_top1(x[lo:lo + 250000], cb)[0] for lo in range(0, n, 250000)]
    assert torch.equal(torch.cat(parts), idx)
    # (6) checksum of the centroid sums: sum over clusters == sum over rows, counts == bincount
    sums, counts = ops.centroid_partials(x, idx, k)
    assert torch.equal(counts, torch.bincount(idx.long(), minlength=k))
    assert torch.allclose(sums.sum(0), x.double().sum(0), rtol=1e-11)
    # (7) agreement with the float64 oracle on a bounded sample
    xs = x[:20000].cpu().numpy()
    rd, ri = oracle.topk(xs, cb.cpu().numpy(), 2)
    must, bad, agree = oracle.check_indices(idx[:20000].cpu().numpy(), ri, rd, 0)
    assert bad == 0 and agree >= 0.9999
    # (9) ALL rows against the direct search on the device (no norm expansion, four times the
    #     work): the same codeword and bit-identical distances on every one of the 1M rows
    e1, ed1, e2, ed2, _ = ops.assign_top2_exact(x, cb)
    assert int((idx != e1).sum()) == 0
    assert torch.equal(dist.view(torch.int32), ed1.view(torch.int32))
    # (8) ragged images (sizes ~ Poisson(1000), two empty ones): same codes, other boundaries
    rng = np.random.default_rng(8)
    sizes = rng.poisson(1000, size=n_img)
    sizes[[17, 600]] = 0
    sizes = (sizes * (n / sizes.sum())).astype(np.int64)
    sizes[-1] += n - sizes.sum()
    rag = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    hr = ops.hist(idx, torch.from_numpy(rag).to(dev), k, "l1")
    code = idx.cpu().numpy()
    for i in (0, 17, 18, 599, 600, 601, n_img - 1):
        ref_row = oracle.build_histogram([code[rag[i]:rag[i + 1]]], k, "l1")[0]
        assert np.array_equal(hr[i].cpu().numpy(), ref_row)


def test_lloyd_trajectory_matches_oracle_downscaled():
    """BASELINE config 4 scaled down (200k rows, 1024 centroids, same generator and init rule):
    inertia per iteration and final labels against the float64 restatement of kmeans2."""
    from image_classification_amd import kmeans2
    n, k, iters = 200_000, 1024, 4
    x = sift_like(n, n_modes=4096, seed=7)
    init = x[np.random.default_rng(8).permutation(n)[:k]].copy()
    cb, labels = kmeans2(x, init, iter=iters, minit="matrix")
    ref_cb, ref_labels, history = oracle.kmeans2_matrix(x, init, iters)
    assert (labels == ref_labels).mean() >= 0.9999
    err = np.linalg.norm(cb - ref_cb, axis=1) / np.maximum(np.linalg.norm(ref_cb, axis=1), 1e-30)
    assert err.max() <= 1e-4
    # inertia (sum of squared distances to the nearest centroid) after every iteration
    x64 = x.astype(np.float64)
    for it in (1, 2, 3):
        cb_i, _ = kmeans2(x, init, iter=it, minit="matrix")
        _, d_dev = oracle.vq(x64, cb_i.astype(np.float64))
        _, d_ref = oracle.vq(x64, history[it - 1])
        assert abs((d_dev ** 2).sum() - (d_ref ** 2).sum()) <= 1e-5 * (d_ref ** 2).sum()
