"""Size-independent properties at the full per-GPU sizes of BASELINE configs 3, 4 and 5 (config 2
is covered in test_gpu_reduce.py).  No CPU oracle can run at these sizes; bounded samples are
checked against it."""
import numpy as np
import pytest
import torch

import oracle
from image_classification_amd import ops
from image_classification_amd.synthetic import image_offsets, sift_like_device

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_config3_logo_search_both_directions():
    # 4k query descriptors, 2M database descriptors, 5 % of the database rows are noisy copies
    nq, ndb = 4000, 2_000_000
    q = sift_like_device(nq, n_modes=4096, seed=5, device=DEV)
    db = sift_like_device(ndb, n_modes=4096, seed=6, device=DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(3)
    planted = torch.randperm(ndb, device=DEV, generator=g)[:ndb // 20]
    src = torch.randint(0, nq, (planted.numel(),), device=DEV, generator=g)
    db[planted] = q[src] + 2.0 * torch.randn((planted.numel(), 128), device=DEV, generator=g)
    # (i) reference direction: database rows searched in the query index
    i1, d1, i2, d2, mask = ops.assign_top2(db, q, 0.8)
    assert torch.equal(i1[planted].long(), src)              # every planted row finds its source
    assert mask[planted].float().mean() > 0.999               # and passes the ratio test
    rest = torch.ones(ndb, dtype=torch.bool, device=DEV)
    rest[planted] = False
    assert mask[rest].float().mean() < 0.02                   # unrelated rows rarely do
    assert torch.all(d1 <= d2)
    # row-sharded (8 blocks of 250k) == unsharded
    for lo in range(0, ndb, 250_000):
        part = ops.assign_top2(db[lo:lo + 250_000], q, 0.8)
        assert torch.equal(part[0], i1[lo:lo + 250_000]) and torch.equal(part[4], mask[lo:lo + 250_000])
    # (ii) transposed: queries searched in the database, 8 index shards merged
    full = ops.assign_top2(q, db, 0.8)
    bounds = list(range(0, ndb + 1, 250_000))
    recs = torch.stack([ops.assign_top2_rec(q, db[lo:hi]) for lo, hi in zip(bounds[:-1], bounds[1:])])
    merged = ops.top2_merge(recs.contiguous(), bounds[:-1], 0.8)
    # every shard's pair is exact, so the merge can differ from the unsharded search only where two
    # different squared distances round to the same float32 square root (measured: no row)
    same = (merged[0] == full[0].long()) & (merged[2] == full[2].long())
    assert same.float().mean() >= 0.9995
    assert torch.equal(merged[1][same], full[1][same])
    # bounded sample against the float64 oracle (queries vs the first 200k database rows):
    # measured agreement 1.0 for both neighbours; one row of 500 is the margin
    rd, ri = oracle.topk(q[:500].cpu().numpy(), db[:200_000].cpu().numpy(), 3)
    got = ops.assign_top2(q[:500], db[:200_000], 0.8)
    direct = ops.assign_top2_exact(q[:500], db[:200_000], 0.8)
    for col, rank in ((0, 0), (2, 1)):
        must, bad, agree = oracle.check_indices(got[col].cpu().numpy(), ri, rd, rank)
        print("config 3 transposed, neighbour %d: %d must-match rows, %d bad, agreement %.4f"
              % (rank + 1, must, bad, agree))
        assert bad == 0 and agree >= 0.998
        assert torch.equal(got[col], direct[col])             # and the direct float32 search, exactly


def test_config4_one_lloyd_iteration_share():
    # one GPU's share of 10M x 65536: 1.25M rows
    n, k = 1_250_000, 65_536
    x = sift_like_device(n, n_modes=4096, seed=7, device=DEV)
    c = x[torch.randperm(n, device=DEV)[:k]].clone()
    idx, dist = ops.assign_top1(x, c)
    sums, counts = ops.centroid_partials(x, idx, k)
    assert int(counts.sum()) == n and torch.equal(counts, torch.bincount(idx.long(), minlength=k))
    assert torch.allclose(sums.sum(0), x.double().sum(0), rtol=1e-11)
    new_c, has = ops.centroid_finalize(sums, counts, c)
    assert torch.equal(has.bool(), counts > 0)
    # Lloyd's guarantee: assigning to the new centroids does not raise the inertia
    idx2, dist2 = ops.assign_top1(x, new_c, hint=idx)
    idx2_plain, dist2_plain = ops.assign_top1(x, new_c)
    assert torch.equal(idx2, idx2_plain) and torch.equal(dist2, dist2_plain)
    assert float((dist2.double() ** 2).sum()) <= float((dist.double() ** 2).sum()) * (1 + 1e-9)
    # every initial centroid is a data row: that row is at distance 0 from its own centroid
    # sample against the oracle
    xs = x[:3000].cpu().numpy()
    rd, ri = oracle.topk(xs, c.cpu().numpy(), 2)
    must, bad, agree = oracle.check_indices(idx[:3000].cpu().numpy(), ri, rd, 0)
    assert bad == 0 and agree >= 0.999


def test_config5_video_batch_share():
    # 2000 frames x 2000 descriptors, 10k words, signed-square-root + L2 histograms
    n_img, per, k = 2000, 2000, 10_000
    x = sift_like_device(n_img * per, n_modes=2048, seed=9, device=DEV)
    cb = sift_like_device(k, n_modes=2048, seed=4, device=DEV)
    offsets = torch.from_numpy(image_offsets(n_img, per)).to(DEV)
    idx, _ = ops.assign_top1(x, cb, want_dist=False)
    h = ops.hist(idx, offsets, k, "power_l2", 0.5)
    assert torch.allclose((h * h).sum(1), torch.ones(n_img, dtype=torch.float64, device=DEV), rtol=1e-12)
    raw = ops.hist(idx, offsets, k, "none")
    assert int(raw.sum()) == n_img * per
    ref = oracle.build_histogram([idx[i * per:(i + 1) * per].cpu().numpy() for i in (0, 999, 1999)],
                                 k, "power_l2", 0.5)
    assert oracle.check_close(h[[0, 999, 1999]].cpu().numpy(), ref, rel=1e-12, abs_floor=1e-300) <= 1e-12
    l1 = ops.hist(idx, offsets, k, "l1")
    assert np.array_equal(l1[[0, 999, 1999]].cpu().numpy(),
                          oracle.build_histogram([idx[i * per:(i + 1) * per].cpu().numpy()
                                                  for i in (0, 999, 1999)], k, "l1"))


def test_config4_full_size_twenty_iterations_and_the_eight_way_sum():
    """BASELINE config 4 whole on one GPU: 10M rows, 65 536 centroids, 20 Lloyd iterations.  Per
    iteration: the inertia must not rise; the centroid sums of the eight 1.25M-row shards a node
    would hold, added in rank order (what the all-reduce leaves, up to its own order), must equal
    the unsharded sums to 1e-12 and give centroids within the 1e-4 the design asks for."""
    n, k, shards = 10_000_000, 65_536, 8
    x = sift_like_device(n, n_modes=4096, seed=7, device=DEV)
    g = torch.Generator(device=DEV)
    g.manual_seed(8)
    c = x[torch.randperm(n, device=DEV, generator=g)[:k]].clone()   # first K of a permutation
    bounds = [n * r // shards for r in range(shards + 1)]
    labels, prev = None, float("inf")
    inertias = []
    for it in range(20):
        labels, dist = ops.assign_top1(x, c, hint=labels)
        inertia = float((dist.double() ** 2).sum())
        assert inertia <= prev * (1 + 1e-7), (it, inertia, prev)
        prev = inertia
        inertias.append(inertia)
        sums, counts = ops.centroid_partials(x, labels, k)
        acc_s = torch.zeros_like(sums)
        acc_c = torch.zeros_like(counts)
        for lo, hi in zip(bounds[:-1], bounds[1:]):
            s_r, c_r = ops.centroid_partials(x[lo:hi], labels[lo:hi], k)
            acc_s += s_r
            acc_c += c_r
        assert torch.equal(acc_c, counts) and int(counts.sum()) == n
        scale = sums.abs().max()
        assert float((acc_s - sums).abs().max() / scale) <= 1e-12
        new_c, _has = ops.centroid_finalize(sums, counts, c)
        sharded_c, _ = ops.centroid_finalize(acc_s, acc_c, c)
        err = ((sharded_c - new_c).abs() / new_c.abs().clamp_min(1e-3)).max()
        assert float(err) <= 1e-4, (it, float(err))
        c = new_c
    print("config 4 inertia, iterations 1 / 2 / 5 / 20: %.6e %.6e %.6e %.6e"
          % (inertias[0], inertias[1], inertias[4], inertias[19]))
    assert inertias[19] < 0.9 * inertias[0]                 # the fit went somewhere
    # the last labels against the direct float32 search on a sample of rows
    c_prev_rows = torch.arange(0, n, n // 5000, device=DEV)[:5000]
    e1 = ops.assign_top2_exact(x[c_prev_rows].contiguous(), c, 0.8)[0]
    chk, _ = ops.assign_top1(x[c_prev_rows].contiguous(), c, want_dist=False)
    assert torch.equal(chk, e1)


def test_config5_full_size_equals_its_five_shards():
    """BASELINE config 5 whole on one GPU: 10 000 frames x 2000 descriptors, 10k words, power-L2
    histograms; the run over all 20M rows must equal, bit for bit, the five 4M-row shards."""
    n_img, per, k, shards = 10_000, 2000, 10_000, 5
    x = sift_like_device(n_img * per, n_modes=2048, seed=9, device=DEV)
    cb = sift_like_device(k, n_modes=2048, seed=4, device=DEV)
    aux = ops.codebook_aux(cb)
    offsets = torch.from_numpy(image_offsets(n_img, per)).to(DEV)
    idx, _ = ops.assign_top1(x, cb, aux, want_dist=False)
    h = ops.hist(idx, offsets, k, "power_l2", 0.5)
    assert torch.allclose((h * h).sum(1), torch.ones(n_img, dtype=torch.float64, device=DEV), rtol=1e-12)
    fs = n_img // shards
    sub_off = torch.from_numpy(image_offsets(fs, per)).to(DEV)
    for s in range(shards):
        lo, hi = s * fs * per, (s + 1) * fs * per
        idx_s, _ = ops.assign_top1(x[lo:hi], cb, aux, want_dist=False)
        assert torch.equal(idx_s, idx[lo:hi])
        h_s = ops.hist(idx_s, sub_off, k, "power_l2", 0.5)
        assert torch.equal(h_s, h[s * fs:(s + 1) * fs])
        del h_s, idx_s
    raw_total = int(ops.hist(idx, offsets, k, "none").sum())
    assert raw_total == n_img * per
