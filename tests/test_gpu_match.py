"""GPU parity of the match compaction and the batched homography RANSAC.

The checker is oracle/ransac.py, the restatement of the reference's own ransac() /
HomographyModel (model/robust_matching.py:228-374).  Both sides get the same samples (the
partitions np.random.shuffle would produce), so the kept iteration, the inlier mask and the
acceptance must be identical and the refitted homography / mean error agree to rounding.
"""
import numpy as np
import pytest
import torch

import oracle
from oracle import ransac_ref
from image_classification_amd import ops

pytestmark = pytest.mark.gpu


def _planted(rng, n, inlier_frac, noise=0.3, span=600.0):
    H = np.array([[0.9, 0.12, 40.0], [-0.07, 1.05, -25.0], [1.2e-4, -0.8e-4, 1.0]])
    H[:2, :2] += rng.normal(0, 0.05, size=(2, 2))
    src = rng.random((n, 2)) * span
    t = np.c_[src, np.ones(n)].dot(H.T)
    dst = t[:, :2] / t[:, 2:] + rng.normal(0, noise, size=(n, 2))
    n_in = int(round(n * inlier_frac))
    dst[n_in:] = rng.random((n - n_in, 2)) * span
    perm = rng.permutation(n)
    truth = np.zeros(n, bool)
    truth[:n_in] = True
    return np.ascontiguousarray(np.c_[src, dst][perm]), truth[perm], H


def test_compact_matches_vs_numpy():
    rng = np.random.default_rng(2)
    sizes = [0, 1, 63, 64, 65, 500, 0, 1000, 7]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    rows = int(off[-1])
    mask = (rng.random(rows) < 0.3).astype(np.uint8)
    mask[off[3]:off[4]] = 1          # a candidate whose rows all match
    mask[off[4]:off[5]] = 0          # one with no match
    nq = 300
    i1 = rng.integers(0, nq, size=rows).astype(np.int32)
    kp_q = rng.random((nq, 2)) * 100
    kp_rows = rng.random((rows, 2)) * 100
    out_off, mq, mi, pts = ops.compact_matches(
        torch.from_numpy(mask).cuda(), torch.from_numpy(i1).cuda(), torch.from_numpy(off).cuda(),
        torch.from_numpy(kp_q).cuda(), torch.from_numpy(kp_rows).cuda())
    out_off, mq, mi, pts = out_off.cpu().numpy(), mq.cpu().numpy(), mi.cpu().numpy(), pts.cpu().numpy()
    pos = 0
    for c in range(len(sizes)):
        local = np.nonzero(mask[off[c]:off[c + 1]])[0]
        assert out_off[c] == pos and out_off[c + 1] == pos + len(local)
        assert np.array_equal(mi[pos:pos + len(local)], local)
        assert np.array_equal(mq[pos:pos + len(local)], i1[off[c] + local])
        assert np.array_equal(pts[pos:pos + len(local), :2], kp_q[i1[off[c] + local]])
        assert np.array_equal(pts[pos:pos + len(local), 2:], kp_rows[off[c] + local])
        pos += len(local)
    # without keypoints: lists only
    out2, mq2, mi2, none = ops.compact_matches(
        torch.from_numpy(mask).cuda(), torch.from_numpy(i1).cuda(), torch.from_numpy(off).cuda())
    assert none is None and np.array_equal(out2.cpu().numpy(), out_off)
    assert np.array_equal(mq2.cpu().numpy()[:pos], mq[:pos])


def _run_case(cands, n_iter, th, seed, min_inliers=10, mean_rule=True):
    rng = np.random.RandomState(seed)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int64)
    pts = np.ascontiguousarray(np.concatenate(cands))
    samples = np.zeros((len(cands), n_iter, 4), dtype=np.int32)
    parts_all = []
    for c, data in enumerate(cands):
        parts = []
        for it in range(n_iter):
            if len(data) >= 4:
                a, b = ransac_ref.random_partition(4, len(data), rng)   # np.random.shuffle of arange
                samples[c, it] = a
                parts.append((a.copy(), b.copy()))
        parts_all.append(parts)
    H, n_inl, err, best_it, mask = ops.ransac_homography(
        torch.from_numpy(pts).cuda(), torch.from_numpy(off).cuda(), n_iter, th,
        torch.from_numpy(samples).cuda(), 0, min_inliers, mean_rule)
    H, n_inl, err, best_it, mask = (t.cpu().numpy() for t in (H, n_inl, err, best_it, mask))
    for c, data in enumerate(cands):
        got_mask = mask[off[c]:off[c + 1]]
        if len(data) < max(4, min_inliers):     # can never be accepted: not even scored
            assert n_inl[c] == 0 and not got_mask.any() and best_it[c] == -1
            assert not H[c].any() and np.isinf(err[c])
            continue
        info = {}
        fit, ref_mask, ref_err = ransac_ref.ransac(
            data, ransac_ref.HomographyModel, 4, 0, th, partitions=parts_all[c], num_it=n_iter,
            min_inliers=min_inliers, accept_mean_err=mean_rule, info=info)
        assert best_it[c] == info["best_it"], "candidate %d kept another iteration" % c
        assert np.array_equal(got_mask.astype(int), ref_mask), "candidate %d inlier mask" % c
        assert n_inl[c] == ref_mask.sum()
        if np.isfinite(ref_err):
            assert abs(err[c] - ref_err) <= 1e-6 * max(ref_err, 1e-3)
        if fit is not None:
            assert np.abs(H[c] - fit).max() <= 1e-6 * np.abs(fit).max()
        else:
            assert not H[c].any()
    return n_inl, mask, off


def test_ransac_same_samples_as_reference_loop():
    rng = np.random.default_rng(11)
    cands, truths = [], []
    for n, frac in [(60, 0.7), (4, 1.0), (3, 1.0), (12, 1.0), (300, 0.4), (40, 0.0), (25, 0.5),
                    (9, 1.0), (130, 0.9)]:
        data, truth, _ = _planted(rng, n, frac)
        cands.append(data)
        truths.append(truth)
    n_iter = ransac_ref.ransac_num_iterations(100, 0.95, 0.5)
    assert n_iter == 46
    n_inl, mask, off = _run_case(cands, n_iter, 3.0, seed=5)
    # the planted structure is recovered where it is large enough, and pure noise is rejected
    assert n_inl[0] >= 40 and n_inl[4] >= 110 and n_inl[8] >= 110
    assert n_inl[1] == 0 and n_inl[2] == 0 and n_inl[5] == 0 and n_inl[7] == 0
    for c in (0, 4, 8):
        got = mask[off[c]:off[c + 1]].astype(bool)
        assert (got & ~truths[c]).sum() <= 2 and (got & truths[c]).sum() >= 0.8 * truths[c].sum()


def test_ransac_without_the_mean_error_rule_and_many_iterations():
    rng = np.random.default_rng(12)
    cands = [_planted(rng, n, f)[0] for n, f in [(50, 0.3), (8, 1.0), (20, 0.25)]]
    _run_case(cands, 200, 2.0, seed=9, min_inliers=4, mean_rule=False)


def _cv_case(cands, n_iter, th, seed):
    """Device loop with OpenCV's rules against its restatement, same samples on both sides."""
    rng = np.random.RandomState(seed)
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int64)
    pts = np.ascontiguousarray(np.concatenate(cands))
    samples = np.zeros((len(cands), n_iter, 4), dtype=np.int32)
    for c, data in enumerate(cands):
        if len(data) >= 4:
            for it in range(n_iter):
                samples[c, it] = rng.permutation(len(data))[:4]
    H, n_inl, err, best_it, mask = ops.ransac_homography(
        torch.from_numpy(pts).cuda(), torch.from_numpy(off).cuda(), n_iter, th,
        torch.from_numpy(samples).cuda(), 0, 4, False, cv_rules=True)
    H, n_inl, err, best_it, mask = (t.cpu().numpy() for t in (H, n_inl, err, best_it, mask))
    ran = []
    for c, data in enumerate(cands):
        got_mask = mask[off[c]:off[c + 1]]
        info = {}
        fit, ref_mask = ransac_ref.cv_find_homography_ransac(data, th, samples[c],
                                                             max_iters=n_iter, info=info)
        ran.append(info["iterations"])
        assert best_it[c] == info["best_it"], "candidate %d kept another iteration" % c
        assert np.array_equal(got_mask.astype(int), ref_mask), "candidate %d inlier mask" % c
        assert n_inl[c] == ref_mask.sum()
        if fit is not None:
            assert np.abs(H[c] - fit).max() <= 1e-6 * np.abs(fit).max(), "candidate %d" % c
            # the polish does not make the fit worse than the plain refit over the inliers
            inl = data[ref_mask.astype(bool)]
            plain, _ = ransac_ref.cv_find_homography_ransac(data, th, samples[c],
                                                            max_iters=n_iter, polish=False)
            assert ransac_ref.cv_sq_error(inl, H[c]).sum() <= \
                ransac_ref.cv_sq_error(inl, plain).sum() * (1 + 1e-9)
        else:
            assert not H[c].any() and n_inl[c] == 0
    return n_inl, mask, off, ran


def test_ransac_opencv_rules_same_samples_as_the_restated_loop():
    rng = np.random.default_rng(21)
    specs = [(60, 0.7), (4, 1.0), (3, 1.0), (12, 1.0), (300, 0.4), (40, 0.0), (25, 0.5),
             (9, 1.0), (130, 0.9), (500, 0.15), (64, 0.3), (5, 0.8)]
    built = [_planted(rng, n, f) for n, f in specs]
    cands = [b[0] for b in built]
    n_inl, mask, off, ran = _cv_case(cands, 2000, 3.0, seed=3)
    # the iteration limit adapts: mostly-inlier candidates stop after a handful of samples, a
    # candidate without structure runs to the end
    assert ran[0] < 40 and ran[8] < 10 and ran[3] <= 5 and ran[5] == 2000
    assert n_inl[2] == 0                      # three pairs: no model
    assert n_inl[1] == 4 and n_inl[3] == 12   # exact fits keep every pair
    for c in (0, 4, 8):     # (15 % inliers need more than 2000 samples to be found reliably)
        truth = built[c][1]
        got = mask[off[c]:off[c + 1]].astype(bool)
        assert (got & truth).sum() >= 0.85 * truth.sum() and (got & ~truth).sum() <= 3
    # a tight threshold (the logo pipeline's 0.5) and few iterations
    _cv_case(cands, 64, 0.5, seed=4)
    _cv_case(cands[:3], 1, 3.0, seed=5)


def test_ransac_device_generated_samples():
    rng = np.random.default_rng(13)
    specs = [(80, 0.6), (200, 0.35), (30, 0.0), (4, 1.0), (2, 1.0), (64, 1.0)]
    built = [_planted(rng, n, f) for n, f in specs]
    cands = [b[0] for b in built]
    off = np.concatenate([[0], np.cumsum([len(c) for c in cands])]).astype(np.int64)
    pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(cands))).cuda()
    offd = torch.from_numpy(off).cuda()
    a = ops.ransac_homography(pts, offd, 300, 3.0, None, 1234)
    b = ops.ransac_homography(pts, offd, 300, 3.0, None, 1234)
    c = ops.ransac_homography(pts, offd, 300, 3.0, None, 99)
    for x, y in zip(a, b):
        assert torch.equal(x, y)                       # same seed, same answer
    assert not torch.equal(a[3], c[3])                 # another seed draws other samples
    n_inl, mask = a[1].cpu().numpy(), a[4].cpu().numpy().astype(bool)
    assert n_inl[2] == 0 and n_inl[3] == 0 and n_inl[4] == 0
    for cidx in (0, 1, 5):
        truth = built[cidx][1]
        got = mask[off[cidx]:off[cidx + 1]]
        assert (got & truth).sum() >= 0.9 * truth.sum() and (got & ~truth).sum() <= 2
        H = a[0][cidx].cpu().numpy()
        data = cands[cidx][truth]
        assert ransac_ref.HomographyModel.get_error(data, H).mean() < 1.0
    # no iterations: everything rejected
    z = ops.ransac_homography(pts, offd, 0, 3.0)
    assert int(z[1].sum()) == 0 and not z[4].any()
