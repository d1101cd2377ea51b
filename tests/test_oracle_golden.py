"""Pins the CPU oracle against the golden vectors (outputs of scipy / scikit-learn / numpy for the
reference's call patterns, see tests/golden/make_golden.py).  Runs without a GPU."""
import os

import numpy as np
import pytest

import oracle

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def test_vq_matches_scipy_float64_vocab():
    g = load("g1_vq")
    code, dist = oracle.vq(g["x"], g["c64"])
    assert code.dtype == np.int32 and dist.dtype == np.float64
    assert np.array_equal(code, g["code64"])
    np.testing.assert_allclose(dist, g["dist64"], rtol=1e-9)


def test_vq_matches_scipy_float32_vocab_within_float32_rounding():
    g = load("g1_vq")
    code, dist = oracle.vq(g["x"], g["c32"])
    rd, ri = oracle.topk(g["x"], g["c32"], 2)
    # scipy computed in float32 here: allow its picks to differ only inside the 1e-5 gap
    must, bad, agree = oracle.check_indices(g["code32"], ri, rd, 0)
    assert bad == 0 and agree > 0.999
    assert np.array_equal(code, ri[:, 0])
    same = code == g["code32"]
    np.testing.assert_allclose(dist[same], g["dist32"][same], rtol=2e-5)


def test_ties_pick_lowest_index():
    g = load("g2_ties")
    code, dist = oracle.vq(g["x"], g["c"])
    assert np.array_equal(code, g["code"])
    assert code.max() < 48
    np.testing.assert_allclose(dist, g["dist"], rtol=1e-9, atol=1e-9)
    assert np.all(dist[:24] == 0)
    rd, ri = oracle.topk(g["x"], g["c"], 3)
    assert np.array_equal(ri[:, 0], g["code"])
    dup = g["code"][:16]
    assert np.array_equal(ri[:16, 1], dup + 48)  # the duplicate is the second neighbour


def test_histogram_l1_matches_numpy_rule():
    g = load("g3_hist")
    off = g["offsets"]
    codes = [g["codes"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    h = oracle.build_histogram(codes, int(g["k"]), "l1")
    assert h.dtype == np.float64
    assert np.array_equal(h, g["hists"])
    assert np.all(h[1] == 0)
    # descriptors=None images are quantised to code=[] (model/features.py:1167-1168)
    codes[5] = []
    h2 = oracle.build_histogram(codes, int(g["k"]), "l1")
    assert np.all(h2[5] == 0)


def test_histogram_extension_modes_are_consistent():
    g = load("g3_hist")
    off = g["offsets"]
    codes = [g["codes"][off[i]:off[i + 1]] for i in range(len(off) - 1)]
    k = int(g["k"])
    raw = oracle.build_histogram(codes, k, "none")
    l2 = oracle.build_histogram(codes, k, "l2")
    p2 = oracle.build_histogram(codes, k, "power_l2")
    for i in range(len(codes)):
        if raw[i].sum() == 0:
            assert np.all(l2[i] == 0) and np.all(p2[i] == 0)
            continue
        np.testing.assert_allclose(np.linalg.norm(l2[i]), 1.0, rtol=1e-12)
        np.testing.assert_allclose(np.linalg.norm(p2[i]), 1.0, rtol=1e-12)
        np.testing.assert_allclose(l2[i] * np.linalg.norm(raw[i]), raw[i], rtol=1e-12)
        np.testing.assert_allclose(p2[i] ** 2 * raw[i].sum(), raw[i], rtol=1e-12, atol=1e-12)


def test_kneighbors_matches_sklearn_both_directions():
    g = load("g4_knn")
    dist, ind = oracle.kneighbors(g["query"], g["cand"], 2)
    assert ind.dtype == np.int64 and dist.dtype == np.float64
    assert np.array_equal(ind, g["ind"])
    np.testing.assert_allclose(dist, g["dist"], rtol=1e-6)
    mq, mi = oracle.ratio_matches(dist, ind, 0.8)
    assert np.array_equal(mi, np.nonzero(g["mask"])[0])
    assert np.array_equal(mq, g["ind"][g["mask"], 0])
    assert set(g["planted"]).issubset(set(mi))
    dist_t, ind_t = oracle.kneighbors(g["cand"], g["query"], 2)
    assert np.array_equal(ind_t, g["ind_t"])
    np.testing.assert_allclose(dist_t, g["dist_t"], rtol=1e-6)


def test_robust_match_stage_applies_min_points_guard():
    g = load("g4_knn")
    cands = [g["cand"][:400], g["cand"][400:405], None, g["cand"][405:]]
    res = oracle.robust_match_descriptors(g["query"], cands, min_points=5, thresh=0.8)
    assert len(res) == 4
    assert len(res[1][0]) == 0 and len(res[2][0]) == 0  # 5 rows is not > min_points
    full = np.nonzero(g["mask"])[0]
    assert np.array_equal(res[0][1], full[full < 400])
    assert np.array_equal(res[3][1] + 405, full[full >= 405])


def test_kmeans2_matches_scipy_including_empty_cluster_rule():
    g = load("g5_kmeans")
    data = g["data"].astype(np.float64)
    cb, label, hist = oracle.kmeans2_matrix(data, g["init"], 5)
    assert np.array_equal(label, g["label"])
    np.testing.assert_allclose(cb, g["cb"], rtol=1e-10)
    np.testing.assert_allclose(hist[0], g["cb1"], rtol=1e-10)
    cb_e, label_e, _ = oracle.kmeans2_matrix(data, g["init_e"], 3)
    assert np.array_equal(label_e, g["label_e"])
    np.testing.assert_allclose(cb_e, g["cb_e"], rtol=1e-10)
    assert np.all(cb_e[3] == 1e4)  # the empty cluster kept its previous centroid
    # minit='points': same generator call as scipy's _kpoints
    rng = np.random.default_rng(int(g["seed_p"]))
    init_p = data[rng.choice(data.shape[0], size=16, replace=False)]
    cb_p, label_p, _ = oracle.kmeans2_matrix(data, init_p, 5)
    assert np.array_equal(label_p, g["label_p"])
    np.testing.assert_allclose(cb_p, g["cb_p"], rtol=1e-10)


def test_compute_vocab_features_matches_sklearn_batches():
    g = load("g6_cvf")
    codes = oracle.compute_vocab_features(g["vocab"], g["feat"])
    assert codes.dtype == np.int64
    rd, ri = oracle.topk(g["feat"], g["vocab"], 2)
    must, bad, agree = oracle.check_indices(g["codes"], ri, rd, 0)
    assert bad == 0 and agree > 0.999   # sklearn computed in float32 blocks
    assert np.array_equal(codes, ri[:, 0])


def test_top2_merge_equals_unsharded_search():
    g = load("g4_knn")
    bounds = [0, 100, 101, 640, 1000]
    shards = [oracle.topk(g["query"], g["cand"][lo:hi], 2) for lo, hi in zip(bounds[:-1], bounds[1:])]
    dist, ind = oracle.top2_merge(shards, bounds[:-1])
    assert np.array_equal(ind, g["ind_t"])
    np.testing.assert_allclose(dist, g["dist_t"], rtol=1e-6)


@pytest.mark.parametrize("n,k", [(0, 5), (3, 1)])
def test_degenerate_shapes(n, k):
    x = np.zeros((n, 8))
    c = np.arange(k * 8, dtype=np.float64).reshape(k, 8)
    code, dist = oracle.vq(x, c)
    assert code.shape == (n,) and dist.shape == (n,)
    d3, i3 = oracle.topk(x, c, 3)
    assert d3.shape == (n, 3)
    if n:
        assert np.all(i3[:, k:] == -1) and np.all(np.isinf(d3[:, k:]))


# ---- the restated RANSAC / homography model: known-answer checks ---------------------------------
def _project(H, pts):
    t = np.c_[pts, np.ones(len(pts))].dot(H.T)
    return t[:, :2] / t[:, 2:]


def test_homography_fit_recovers_a_known_map_and_error_rule():
    from oracle import ransac_ref
    rng = np.random.default_rng(0)
    H = np.array([[0.8, -0.2, 31.0], [0.15, 1.1, -12.0], [2e-4, -1e-4, 1.0]])
    src = rng.random((40, 2)) * 400
    data = np.c_[src, _project(H, src)]
    for n in (4, 5, 40):                                  # minimal and over-determined
        fit = ransac_ref.HomographyModel.fit(data[:n])
        assert np.abs(fit - H).max() <= 1e-7 * np.abs(H).max()
    err = ransac_ref.HomographyModel.get_error(data, H)
    assert err.shape == (40,) and err.max() < 1e-6
    # the error is the Euclidean reprojection distance (plus the reference's epsilon quirk)
    moved = data.copy()
    moved[:, 2] += 3.0
    moved[:, 3] -= 4.0
    assert np.allclose(ransac_ref.HomographyModel.get_error(moved, H), 5.0, atol=1e-6)
    # degenerate input (all source points equal) gives no model
    same = np.c_[np.ones((6, 2)), rng.random((6, 2))]
    assert ransac_ref.HomographyModel.fit(same) is None


def test_ransac_restatement_semantics():
    from oracle import ransac_ref
    rng = np.random.RandomState(4)
    H = np.array([[1.05, 0.02, -8.0], [-0.03, 0.97, 14.0], [1e-5, 3e-5, 1.0]])
    src = rng.rand(80, 2) * 500
    dst = _project(H, src) + rng.randn(80, 2) * 0.2
    dst[50:] = rng.rand(30, 2) * 500                      # 30 outliers
    data = np.c_[src, dst]
    assert ransac_ref.ransac_num_iterations(100, 0.95, 0.5) == 46   # int(log(.05)/log(1-.5^4))
    assert ransac_ref.ransac_num_iterations(10, 0.95, 0.5) == 10
    info = {}
    fit, mask, err = ransac_ref.ransac(data, ransac_ref.HomographyModel, 4, 100, 3.0, rng=rng,
                                       info=info)
    assert fit is not None and mask[:50].sum() >= 45 and mask[50:].sum() <= 1 and err < 1.0
    assert 0 <= info["best_it"] < 46
    # fewer than MIN_INLIERS_TH inliers, or a mean error above the threshold: rejected
    few = data[np.r_[0:8, 50:80]]
    fit, mask, err = ransac_ref.ransac(few, ransac_ref.HomographyModel, 4, 100, 3.0, rng=rng)
    assert fit is None and mask.sum() == 0
    # random_partition is np.random.shuffle of arange: first n and the rest
    a, b = ransac_ref.random_partition(4, 10, np.random.RandomState(1))
    ref = np.arange(10)
    np.random.RandomState(1).shuffle(ref)
    assert np.array_equal(a, ref[:4]) and np.array_equal(b, ref[4:])
