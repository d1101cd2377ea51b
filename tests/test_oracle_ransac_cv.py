"""The restated loop of cv2.findHomography(..., cv2.RANSAC, th) in oracle/ransac.py (the
reference's default estimator, model/robust_matching.py:214-216).  OpenCV is not installed and the
reference holds no output of it: these tests pin the restatement to the published formulae and to
properties, not to OpenCV's output (parity unpinned)."""
import numpy as np

from oracle import ransac_ref as R


def test_iteration_limit_formula():
    # log(1 - p) / log(1 - (1 - ep)^4), rounded to nearest, capped by the limit in force
    assert R.cv_update_num_iters(0.99, 0.5, 4, 2000) == 71        # 71.36
    assert R.cv_update_num_iters(0.995, 0.4, 4, 2000) == 38       # 38.2
    assert R.cv_update_num_iters(0.995, 0.4, 4, 20) == 20         # never raises the limit
    assert R.cv_update_num_iters(0.995, 0.0, 4, 2000) == 0        # no outliers: stop
    assert R.cv_update_num_iters(0.995, 1.0, 4, 2000) == 2000     # no inliers: no information
    limits = [R.cv_update_num_iters(0.995, ep, 4, 2000) for ep in np.linspace(0, 1, 41)]
    assert all(a <= b for a, b in zip(limits, limits[1:]))


def _scene(rng, n, n_in, noise):
    H = np.array([[1.1, -0.2, 15.0], [0.15, 0.95, -30.0], [2e-4, 1e-4, 1.0]])
    src = rng.random((n, 2)) * 500
    t = np.c_[src, np.ones(n)].dot(H.T)
    dst = t[:, :2] / t[:, 2:] + rng.normal(0, noise, (n, 2))
    dst[n_in:] = rng.random((n - n_in, 2)) * 500
    return np.c_[src, dst], H


def test_loop_keeps_the_first_best_model_and_stops_early():
    rng = np.random.default_rng(1)
    data, H = _scene(rng, 150, 110, 0.2)
    samples = [rng.permutation(150)[:4] for _ in range(2000)]
    info = {}
    fit, mask = R.cv_find_homography_ransac(data, 2.0, samples, info=info)
    assert mask[:110].sum() >= 105 and mask[110:].sum() <= 2
    assert 0 <= info["best_it"] < info["iterations"] < 100
    # the kept model is the first with that many inliers among the iterations that ran
    counts = []
    for it in range(info["iterations"]):
        m = R.HomographyModel.fit(data[samples[it]])
        counts.append(int((R.cv_sq_error(data, m) <= 4.0).sum()) if m is not None else 0)
    assert info["best_it"] == int(np.argmax(counts)) and counts[info["best_it"]] == mask.sum()
    assert np.abs(fit - H).max() < 0.5 and abs(fit[2, 2] - 1) < 1e-12


def test_polish_lowers_the_inliers_squared_error():
    rng = np.random.default_rng(2)
    data, _ = _scene(rng, 80, 80, 0.5)
    plain = R.HomographyModel.fit(data)
    polished = R.cv_lm_refine(plain, data)
    assert R.cv_sq_error(data, polished).sum() <= R.cv_sq_error(data, plain).sum()
    # an exact model is a fixed point
    exact, H = _scene(rng, 30, 30, 0.0)
    again = R.cv_lm_refine(H, exact)
    assert np.abs(again - H).max() < 1e-9


def test_fewer_than_four_pairs_or_no_consensus():
    rng = np.random.default_rng(3)
    data, _ = _scene(rng, 3, 3, 0.0)
    fit, mask = R.cv_find_homography_ransac(data, 3.0, [])
    assert fit is None and mask.sum() == 0
    noise = np.c_[rng.random((40, 2)) * 500, rng.random((40, 2)) * 500]
    samples = [rng.permutation(40)[:4] for _ in range(50)]
    info = {}
    fit, mask = R.cv_find_homography_ransac(noise, 0.01, samples, info=info)
    # a sample always fits itself: four inliers, the model of the first sample is kept
    assert mask.sum() == 4 and info["best_it"] == 0 and info["iterations"] == 50
