"""Restatement of the reference's own RANSAC and homography model (test oracle).

Follows model/robust_matching.py:228-374 statement by statement: `HomographyModel.get_error`
(:259-283), `ransac` (:291-365) and `random_partition` (:368-374) are pure numpy/scipy there and
are restated as they are written.  `HomographyModel.fit` (:237-257) is one call of
cv2.findHomography(method=0); OpenCV is not installed here, so `fit` restates the published
algorithm of that function without its final Levenberg-Marquardt polish: Hartley normalisation of
both point sets (centroid, mean absolute deviation per axis), the 9x9 matrix L^T L of the
direct linear transform, its eigenvector of the smallest eigenvalue, denormalisation, H[2,2] = 1.
For exactly four points that is the unique homography; for more points the value OpenCV would
return differs by the polish step, and that difference is unpinned.

The reference's DEFAULT is not its own loop but cv2.findHomography(q, im, cv2.RANSAC, th)
(model/robust_matching.py:214-216).  `cv_find_homography_ransac` restates the published algorithm
of that call (OpenCV 4, modules/calib3d: RANSACPointSetRegistrator::run, RANSACUpdateNumIters,
HomographyEstimatorCallback::computeError, HomographyRefineCallback, LMSolverImpl::run) from its
documentation and source as remembered; no OpenCV is installed and the reference holds no output
of it, so this part is PARITY UNPINNED.  Known departures: errors are evaluated in float64
(OpenCV: float32 points and errors); samples come from the caller (OpenCV: its own generator,
re-drawing samples with three collinear points).
"""
import numpy as np

__all__ = ["HomographyModel", "ransac", "random_partition", "ransac_num_iterations",
           "MIN_INLIERS_TH", "cv_update_num_iters", "cv_sq_error", "cv_lm_refine",
           "cv_find_homography_ransac"]

MIN_INLIERS_TH = 10   # model/robust_matching.py:13


class HomographyModel(object):
    MIN_POINTS_FOR_H = 4

    @staticmethod
    def fit(data):
        data = np.asarray(data, dtype=np.float64)
        src, dst = data[:, :2], data[:, 2:]
        n = data.shape[0]
        cm, cM = dst.mean(axis=0), src.mean(axis=0)
        sm, sM = np.abs(dst - cm).sum(axis=0), np.abs(src - cM).sum(axis=0)
        if (np.abs(sm) < np.finfo(np.float64).eps).any() or (np.abs(sM) < np.finfo(np.float64).eps).any():
            return None
        sm, sM = n / sm, n / sM
        inv_hnorm = np.array([[1.0 / sm[0], 0, cm[0]], [0, 1.0 / sm[1], cm[1]], [0, 0, 1]])
        hnorm2 = np.array([[sM[0], 0, -cM[0] * sM[0]], [0, sM[1], -cM[1] * sM[1]], [0, 0, 1]])
        x, y = (dst[:, 0] - cm[0]) * sm[0], (dst[:, 1] - cm[1]) * sm[1]
        X, Y = (src[:, 0] - cM[0]) * sM[0], (src[:, 1] - cM[1]) * sM[1]
        zeros, ones = np.zeros(n), np.ones(n)
        lx = np.stack([X, Y, ones, zeros, zeros, zeros, -x * X, -x * Y, -x], axis=1)
        ly = np.stack([zeros, zeros, zeros, X, Y, ones, -y * X, -y * Y, -y], axis=1)
        ltl = lx.T.dot(lx) + ly.T.dot(ly)
        w, v = np.linalg.eigh(ltl)
        h0 = v[:, 0].reshape(3, 3)
        h = inv_hnorm.dot(h0).dot(hnorm2)
        return h / h[2, 2]

    @staticmethod
    def get_error(data, model):
        data = np.asarray(data, dtype=np.float64)
        list_ones = np.ones((data.shape[0], 1))
        src_p = np.append(data[:, :2], list_ones, axis=1)
        dst_p = np.append(data[:, 2:], list_ones, axis=1)
        t = np.dot(model, src_p.T)
        for i in range(3):
            t[i] /= (t[2] + 0.0000000001)     # row 2 is rescaled last, as in the reference's loop
        return np.sqrt(np.sum((dst_p.T - t) ** 2, axis=0))


def random_partition(n, n_data, rng=np.random):
    all_idxs = np.arange(n_data)
    rng.shuffle(all_idxs)
    return all_idxs[:n], all_idxs[n:]


def ransac_num_iterations(max_iter, p, w, min_model_points=4):
    num = np.log(1 - p) / np.log(1 - (w ** min_model_points))
    if max_iter > 0:
        num = np.min([max_iter, int(num)])
    return int(num)


def ransac(data, model, min_model_points, max_iter, th, p=0.95, w=0.5, rng=np.random,
           partitions=None, num_it=None, min_inliers=MIN_INLIERS_TH, accept_mean_err=True,
           info=None):
    """model/robust_matching.py:291-365.  Test hooks that the reference does not have:
    `partitions` replaces random_partition with a list of (candidate_idxs, test_idxs) so that the
    same samples can be fed elsewhere; `num_it`, `min_inliers`, `accept_mean_err` override the
    iteration count and the acceptance rule of :360; `info` (a dict) receives the iteration whose
    model was kept."""
    if num_it is None:
        num_it = ransac_num_iterations(max_iter, p, w, min_model_points)
    best_fit, best_err, best_inlier_idxs = None, np.inf, []
    best_it = -1
    n = data.shape[0]
    for it in range(num_it):
        if partitions is not None:
            cand_idxs, test_idxs = partitions[it]
        else:
            cand_idxs, test_idxs = random_partition(min_model_points, n, rng)
        cand = data[cand_idxs]
        cand_model = model.fit(cand)
        if cand_model is None or not np.isfinite(cand_model).all():
            test_err = np.full(len(test_idxs), np.inf)
        else:
            test_err = np.array(model.get_error(data[test_idxs], cand_model))
        inlier_idxs = test_idxs[np.argwhere(test_err < th).flatten()]
        additional = data[inlier_idxs, :]
        if len(additional) > len(best_inlier_idxs) - min_model_points:
            current = np.concatenate((cand, additional))
            current_model = model.fit(current)
            if current_model is None or not np.isfinite(current_model).all():
                current_err = np.inf
            else:
                current_err = np.mean(model.get_error(current, current_model))
            best_fit, best_err = current_model, current_err
            best_inlier_idxs = np.concatenate((cand_idxs, inlier_idxs))
            best_it = it
        if len(best_inlier_idxs) == n:
            break
    if info is not None:
        info["best_it"] = best_it
    mask = np.zeros(n, dtype=int)
    if (best_err <= th or not accept_mean_err) and len(best_inlier_idxs) >= min_inliers \
            and best_fit is not None:
        mask[best_inlier_idxs] = 1
    else:
        best_fit = None
    return best_fit, mask, best_err


# ---------------------------------------------------------------------------------------------
# cv2.findHomography(..., cv2.RANSAC, th): the reference's default (model/robust_matching.py:214)
# ---------------------------------------------------------------------------------------------
CV_CONFIDENCE = 0.995
CV_MAX_ITERS = 2000
_DBL_MIN = np.finfo(np.float64).tiny
_DBL_EPS = np.finfo(np.float64).eps


def cv_update_num_iters(p, ep, model_points, max_iters):
    """RANSACUpdateNumIters: iterations after which a sample free of outliers has been drawn with
    probability p when a fraction ep of the data are outliers, never more than max_iters."""
    p = min(max(p, 0.0), 1.0)
    ep = min(max(ep, 0.0), 1.0)
    num = max(1.0 - p, _DBL_MIN)
    denom = 1.0 - (1.0 - ep) ** model_points
    if denom < _DBL_MIN:
        return 0
    num, denom = np.log(num), np.log(denom)
    if denom >= 0 or -num >= max_iters * (-denom):
        return max_iters
    return int(np.rint(num / denom))        # cvRound: to nearest, ties to even


def cv_sq_error(data, H):
    """HomographyEstimatorCallback::computeError: squared distance between the projected source
    point and the destination point (H[2,2] = 1)."""
    data = np.asarray(data, dtype=np.float64)
    h = np.asarray(H, dtype=np.float64).reshape(9)
    x, y, u, v = data[:, 0], data[:, 1], data[:, 2], data[:, 3]
    with np.errstate(all="ignore"):
        ww = 1.0 / (h[6] * x + h[7] * y + 1.0)
        dx = (h[0] * x + h[1] * y + h[2]) * ww - u
        dy = (h[3] * x + h[4] * y + h[5]) * ww - v
        return dx * dx + dy * dy


def _refine_residuals(h8, data, want_jac):
    x, y, u, v = data[:, 0], data[:, 1], data[:, 2], data[:, 3]
    ww = h8[6] * x + h8[7] * y + 1.0
    ww = np.where(np.abs(ww) > _DBL_EPS, 1.0 / np.where(ww == 0, 1.0, ww), 0.0)
    xi = (h8[0] * x + h8[1] * y + h8[2]) * ww
    yi = (h8[3] * x + h8[4] * y + h8[5]) * ww
    n = data.shape[0]
    r = np.empty(2 * n)
    r[0::2], r[1::2] = xi - u, yi - v
    if not want_jac:
        return r, None
    J = np.zeros((2 * n, 8))
    J[0::2, 0], J[0::2, 1], J[0::2, 2] = x * ww, y * ww, ww
    J[0::2, 6], J[0::2, 7] = -x * ww * xi, -y * ww * xi
    J[1::2, 3], J[1::2, 4], J[1::2, 5] = x * ww, y * ww, ww
    J[1::2, 6], J[1::2, 7] = -x * ww * yi, -y * ww * yi
    return r, J


def cv_lm_refine(H, data, max_iters=10):
    """The polish findHomography applies to the refitted model: Levenberg-Marquardt over the first
    eight entries of H on the inliers' reprojection residuals (HomographyRefineCallback,
    LMSolverImpl::run with maxIters = 10, epsx = epsf = DBL_EPSILON)."""
    data = np.asarray(data, dtype=np.float64)
    x = np.asarray(H, dtype=np.float64).reshape(9)[:8].copy()
    r, J = _refine_residuals(x, data, True)
    S = float(r.dot(r))
    A, v = J.T.dot(J), J.T.dot(r)
    D = np.diag(A).copy()
    r_lo, r_hi = 0.25, 0.75
    lam, lc = 1.0, 0.75
    it = 0
    while True:
        Ap = A + np.diag(lam * D)
        try:
            d = np.linalg.solve(Ap, v)
        except np.linalg.LinAlgError:
            d = np.linalg.lstsq(Ap, v, rcond=None)[0]
        xd = x - d
        rd, _ = _refine_residuals(xd, data, False)
        Sd = float(rd.dot(rd))
        dS = float(d.dot(2.0 * v - A.dot(d)))
        R = (S - Sd) / (dS if abs(dS) > _DBL_EPS else 1.0)
        if R > r_hi:
            lam *= 0.5
            if lam < lc:
                lam = 0.0
        elif R < r_lo:
            t = float(d.dot(v))
            nu = (Sd - S) / (t if abs(t) > _DBL_EPS else 1.0) + 2.0
            nu = min(max(nu, 2.0), 10.0)
            if lam == 0:
                inv = np.linalg.pinv(A)
                maxval = max(_DBL_EPS, float(np.abs(np.diag(inv)).max()))
                lam = lc = 1.0 / maxval
                nu *= 0.5
            lam *= nu
        if Sd < S:
            S, x = Sd, xd
            r, J = _refine_residuals(x, data, True)
            A, v = J.T.dot(J), J.T.dot(r)
        it += 1
        if not (it < max_iters and np.abs(d).max() >= _DBL_EPS and np.abs(r).max() >= _DBL_EPS):
            break
    return np.append(x, 1.0).reshape(3, 3)


def cv_find_homography_ransac(data, th, samples, max_iters=CV_MAX_ITERS, confidence=CV_CONFIDENCE,
                              polish=True, info=None):
    """cv2.findHomography(src, dst, cv2.RANSAC, th) on rows (x_src, y_src, x_dst, y_dst), with the
    four-point samples given: samples[it] are the pair indices of iteration `it`.

    The loop of RANSACPointSetRegistrator::run: every iteration fits the sample's homography,
    counts the pairs (all of them, the sample included) whose squared error is <= th^2, keeps the
    model when it has strictly more inliers than the best so far (and more than three), and then
    lowers the iteration limit to cv_update_num_iters(confidence, outlier fraction).  Afterwards
    findHomography refits over the kept model's inliers (normalised DLT) and polishes the result;
    the mask it returns is the kept minimal model's.  Returns (H or None, mask int [n]);
    info receives best_it and the number of iterations the loop ran.
    """
    data = np.asarray(data, dtype=np.float64)
    n = data.shape[0]
    mask = np.zeros(n, dtype=int)
    if info is not None:
        info["best_it"], info["iterations"] = -1, 0
    if n < 4:
        return None, mask
    niters = min(max_iters, len(samples))
    best_count, best_mask, best_it = 0, None, -1
    it = 0
    while it < niters:
        model = HomographyModel.fit(data[np.asarray(samples[it])])
        if model is not None and np.isfinite(model).all():
            inl = cv_sq_error(data, model) <= th * th
            good = int(inl.sum())
            if good > max(best_count, 3):
                best_count, best_mask, best_it = good, inl, it
                niters = cv_update_num_iters(confidence, (n - good) / float(n), 4, niters)
        it += 1
    if info is not None:
        info["best_it"], info["iterations"] = best_it, it
    if best_mask is None:
        return None, mask
    mask[best_mask] = 1
    H = HomographyModel.fit(data[best_mask])
    if H is None or not np.isfinite(H).all():
        return None, np.zeros(n, dtype=int)
    if polish:
        H = cv_lm_refine(H, data[best_mask])
    return H, mask
