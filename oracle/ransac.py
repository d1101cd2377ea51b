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
"""
import numpy as np

__all__ = ["HomographyModel", "ransac", "random_partition", "ransac_num_iterations",
           "MIN_INLIERS_TH"]

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
