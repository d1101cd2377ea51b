"""`RobustMatcher` with the reference's interface; matching and verification run on the device.

Mirrors model/robust_matching.py:16-96 and :156-226.  In `match(query, image_list)` the reference
fits a brute-force 2-NN index on the QUERY descriptors and, for every candidate image with more
than `min_points` descriptors, searches the candidate's descriptors in it, keeps row `idx` when
`dist[idx, 0] < dist[idx, 1] * knn_thresh` (:73-84, a Python loop over rows), gathers the matched
keypoints and scores them with a homography RANSAC (:85-95, :156-226).  Here the descriptors of
all candidates are concatenated and the whole query is four device calls: both neighbours and
the ratio mask for every row, the per-candidate match lists with their keypoint pairs, and one
batched RANSAC over all candidates.

RANSAC variants (`ransac_algorithm`): any value other than cv2.RANSAC (8) selects the reference's
own `ransac()` (:291-365): int(log(1-prob)/log(1-inlier_r^4)) iterations capped by
`ransac_max_iter`, at least MIN_INLIERS_TH = 10 inliers and a mean reprojection error within
`ransac_th`.  The default 8 stands for cv2.findHomography(..., cv2.RANSAC, ransac_th); OpenCV is
not used here, the device kernels follow that function's published loop instead (inliers at
squared error <= ransac_th^2, a model kept when it has strictly more inliers than every earlier
one, the iteration limit lowered from 2000 at confidence 0.995 with every kept model, refit over
the inliers and a Levenberg-Marquardt polish; the mask is the kept minimal model's).  No OpenCV
output exists to pin this against.  Samples come from a counter-based generator
on the device, seeded per call from numpy's global generator (so `np.random.seed` makes a run
repeatable, as it does for the reference's `np.random.shuffle`).  A `homography_fn`
callable `(q_points [m,2], im_points [m,2]) -> (H, mask)` replaces the device RANSAC per
candidate when given.
"""
from logging import getLogger

import numpy as np
import torch

from .backend import default_backend

__all__ = ["RobustMatcher"]

logger = getLogger("affine.detection.model.robust_matching")

CV_RANSAC = 8  # value of cv2.RANSAC, the reference's default `ransac_algorithm`
MIN_INLIERS_TH = 10        # model/robust_matching.py:13
CV_RANSAC_MAX_ITERS = 2000  # OpenCV's default maxIters of findHomography


def ransac_num_iterations(max_iter, p, w, min_model_points=4):
    """Iteration count of the reference's ransac() (model/robust_matching.py:320-322)."""
    num = np.log(1 - p) / np.log(1 - (w ** min_model_points))
    if max_iter > 0:
        num = np.min([max_iter, int(num)])
    return int(num)


class RobustMatcher(object):

    def __init__(self, min_points, min_matches, ransac_th=3, accept_th=0.3, knn_thresh=0.8,
                 ransac_algorithm=CV_RANSAC, ransac_max_iter=100, ransac_prob=0.95, inlier_r=0.5,
                 debug=False, homography_fn=None, backend=None):
        self.min_points = min_points
        self.min_matches = min_matches
        self.ransac_th = ransac_th
        self.accept_th = accept_th
        self.knn_thresh = knn_thresh
        self.debug = debug
        self.ransac_algorithm = ransac_algorithm
        self.ransac_max_iter = ransac_max_iter
        self.ransac_prob = ransac_prob
        self.inlier_r = inlier_r
        self.homography_fn = homography_fn
        self._backend = backend

    def _b(self):
        return self._backend or default_backend()

    def match_descriptors(self, query, image_list):
        """Ratio-test matches of every candidate against the query.

        Returns a list with one `(matches_q, matches_im)` pair of int64 arrays per candidate:
        `matches_im` are the candidate's descriptor rows that passed the ratio test and
        `matches_q` the query rows they matched.  Candidates with `desc_rows <= min_points` give
        two empty arrays (the reference skips them, model/robust_matching.py:79).
        """
        empty = (np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64))
        out = [empty] * len(image_list)
        use = [n for n, img in enumerate(image_list) if img.desc_rows > self.min_points]
        if not use:
            return out
        nq = query.desc_rows
        if nq < 2:
            raise ValueError("Expected n_neighbors <= n_samples_fit, but n_neighbors = 2, "
                             "n_samples_fit = %d" % nq)
        b = self._b()
        rows = [0]
        parts = []
        for n in use:
            parts.append(np.asarray(image_list[n].descriptors))
            rows.append(rows[-1] + parts[-1].shape[0])
        i1, _d1, _i2, _d2, mask = b.assign_top2_parts(parts, b.codebook(query.descriptors),
                                                   ratio=self.knn_thresh)
        i1 = i1.cpu().numpy()
        mask = mask.cpu().numpy().astype(bool)
        for slot, n in enumerate(use):
            lo, hi = rows[slot], rows[slot + 1]
            m = mask[lo:hi]
            out[n] = (i1[lo:hi][m].astype(np.int64), np.nonzero(m)[0].astype(np.int64))
        return out

    def _ransac_params(self):
        """(iteration limit, fewest inliers, mean-error rule, OpenCV's loop rules)"""
        if self.ransac_algorithm == CV_RANSAC:
            return CV_RANSAC_MAX_ITERS, 4, False, True
        return (ransac_num_iterations(self.ransac_max_iter, self.ransac_prob, self.inlier_r),
                MIN_INLIERS_TH, True, False)

    def match_and_verify(self, query, image_list, want_matches=True):
        """The whole of `match` on the device.  Returns one dict per candidate with `n_matches`,
        `inliers` and -- unless `want_matches` is false, which spares three downloads --
        `matches_q`, `matches_im` (int64 rows) and `mask` (uint8 per match)."""
        empty_i = np.empty(0, dtype=np.int64)
        empty_m = np.empty(0, dtype=np.uint8)
        out = [dict(n_matches=0, inliers=0, matches_q=empty_i, matches_im=empty_i, mask=empty_m)
               for _ in image_list]
        use = [n for n, img in enumerate(image_list) if img.desc_rows > self.min_points]
        if not use:
            return out
        if query.desc_rows < 2:
            raise ValueError("Expected n_neighbors <= n_samples_fit, but n_neighbors = 2, "
                             "n_samples_fit = %d" % query.desc_rows)
        b = self._b()
        rows = [0]
        parts, kps = [], []
        for n in use:
            parts.append(np.asarray(image_list[n].descriptors))
            kps.append(np.asarray(image_list[n].keypoints, dtype=np.float64).reshape(-1, 2))
            rows.append(rows[-1] + parts[-1].shape[0])
        i1, _d1, _i2, _d2, mask = b.assign_top2_parts(parts, b.codebook(query.descriptors),
                                                   ratio=self.knn_thresh)
        # the keypoints of both sides and the candidates' row offsets go up in ONE copy (a copy
        # from pageable memory blocks the caller until it has happened; three of them cost more
        # than the kernels of a per-box call)
        kq = np.asarray(query.keypoints, dtype=np.float64).reshape(-1, 2)
        n_q, n_r = kq.shape[0], rows[-1]
        host = np.empty(2 * n_q + 2 * n_r + len(rows), dtype=np.float64)
        host[:2 * n_q] = kq.reshape(-1)
        at = 2 * n_q
        for kp in kps:
            host[at:at + kp.size] = kp.reshape(-1)
            at += kp.size
        host[at:] = rows               # exact: row counts are far below 2^53
        dev = b.f64(host)
        kp_q = dev[:2 * n_q].view(n_q, 2)
        kp_rows = dev[2 * n_q:2 * n_q + 2 * n_r].view(n_r, 2)
        row_off = dev[2 * n_q + 2 * n_r:].to(torch.int64)
        out_off, match_q, match_im, pts = b.compact_matches(mask, i1, row_off, kp_q, kp_rows)
        n_iter, min_inliers, mean_rule, cv_rules = self._ransac_params()
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        _H, n_inl, _err, _it, in_mask = b.ransac_homography(
            pts, out_off, n_iter, float(self.ransac_th), None, seed, min_inliers, mean_rule,
            cv_rules)
        # offsets and inlier counts come down together
        small = torch.cat([out_off, n_inl.to(torch.int64)]).cpu().numpy()
        out_off, n_inl = small[:len(use) + 1], small[len(use) + 1:]
        if want_matches:
            total = int(out_off[-1])
            match_q = match_q[:total].cpu().numpy().astype(np.int64)
            match_im = match_im[:total].cpu().numpy().astype(np.int64)
            in_mask = in_mask[:total].cpu().numpy()
        for slot, n in enumerate(use):
            lo, hi = int(out_off[slot]), int(out_off[slot + 1])
            verified = (hi - lo) > self.min_matches
            res = dict(n_matches=hi - lo, inliers=int(n_inl[slot]) if verified else 0,
                       matches_q=empty_i, matches_im=empty_i, mask=empty_m)
            if want_matches:
                res.update(matches_q=match_q[lo:hi], matches_im=match_im[lo:hi],
                           mask=in_mask[lo:hi] if verified else empty_m)
            out[n] = res
        return out

    def match(self, query, image_list):
        """Score of the query against each candidate: inliers / (number of matches or 1)."""
        if self.debug:
            self.query_kpts = []
            self.imgs_kpts = []
            self.mask_list = []
        num_feat_query = query.desc_rows
        if num_feat_query < self.min_points:
            return [0] * len(image_list)
        if self.homography_fn is not None:
            return self._match_with_callable(query, image_list)
        img_scores = []
        # (the match lists are only looked at in debug mode)
        for img, res in zip(image_list, self.match_and_verify(query, image_list, self.debug)):
            if self.debug:
                self.query_kpts.append(query.keypoints[res["matches_q"]])
                self.imgs_kpts.append(img.keypoints[res["matches_im"]])
                self.mask_list.append(res["mask"].reshape(-1, 1) if len(res["mask"]) else [])
            img_scores.append(float(res["inliers"]) / (res["n_matches"] or 1))
            logger.debug("%d inliers out of %d features in query %d features in model-img",
                         res["inliers"], num_feat_query, img.desc_rows)
        logger.debug("Scores obtained in robust matching %s", img_scores)
        return img_scores

    def _match_with_callable(self, query, image_list):
        matches = self.match_descriptors(query, image_list)
        img_scores = []
        for img, (matches_q, matches_im) in zip(image_list, matches):
            inliers = 0
            mask = []
            if img.desc_rows > self.min_points:
                inliers, mask = self._score_matches(query.keypoints[matches_q],
                                                    img.keypoints[matches_im])
            if self.debug:
                self.query_kpts.append(query.keypoints[matches_q])
                self.imgs_kpts.append(img.keypoints[matches_im])
                self.mask_list.append(mask)
            img_scores.append(float(inliers) / (len(matches_im) or 1))
        return img_scores

    def _score_matches(self, q_points, im_points):
        assert q_points.shape == im_points.shape, "different number of points to match"
        mask = []
        num_inliers = 0
        if len(q_points) > self.min_matches:
            _retval, mask = self._robust_h_estimation(q_points, im_points,
                                                      opt=self.ransac_algorithm)
            num_inliers = np.sum(mask)
        return num_inliers, mask

    def _robust_h_estimation(self, q_points, im_points, opt=CV_RANSAC):
        """(homography or None, inlier mask [m, 1]) for one set of matched points."""
        if self.homography_fn is not None:
            return self.homography_fn(q_points, im_points)
        b = self._b()
        data = np.ascontiguousarray(np.append(np.asarray(q_points, dtype=np.float64).reshape(-1, 2),
                                              np.asarray(im_points, dtype=np.float64).reshape(-1, 2),
                                              axis=1))
        saved, self.ransac_algorithm = self.ransac_algorithm, opt
        try:
            n_iter, min_inliers, mean_rule, cv_rules = self._ransac_params()
        finally:
            self.ransac_algorithm = saved
        seed = int(np.random.randint(0, 2 ** 31 - 1))
        H, n_inl, _err, _it, mask = b.ransac_homography(
            b.f64(data), b.i64([0, data.shape[0]]), n_iter, float(self.ransac_th), None, seed,
            min_inliers, mean_rule, cv_rules)
        homography = H[0].cpu().numpy() if int(n_inl[0]) > 0 else None
        return homography, mask.cpu().numpy().reshape(-1, 1)

    def inlier_pairs(self):
        """Debug mode, after `match`: per candidate a float array [m, 4] of the inlier matches as
        (x_query, y_query, x_candidate, y_candidate) -- what the reference's `draw_matches`
        (model/robust_matching.py:98-155) draws as circles joined by lines.  That method is image
        I/O through OpenCV and is not part of this package; it reads `query_kpts`, `imgs_kpts` and
        `mask_list`, which `match` fills here under the same names and shapes."""
        assert self.debug is True, "You need to be in debug mode"
        for inst in ["query_kpts", "imgs_kpts", "mask_list"]:
            assert hasattr(self, inst), "You need to run match function first"
        out = []
        for q, im, mask in zip(self.query_kpts, self.imgs_kpts, self.mask_list):
            if len(mask) == 0:
                out.append(np.empty((0, 4), dtype=np.float64))
                continue
            keep = np.asarray(mask).reshape(-1) == 1
            out.append(np.concatenate([np.asarray(q, dtype=np.float64)[keep],
                                       np.asarray(im, dtype=np.float64)[keep]], axis=1))
        return out

    def predict(self, scores, labels):
        """Label of the best score if it reaches `accept_th`, else -1."""
        assert len(scores) == len(labels), "Labels and scores have different size"
        img = -1
        scores = [0 if scr > 1 else scr for scr in scores]
        if np.max(scores) >= self.accept_th:
            img = labels[np.argmax(scores)]
            logger.info("Label %s found!, score %.2f", img, np.max(scores))
        return img
