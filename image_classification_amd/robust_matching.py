"""`RobustMatcher` with the reference's interface; its descriptor-matching stage runs on the device.

Mirrors model/robust_matching.py:16-96 and :177-194.  In `match(query, image_list)` the reference
fits a brute-force 2-NN index on the QUERY descriptors and, for every candidate image with more
than `min_points` descriptors, searches the candidate's descriptors in it and keeps row `idx`
when `dist[idx, 0] < dist[idx, 1] * knn_thresh` (:73-84, a Python loop over rows).  Here the
descriptors of all candidates are concatenated, one device call returns both neighbours and the
ratio mask for every row, and the per-candidate match lists are cut out of that.

The geometric verification that follows (`_score_matches` -> homography RANSAC through
cv2.findHomography, :156-226) is not part of the accelerated path: pass `homography_fn`, a
callable `(q_points [m,2], im_points [m,2]) -> (retval, mask)`, or have cv2 importable.
"""
from logging import getLogger

import numpy as np

from .backend import default_backend

__all__ = ["RobustMatcher"]

logger = getLogger("affine.detection.model.robust_matching")

CV_RANSAC = 8  # value of cv2.RANSAC, the reference's default `ransac_algorithm`


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
            parts.append(np.asarray(image_list[n].descriptors, dtype=np.float32))
            rows.append(rows[-1] + parts[-1].shape[0])
        allx = parts[0] if len(parts) == 1 else np.concatenate(parts)
        i1, _d1, _i2, _d2, mask = b.assign_top2_np(allx, b.codebook(query.descriptors),
                                                ratio=self.knn_thresh)
        i1 = i1.cpu().numpy()
        mask = mask.cpu().numpy().astype(bool)
        for slot, n in enumerate(use):
            lo, hi = rows[slot], rows[slot + 1]
            m = mask[lo:hi]
            out[n] = (i1[lo:hi][m].astype(np.int64), np.nonzero(m)[0].astype(np.int64))
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
            logger.debug("%d inliers out of %d features in query %d features in model-img",
                         inliers, num_feat_query, img.desc_rows)
        logger.debug("Scores obtained in robust matching %s", img_scores)
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
        if self.homography_fn is not None:
            return self.homography_fn(q_points, im_points)
        try:
            import cv2
        except ImportError:
            raise NotImplementedError(
                "homography estimation needs cv2 or a `homography_fn`; the device path covers the "
                "descriptor matching stage (match_descriptors) only")
        return cv2.findHomography(np.asarray(q_points), np.asarray(im_points), cv2.RANSAC,
                                  self.ransac_th)

    def predict(self, scores, labels):
        """Label of the best score if it reaches `accept_th`, else -1."""
        assert len(scores) == len(labels), "Labels and scores have different size"
        img = -1
        scores = [0 if scr > 1 else scr for scr in scores]
        if np.max(scores) >= self.accept_th:
            img = labels[np.argmax(scores)]
            logger.info("Label %s found!, score %.2f", img, np.max(scores))
        return img
