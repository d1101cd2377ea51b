#!/usr/bin/env python3
"""One RobustMatcher.match call as the logo flow issues it (matching_flow.py:26: a query box of
about a thousand descriptors against a shortlist of candidate logos): time per call and where
it goes."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ImageDesc, RobustMatcher  # noqa: E402
from image_classification_amd.synthetic import sift_like  # noqa: E402

rng = np.random.default_rng(3)


def desc(n, seed):
    return ImageDesc(rng.random((n, 2)) * 500, sift_like(n, seed=seed))


query = desc(1000, 1)
cands = [desc(500, 10 + i) for i in range(20)]
# two candidates really contain a part of the query
for c in cands[:2]:
    src = rng.integers(0, 1000, 120)
    c.descriptors[:120] = query.descriptors[src] + rng.normal(0, 2, (120, 128)).astype(np.float32)
    c.keypoints[:120] = query.keypoints[src] * 1.1 + 20
rm = RobustMatcher(min_points=10, min_matches=8, ransac_th=3.0)
for _ in range(20):
    scores = rm.match(query, cands)
print("scores of the planted candidates %s, others max %.2f" % (scores[:2], max(scores[2:])))
t0 = time.perf_counter()
for _ in range(200):
    rm.match(query, cands)
print("RobustMatcher.match(1000-descriptor query, 20 candidates x 500): %.3f ms per call"
      % ((time.perf_counter() - t0) / 200 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    rm.match(query, cands)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(32)
