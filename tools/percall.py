#!/usr/bin/env python3
"""Where a small vq() call spends its time (the reference's flow calls vq once per image):
cProfile over 300 calls of 1000 x 128 rows against a 10 000-word float64 vocabulary."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import vq  # noqa: E402
from image_classification_amd.synthetic import sift_like  # noqa: E402

x = sift_like(1000, seed=1)
c = sift_like(10000, seed=2).astype(np.float64)
for _ in range(20):
    vq(x, c)
t0 = time.perf_counter()
for _ in range(300):
    vq(x, c)
print("vq(1000 x 10000): %.3f ms per call" % ((time.perf_counter() - t0) / 300 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    vq(x, c)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)

# the reference's per-image step: Step("feat_ext", lm.bow, 'extract') on one image
# (vision/logo_recognition/matching_flow.py:22): BagOfWords.extract([image]) -> (descs, hist[1, K])
from image_classification_amd import BagOfWords  # noqa: E402


class _Pre(object):
    def extract(self, image_list):
        return [([(0.0, 0.0)] * 1000, x) for _ in image_list]


bow = BagOfWords(_Pre(), vocab=c)
for _ in range(20):
    bow.extract([0])
t0 = time.perf_counter()
for _ in range(300):
    bow.extract([0])
print("BagOfWords.extract([one image of 1000 descriptors]) x 10000 words: %.3f ms per call"
      % ((time.perf_counter() - t0) / 300 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    bow.extract([0])
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(30)
