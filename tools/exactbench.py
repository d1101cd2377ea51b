#!/usr/bin/env python3
"""Timing of the direct (no-expansion) search kernel: exactbench.py N K."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402
from image_classification_amd.synthetic import sift_like_device  # noqa: E402

n, k = int(sys.argv[1]), int(sys.argv[2])
x = sift_like_device(n, seed=1, device="cuda")
c = sift_like_device(k, seed=2, device="cuda")
ops.assign_top2_exact(x[:1000], c)
torch.cuda.synchronize()
t0 = time.perf_counter()
ops.assign_top2_exact(x, c)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("direct search %d x %d: %.1f ms, %.1f T pair-components/s (vector-ALU bound: 2 ops each)"
      % (n, k, dt * 1e3, n * k * 128 / dt / 1e12))
