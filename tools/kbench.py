#!/usr/bin/env python3
"""Micro-benchmark of the distance / top-2 kernel: kbench.py N K [reps]  (several N,K pairs as
N1xK1 N2xK2 ...).  Prints the HIP-event kernel time and TFLOP/s (2*N*K*128)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402
from image_classification_amd.synthetic import sift_like_device  # noqa: E402


def run(n, k, reps=5):
    dev = torch.device("cuda", 0)
    x = sift_like_device(n, seed=1, device=dev)
    c = sift_like_device(k, seed=2, device=dev)
    cn = ops.codebook_aux(c)
    # The shader clock ramps with sustained load (tools/clockramp.py: a 0.25 ms search runs 12 %
    # slower in the first milliseconds after an idle period than 20 ms into a busy loop), so warm
    # up by time, not by count, and time enough calls to fill 20 ms.
    t_end = time.perf_counter() + 0.05
    calls = 0
    while time.perf_counter() < t_end or calls < 2:
        ops.assign_top1(x, c, cn, want_dist=False)
        torch.cuda.synchronize()
        calls += 1
    reps = max(reps, min(200, int(calls * 0.4)))
    ops.profile_begin(0)
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.assign_top1(x, c, cn, want_dist=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, launches = ops.profile_end(0)
    ms /= launches
    tf = 2.0 * n * k * 128 / (ms * 1e-3) / 1e12
    print("N=%9d K=%8d kernel %8.3f ms  %6.1f TFLOP/s (%4.1f%% of 157.3)  call wall %8.3f ms"
          % (n, k, ms, tf, 100 * tf / 157.3, wall * 1e3), flush=True)


if __name__ == "__main__":
    for arg in sys.argv[1:]:
        n, k = arg.lower().split("x")
        run(int(n), int(k))
