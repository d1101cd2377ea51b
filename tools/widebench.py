#!/usr/bin/env python3
"""Search of rows wider than 128 columns: widebench.py N K D [...]  (kernel time by HIP events)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402


def run(n, k, d):
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    centres = torch.randn((64, d), device=dev, generator=g) * 3
    x = centres[torch.randint(0, 64, (n,), device=dev, generator=g)] + torch.randn((n, d), device=dev, generator=g)
    c = centres[torch.randint(0, 64, (k,), device=dev, generator=g)] + torch.randn((k, d), device=dev, generator=g)
    aux = ops.codebook_aux(c)
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        ops.assign_top1(x, c, aux, want_dist=False)
        torch.cuda.synchronize()
    ops.profile_begin(0)
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        ops.assign_top1(x, c, aux, want_dist=False)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, launches = ops.profile_end(0)
    ms /= launches
    dpad = (d + 127) // 128 * 128
    tf = 2.0 * n * k * d / (ms * 1e-3) / 1e12
    print("N=%8d K=%6d D=%5d (padded %5d): search kernel %8.3f ms  %6.1f TFLOP/s of useful columns "
          "(%4.1f%% of 157.3)  call wall %8.3f ms" % (n, k, d, dpad, ms, tf, 100 * tf / 157.3, wall * 1e3),
          flush=True)


if __name__ == "__main__":
    a = sys.argv[1:]
    for i in range(0, len(a), 3):
        run(int(a[i]), int(a[i + 1]), int(a[i + 2]))
