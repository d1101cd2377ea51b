#!/usr/bin/env python3
"""Does the time of a short search depend on what the GPU did just before?  clockramp.py [N K]

Times the search kernel (HIP events around it) in blocks of back-to-back calls after an idle
period, and right after a long stretch of heavy work: the shader clock of this part ramps with
sustained load, so a 0.25 ms kernel measured cold runs at a lower clock than inside a busy loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402
from image_classification_amd.synthetic import sift_like_device  # noqa: E402


def block(x, c, aux, reps):
    ops.profile_begin(0)
    for _ in range(reps):
        ops.assign_top1(x, c, aux, want_dist=False)
    torch.cuda.synchronize()
    ms, launches = ops.profile_end(0)
    return ms / launches


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dev = torch.device("cuda", 0)
    x = sift_like_device(n, seed=1, device=dev)
    c = sift_like_device(k, seed=2, device=dev)
    aux = ops.codebook_aux(c)
    bx = sift_like_device(1_000_000, seed=3, device=dev)
    bc = sift_like_device(10_000, seed=4, device=dev)
    baux = ops.codebook_aux(bc)
    block(x, c, aux, 3)
    flops = 2.0 * n * k * 128
    pct = lambda ms: 100 * flops / (ms * 1e-3) / 157.3e12
    time.sleep(1.0)
    print("after 1 s idle, blocks of 20 back-to-back calls:")
    for i in range(10):
        ms = block(x, c, aux, 20)
        print("   block %2d: kernel %.4f ms  %.1f %%" % (i, ms, pct(ms)))
    for busy in (5, 20):
        for _ in range(busy):
            ops.assign_top1(bx, bc, baux, want_dist=False)   # 18 ms each
        ms = block(x, c, aux, 5)
        print("right after %3d ms of the 1M x 10k search: kernel %.4f ms  %.1f %%" % (busy * 18, ms, pct(ms)))
    time.sleep(0.2)
    ms = block(x, c, aux, 1)
    print("one call after 0.2 s idle: kernel %.4f ms  %.1f %%" % (ms, pct(ms)))


if __name__ == "__main__":
    main()
