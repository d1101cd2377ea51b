#!/usr/bin/env python3
"""The ranking gate's worst case: every row ranks the codewords the same way and the codebook is
stored farthest-first, so every candidate of every tile improves every lane's running top-2.
Prints the search kernel's time for that order, for the reverse (nearest-first: the gate's best
case) and for a random order of the same codewords."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import ops  # noqa: E402
from image_classification_amd.synthetic import sift_like_device  # noqa: E402

n, k = 1_000_000, 10_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)
centre = sift_like_device(1, seed=9, device=dev)
x = centre + 0.5 * torch.randn((n, 128), device=dev, generator=g)      # one tight cluster
c = sift_like_device(k, n_modes=2048, seed=4, device=dev)
dist = torch.cdist(c, centre).reshape(-1)
orders = {"farthest first (worst)": torch.argsort(dist, descending=True),
          "nearest first (best)": torch.argsort(dist),
          "random": torch.randperm(k, device=dev, generator=g)}
ref = None
for name, perm in orders.items():
    cb = c[perm].contiguous()
    aux = ops.codebook_aux(cb)
    for _ in range(2):
        ops.assign_top1(x, cb, aux, want_dist=False)
    torch.cuda.synchronize()
    ops.profile_begin(0)
    for _ in range(5):
        idx, _ = ops.assign_top1(x, cb, aux, want_dist=False)
    torch.cuda.synchronize()
    ms, launches = ops.profile_end(0)
    ms /= launches
    word = perm[idx.long()]                       # same codeword whatever the storage order
    if ref is None:
        ref = word
    tf = 2.0 * n * k * 128 / (ms * 1e-3) / 1e12
    # rows may differ only where two codewords are at exactly the same float32 distance (the
    # lower INDEX wins, and the index depends on the storage order)
    print("%-24s kernel %7.3f ms  %6.1f TFLOP/s (%4.1f%% of 157.3)  rows with another codeword: %d"
          % (name, ms, tf, 100 * tf / 157.3, int((word != ref).sum())), flush=True)
