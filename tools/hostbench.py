#!/usr/bin/env python3
"""End-to-end timing of the numpy-in / numpy-out API (PCIe included): hostbench.py [N] [K]."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd import vq  # noqa: E402
from image_classification_amd.synthetic import sift_like  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000
x = sift_like(n, n_modes=2048, seed=3)
c = sift_like(k, n_modes=2048, seed=4).astype(np.float64)
vq(x[:1000], c)  # library load, handle, codebook upload
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    code, dist = vq(x, c, check_finite=False)
    t1 = time.perf_counter()
    print("vq(%d x %d) numpy in/out: %.1f ms -> %.3g assignments/s (PCIe included)"
          % (n, k, (t1 - t0) * 1e3, n / (t1 - t0)), flush=True)
t0 = time.perf_counter()
xd = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
print("pageable H2D of %.0f MB: %.1f ms" % (x.nbytes / 1e6, (time.perf_counter() - t0) * 1e3))
xp = torch.from_numpy(x).pin_memory()
t0 = time.perf_counter()
xd = xp.cuda(non_blocking=True)
torch.cuda.synchronize()
print("pinned   H2D of %.0f MB: %.1f ms" % (x.nbytes / 1e6, (time.perf_counter() - t0) * 1e3))
t0 = time.perf_counter()
ok = np.isfinite(x).all()
print("np.isfinite check: %.1f ms" % ((time.perf_counter() - t0) * 1e3))

# per-image calls, as the reference's flow issues them (one image's descriptors at a time)
from image_classification_amd import KNNScikit  # noqa: E402
for rows in (1000, 4000):
    xi = x[:rows]
    vq(xi, c)
    t0 = time.perf_counter()
    for _ in range(50):
        vq(xi, c)
    print("vq(%d x %d) per-image call: %.3f ms" % (rows, k, (time.perf_counter() - t0) / 50 * 1e3))
    frozen = c.copy()
    frozen.setflags(write=False)     # recognised without hashing its 10 MB on every call
    vq(xi, frozen)
    t0 = time.perf_counter()
    for _ in range(50):
        vq(xi, frozen)
    print("vq(%d x %d) per-image call, read-only vocabulary: %.3f ms"
          % (rows, k, (time.perf_counter() - t0) / 50 * 1e3))
hist = np.random.default_rng(0).random((2000, k))
hist /= hist.sum(1, keepdims=True)
knn = KNNScikit(3, metric="manhattan")
knn.train(hist, list(range(2000)))
knn.get_neighbors(hist[:1], 3)
t0 = time.perf_counter()
for _ in range(50):
    knn.get_neighbors(hist[:1], 3)
print("KNNScikit(manhattan).get_neighbors(1 x %d vs 2000 logos): %.3f ms"
      % (k, (time.perf_counter() - t0) / 50 * 1e3))

# BagOfWords.extract over 1000 images x 1000 descriptors held as separate numpy arrays
from image_classification_amd import BagOfWords  # noqa: E402


class _Pre(object):
    def __init__(self, per_image):
        self.per_image = per_image

    def extract(self, image_list):
        return [(self.per_image[i][0], self.per_image[i][1]) for i in image_list]


n_img = n // 1000
kps = [(0.0, 0.0)] * 1000
per_image = [(kps, x[i * 1000:(i + 1) * 1000].copy()) for i in range(n_img)]
bow = BagOfWords(_Pre(per_image), vocab=c)
bow.extract(list(range(4)))
for rep in range(3):
    t0 = time.perf_counter()
    descs, hists = bow.extract(list(range(n_img)))
    dt = time.perf_counter() - t0
    print("BagOfWords.extract(%d images x 1000 descriptors, %d words): %.1f ms -> %.3g assignments/s"
          % (n_img, k, dt * 1e3, n_img * 1000 / dt), flush=True)
# the same work as three plain phases (search of all rows, then all histograms, then the staged
# download), alternating with the pipeline in one process
b_ = bow._b()
for rep in range(3):
    t0 = time.perf_counter()
    ids = bow._describe(list(range(n_img)))
    codes_, offs_ = bow._assign_all(ids)
    h3 = b_.to_host(b_.hist(codes_, b_.i64(offs_), len(bow.vocab), "l1", 0.5))
    t_three = time.perf_counter() - t0
    t0 = time.perf_counter()
    _, hp = bow.extract(list(range(n_img)))
    t_pipe = time.perf_counter() - t0
    assert np.array_equal(h3, hp)
    print("  same process: three phases %.1f ms | pipeline %.1f ms" % (t_three * 1e3, t_pipe * 1e3),
          flush=True)
t0 = time.perf_counter()
image_descs = bow._describe(list(range(n_img)))
t1 = time.perf_counter()
codes, offsets = bow._assign_all(image_descs)
torch.cuda.synchronize()
t2 = time.perf_counter()
hd = bow._b().hist(codes, bow._b().i64(offsets), len(bow.vocab), "l1", 0.5)
torch.cuda.synchronize()
t3 = time.perf_counter()
hh = hd.cpu().numpy()
t4 = time.perf_counter()
print("  ImageDesc objects %.1f ms | gather + upload + search %.1f ms | histogram %.1f ms | "
      "download of %d MB %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3,
                                      hh.nbytes >> 20, (t4 - t3) * 1e3))
