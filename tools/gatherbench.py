#!/usr/bin/env python3
"""Host-side gather of per-image descriptor arrays into one page-locked buffer: how fast the
strategies available from Python are on this machine (the upload pipeline's first stage)."""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


def main():
    parts = [np.random.rand(1000, 128).astype(np.float32) for _ in range(1000)]
    pinned = torch.empty((1000000, 128), dtype=torch.float32).pin_memory()
    dst = pinned.numpy()
    dst[:] = 0

    def cp(d, s):
        np.copyto(d, s, casting="same_kind")

    def many(jobs):
        for d, s in jobs:
            np.copyto(d, s, casting="same_kind")

    jobs = [(dst[i * 1000:(i + 1) * 1000], p) for i, p in enumerate(parts)]
    for threads in (4, 8, 12, 16):
        pool = ThreadPoolExecutor(threads)
        best = [1e9, 1e9]
        for _ in range(5):
            t = time.perf_counter()
            fs = [pool.submit(cp, d, s) for d, s in jobs]
            for f in fs:
                f.result()
            t1 = time.perf_counter()
            fs = [pool.submit(many, jobs[w::threads]) for w in range(threads)]
            for f in fs:
                f.result()
            t2 = time.perf_counter()
            best = [min(best[0], t1 - t), min(best[1], t2 - t1)]
        print("%2d threads: one future per part %.1f ms | one batch per thread %.1f ms"
              % (threads, best[0] * 1e3, best[1] * 1e3), flush=True)
        pool.shutdown()
    t = time.perf_counter()
    many(jobs)
    print("single thread %.1f ms" % ((time.perf_counter() - t) * 1e3))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from image_classification_amd import ops
    rows = [(p, 0, p.shape[0]) for p in parts]
    for threads in (4, 8, 16):
        ops.GATHER_THREADS = threads
        best = 1e9
        for _ in range(5):
            t = time.perf_counter()
            ops._gather_rows(pinned, rows, 128)
            best = min(best, time.perf_counter() - t)
        print("native gather, %2d threads: %.1f ms" % (threads, best * 1e3), flush=True)
    big = np.concatenate(parts)
    t = time.perf_counter()
    np.copyto(dst, big)
    print("one 512 MB copy, single thread %.1f ms" % ((time.perf_counter() - t) * 1e3))


if __name__ == "__main__":
    main()
