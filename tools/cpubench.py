#!/usr/bin/env python3
"""CPU baselines beside the GPU path on the same box, for the configuration the reference can run
itself (100 000 descriptors x 1000-word codebook x 128-d): scipy.cluster.vq.vq with the float64
codebook the reference stores (model/features.py:1166) and with a float32 one, the batched
euclidean_distances + argmin of compute_vocab_features (vision/utils/scene_functions.py:375-392)
through scikit-learn where installed, each on all threads and on one; then this package's vq()
(numpy in, numpy out) and the resident search.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image_classification_amd.synthetic import sift_like  # noqa: E402


def best_of(fn, reps=3):
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    n, k = 100_000, 1000
    x = sift_like(n, seed=1)
    c32 = sift_like(k, seed=2)
    c64 = c32.astype(np.float64)
    out = {"rows": n, "codebook": k, "dim": 128}
    from threadpoolctl import threadpool_info, threadpool_limits
    out["blas"] = sorted({(p.get("internal_api"), p.get("num_threads")) for p in threadpool_info()
                          if p.get("user_api") == "blas"})
    cases = {}
    try:
        from scipy.cluster.vq import vq as scipy_vq
        cases["scipy_vq_f64_codebook"] = lambda: scipy_vq(x, c64)
        cases["scipy_vq_f32_codebook"] = lambda: scipy_vq(x, c32)
    except ImportError:
        pass
    try:
        from sklearn.metrics.pairwise import euclidean_distances

        def cvf():
            codes = np.zeros(n, dtype=np.int64)
            for lo in range(0, n, k):
                codes[lo:lo + k] = np.argmin(euclidean_distances(x[lo:lo + k], c64, squared=True), axis=1)
            return codes
        cases["sklearn_compute_vocab_features"] = cvf
    except ImportError:
        pass
    for name, fn in cases.items():
        fn()
        out[name + "_all_threads_rows_per_s"] = n / best_of(fn)
        with threadpool_limits(limits=1):
            out[name + "_one_thread_rows_per_s"] = n / best_of(fn, reps=1)

    import torch
    from image_classification_amd import ops, vq
    code, dist = vq(x, c64)
    if "scipy_vq_f64_codebook" in cases:
        ref_code, ref_dist = cases["scipy_vq_f64_codebook"]()
        out["gpu_vs_scipy_index_agreement"] = float(np.mean(code == ref_code))
        out["gpu_vs_scipy_max_rel_dist_err"] = float(np.max(np.abs(dist - ref_dist) / np.maximum(ref_dist, 1e-30)))
    out["gpu_vq_numpy_in_out_rows_per_s"] = n / best_of(lambda: vq(x, c64), reps=5)
    xd, cd = torch.from_numpy(x).cuda(), torch.from_numpy(c32).cuda()
    cn = ops.codebook_aux(cd)

    def resident():
        ops.assign_top1(xd, cd, cn)
        torch.cuda.synchronize()
    resident()
    out["gpu_resident_rows_per_s"] = n / best_of(resident, reps=10)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
