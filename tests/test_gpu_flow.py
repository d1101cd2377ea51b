"""The threading / forking behaviour the reference's flow engine relies on.

model/mlflow/flow.py:115-116 runs every Step on its own thread and calls
`getattr(actor, activity)(*inputs)`; model/mlflow/parallel_flow.py:50-53 forks worker processes
after the model objects have been built.  The library handle is per (process, device), created on
first use and guarded by a mutex, so both patterns must give the same answers as a serial run.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle
from image_classification_amd import BagOfWords, ImageDesc, KNNScikit, RobustMatcher, vq
from image_classification_amd.synthetic import sift_like

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _FeatExt(object):
    """Stands for the SURF extractor: image 'paths' are seeds of synthetic descriptors."""
    def extract(self, image_list):
        out = []
        for seed in image_list:
            d = sift_like(200 + 10 * seed, seed=seed)
            out.append(([(float(i % 40), float(i // 40)) for i in range(len(d))], d))
        return out


def _step(actor, activity, *inputs):
    return getattr(actor, activity)(*inputs)


def test_steps_on_threads_match_serial_run():
    vocab = sift_like(256, seed=500).astype(np.float64)
    bow = BagOfWords(_FeatExt(), vocab=vocab)
    images = list(range(1, 9))
    descs, hists = bow.extract(images)
    knn = KNNScikit(3, metric="manhattan")
    knn.train(hists, list(range(8)))
    rm = RobustMatcher(min_points=10, min_matches=8, ransac_algorithm=0)
    serial_nb = knn.get_neighbors(hists[:2], 3, dist=False)
    np.random.seed(1)
    serial_scores = rm.match(descs[0], descs[1:4])
    with ThreadPoolExecutor(max_workers=3) as pool:          # one thread per step, as Flow does
        f1 = pool.submit(_step, bow, "extract", images)
        descs_t, hists_t = f1.result()
        f2 = pool.submit(_step, knn, "get_neighbors", hists_t[:2], 3, False)
        np.random.seed(1)
        f3 = pool.submit(_step, rm, "match", descs_t[0], descs_t[1:4])
        assert np.array_equal(f2.result(), serial_nb)
        assert f3.result() == serial_scores
    assert np.array_equal(hists_t, hists)
    # histogram of every image against the oracle (bit-exact L1 rule)
    for seed, h in zip(images, hists):
        code, _ = oracle.vq(sift_like(200 + 10 * seed, seed=seed), vocab)
        assert np.array_equal(h, oracle.build_histogram([code], 256)[0])


def test_concurrent_calls_on_one_handle():
    c = sift_like(700, seed=77).astype(np.float64)
    xs = [sift_like(3000 + 97 * i, seed=i) for i in range(8)]
    serial = [vq(x, c) for x in xs]
    with ThreadPoolExecutor(max_workers=8) as pool:
        for _ in range(3):
            got = list(pool.map(lambda x: vq(x, c), xs))
            for (gc, gd), (sc, sd) in zip(got, serial):
                assert np.array_equal(gc, sc) and np.array_equal(gd, sd)


_FORK_SCRIPT = r"""
import multiprocessing as mp, sys
import numpy as np
sys.path.insert(0, %r)
import image_classification_amd as ic            # import only: nothing touches the GPU yet
from image_classification_amd.synthetic import sift_like

CODEBOOK = sift_like(300, seed=99).astype(np.float64)   # "the model", built before the fork

def work(seed):
    code, dist = ic.vq(sift_like(2000, seed=seed), CODEBOOK)
    return seed, code

if __name__ == "__main__":
    with mp.get_context("fork").Pool(2) as pool:
        res = pool.map(work, [1, 2, 3, 4])
    import oracle
    for seed, code in res:
        ref, _ = oracle.vq(sift_like(2000, seed=seed), CODEBOOK)
        assert (code == ref).mean() > 0.999, seed
    print("FORK_OK")
"""


def test_workers_forked_after_model_load():
    out = subprocess.run([sys.executable, "-c", _FORK_SCRIPT % ROOT], capture_output=True,
                         text=True, timeout=110)
    assert out.returncode == 0 and "FORK_OK" in out.stdout, out.stderr[-2000:]
