"""Worker of tests/test_zy_two_ranks_gpu.py: two ranks share the one GPU of the test box and talk
over gloo (RCCL refuses two ranks on one device), so the sharded code paths run with the real
kernels.  Every sharded result is compared with the unsharded run on the same device."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from image_classification_amd import distributed as D  # noqa: E402
from image_classification_amd.backend import HipBackend  # noqa: E402
from image_classification_amd.synthetic import sift_like  # noqa: E402


def main():
    # a rank stuck in a collective or a kernel ends itself with a traceback instead of hanging
    import faulthandler
    faulthandler.dump_traceback_later(90, exit=True)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    b = HipBackend(torch.device("cuda", 0))

    # 1. BoW quantisation: images sharded, no collective except the optional gather
    sizes = [300, 0, 1200, 50, 700, 900, 10, 400]
    offsets = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    x = sift_like(int(offsets[-1]), seed=1)
    vocab = sift_like(300, seed=2)
    bounds = D.shard_images(offsets, world)
    lo_i, hi_i = int(bounds[rank]), int(bounds[rank + 1])
    lo, hi = int(offsets[lo_i]), int(offsets[hi_i])
    codes, hists, gathered = D.quantize_sharded(x[lo:hi], offsets[lo_i:hi_i + 1] - lo, vocab,
                                                backend=b, gather_to=0)
    full_codes, full_hists, _ = D.quantize_sharded(x, offsets, vocab, backend=b)
    assert torch.equal(codes, full_codes[lo:hi]) and torch.equal(hists, full_hists[lo_i:hi_i])
    if rank == 0:
        assert torch.equal(torch.cat(gathered), full_hists)
    else:
        assert gathered is None

    # 2. k-means: rows sharded, one all-reduce per iteration; identical centroids on all ranks and
    #    equal (to summation order) to the unsharded run
    data = sift_like(6000, n_modes=64, seed=3)
    init = data[:40].copy()
    rb = D.shard_bounds(len(data), world)
    cb, labels, _ = D.kmeans_fit_sharded(data[rb[rank]:rb[rank + 1]], init, 4, backend=b)
    from image_classification_amd.clustering import lloyd
    ref_cb, ref_labels, _ = lloyd(b.f32(data), b.f32(init), 4, b)
    assert torch.equal(labels, ref_labels[rb[rank]:rb[rank + 1]])
    assert torch.allclose(cb, ref_cb, rtol=1e-6, atol=1e-4)
    mine = cb.cpu()
    other = mine.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(mine, other)

    # 3. 2-NN with the index sharded: all-gather of the records + merge == unsharded search
    index = sift_like(5000, n_modes=512, seed=4)
    q = sift_like(700, seed=5)
    ib = D.shard_bounds(len(index), world)
    got = D.knn2_index_sharded(q, index[ib[rank]:ib[rank + 1]], int(ib[rank]), 0.8, backend=b)
    ref = b.assign_top2(b.f32(q), b.codebook(index), 0.8)
    same = (got[0] == ref[0].long()) & (got[2] == ref[2].long())
    assert same.float().mean() > 0.999
    assert torch.equal(got[1][same], ref[1][same])

    # 4. 2-NN with the search rows sharded (the reference's direction): results concatenate
    rows = sift_like(3001, seed=6)
    sb = D.shard_bounds(len(rows), world)
    res = D.knn2_rows_sharded(q, rows[sb[rank]:sb[rank + 1]], 0.8, backend=b, gather=True)
    full = b.assign_top2(b.f32(rows), b.codebook(q), 0.8)
    for a, c in zip(res, full):
        assert torch.equal(a, c)

    dist.barrier()
    if rank == 0:
        print("DIST_GPU_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
