"""The N > 1 path with world_size 2 on CPU (gloo): sharding arithmetic, the packed all-reduce of
the Lloyd loop, the all-gather + merge of index-sharded 2-NN.  Compute comes from the checker
backend (tests/checker_backend.py); the collectives and host logic are the shipped ones."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world_size, port, out_dir):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        import oracle
        from checker_backend import CheckerBackend
        from image_classification_amd import distributed as D
        from image_classification_amd.synthetic import image_offsets, sift_like

        b = CheckerBackend()
        res = {}

        # --- k-means: rows sharded, packed all-reduce per iteration ---------------------------
        data = sift_like(1200, n_modes=12, seed=1)
        init = data[np.random.default_rng(2).choice(1200, 20, replace=False)].copy()
        init[7] = 5e3                                   # stays empty on every rank
        bounds = D.shard_bounds(1200, world_size)
        lo, hi = bounds[rank], bounds[rank + 1]
        empties_seen = []
        cb, labels, empties = D.kmeans_fit_sharded(data[lo:hi], init if rank == 0 else init * 0,
                                                   4, backend=b,
                                                   on_empty=lambda: empties_seen.append(1))
        ref_cb, ref_label, _ = oracle.kmeans2_matrix(data, init, 4)
        res["kmeans_cb_err"] = float(np.abs(cb.numpy() - ref_cb.astype(np.float32)).max())
        res["kmeans_labels_ok"] = bool(np.array_equal(labels.numpy(), ref_label[lo:hi]))
        res["kmeans_empty_kept"] = bool(np.all(cb.numpy()[7] == np.float32(5e3)))
        res["kmeans_empties"] = empties
        res["kmeans_on_empty_calls"] = len(empties_seen)
        both = [torch.zeros_like(cb) for _ in range(world_size)]
        dist.all_gather(both, cb)
        res["kmeans_identical_on_ranks"] = bool(torch.equal(both[0], both[1]))

        # --- 2-NN, index sharded: all-gather + merge ---------------------------------------------
        queries = sift_like(150, seed=3)
        index = sift_like(901, seed=4)
        ib = D.shard_bounds(901, world_size)
        i1, d1, i2, d2, mask = D.knn2_index_sharded(queries, index[ib[rank]:ib[rank + 1]],
                                                    ib[rank], 0.8, backend=b)
        rd, ri = oracle.topk(queries, index, 2)
        res["knn_index_ok"] = bool(np.array_equal(i1.numpy(), ri[:, 0]) and
                                   np.array_equal(i2.numpy(), ri[:, 1]))
        res["knn_dist_err"] = float(np.abs(d1.numpy() - rd[:, 0]).max() / rd[:, 0].max())
        res["knn_mask_ok"] = bool(np.array_equal(mask.numpy().astype(bool),
                                                 rd[:, 0].astype(np.float32) <
                                                 rd[:, 1].astype(np.float32) * np.float32(0.8)))

        # --- 2-NN, rows sharded: concatenation ------------------------------------------------------
        rb = D.shard_bounds(901, world_size)
        g = D.knn2_rows_sharded(queries, index[rb[rank]:rb[rank + 1]], 0.8, backend=b, gather=True)
        rd2, ri2 = oracle.topk(index, queries, 2)
        res["knn_rows_ok"] = bool(np.array_equal(g[0].numpy(), ri2[:, 0]) and g[0].shape[0] == 901)

        # --- BoW quantisation: images sharded, no collective in the data path -----------------------
        offsets = image_offsets(9, 40, ragged=True, seed=5)
        x = sift_like(int(offsets[-1]), seed=6)
        vocab = sift_like(30, seed=7)
        img_b = D.shard_images(offsets, world_size)
        r0, r1 = offsets[img_b[rank]], offsets[img_b[rank + 1]]
        local_off = offsets[img_b[rank]:img_b[rank + 1] + 1] - r0
        n_coll_before = len(b.calls)
        codes, hists, gathered = D.quantize_sharded(x[r0:r1], local_off, vocab, backend=b,
                                                    gather_to=0)
        code_ref, _ = oracle.vq(x, vocab)
        per = [code_ref[offsets[i]:offsets[i + 1]] for i in range(9)]
        ref_h = oracle.build_histogram(per, 30)
        res["bow_local_ok"] = bool(np.array_equal(hists.numpy(), ref_h[img_b[rank]:img_b[rank + 1]]))
        if rank == 0:
            res["bow_gather_ok"] = bool(np.array_equal(torch.cat(gathered).numpy(), ref_h))
        else:
            res["bow_gather_ok"] = gathered is None
        res["bow_calls"] = [c[0] for c in b.calls[n_coll_before:]]
        torch.save(res, os.path.join(out_dir, "rank%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        res = torch.load(os.path.join(str(tmp_path), "rank%d.pt" % rank), weights_only=True)
        assert res["kmeans_cb_err"] <= 1e-3, res           # float32 rounding of ~500-norm centroids
        assert res["kmeans_labels_ok"] and res["kmeans_empty_kept"]
        assert res["kmeans_identical_on_ranks"]
        assert res["kmeans_empties"] and min(res["kmeans_empties"]) >= 1
        assert res["kmeans_on_empty_calls"] == 4
        assert res["knn_index_ok"] and res["knn_mask_ok"] and res["knn_dist_err"] < 1e-6
        assert res["knn_rows_ok"]
        assert res["bow_local_ok"] and res["bow_gather_ok"]
        assert res["bow_calls"] == ["search", "hist"]


def test_shard_helpers():
    from image_classification_amd import distributed as D
    assert list(D.shard_bounds(10, 3)) == [0, 4, 7, 10]
    assert list(D.shard_bounds(2, 4)) == [0, 1, 2, 2, 2]
    off = np.array([0, 10, 10, 30, 35, 100], dtype=np.int64)
    b = D.shard_images(off, 2)
    assert b[0] == 0 and b[-1] == 5 and np.all(np.diff(b) >= 0)
    b8 = D.shard_images(off, 8)
    assert len(b8) == 9 and b8[-1] == 5 and np.all(np.diff(b8) >= 0)
    assert D.world() == (0, 1)
