"""A backend for the host-logic tests on machines without a GPU (TEST INFRASTRUCTURE).

It implements the interface of image_classification_amd.backend.HipBackend with the float64 CPU
oracle and torch CPU tensors, so that batching, sharding, the Lloyd loop and the collectives can be
exercised by `pytest -m "not gpu"` and by the gloo multi-process tests.  It lives under tests/ on
purpose: the package never imports it and has no CPU path of its own.
"""
import numpy as np
import torch

import oracle


class _Codebook(object):
    def __init__(self, c):
        self.c = c
        self.sqnorm = (c.double() ** 2).sum(1).float()
        self.device = c.device

    @property
    def shape(self):
        return tuple(self.c.shape)


class CheckerBackend(object):
    name = "cpu-oracle-checker"

    def __init__(self):
        self.device = torch.device("cpu")
        self.calls = []

    def f32(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(torch.float32).contiguous()
        return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))

    def i32(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.int32)

    def i64(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.int64)

    def codebook(self, c):
        if isinstance(c, _Codebook):
            return c
        return _Codebook(self.f32(np.asarray(c)))

    def make_codebook(self, c_dev):
        return _Codebook(c_dev)

    def _top(self, x, cb, kk):
        self.calls.append(("search", tuple(x.shape), cb.shape))
        return oracle.topk(x.numpy(), cb.c.numpy(), kk)

    def assign_top1(self, x, cb, want_dist=True, hint=None):
        d, i = self._top(x, cb, 1)
        idx = torch.from_numpy(i[:, 0].astype(np.int32))
        return idx, (torch.from_numpy(d[:, 0].astype(np.float32)) if want_dist else None)

    def assign_top1_np(self, x_np, cb, want_dist=True):
        return self.assign_top1(self.f32(x_np), cb, want_dist)

    def assign_top1_parts(self, parts, cb, want_dist=True):
        return self.assign_top1(self.f32(np.concatenate(parts)), cb, want_dist)

    def assign_top2_parts(self, parts, cb, ratio=0.8):
        return self.assign_top2(self.f32(np.concatenate(parts)), cb, ratio)

    def assign_top2_np(self, x_np, cb, ratio=0.8):
        return self.assign_top2(self.f32(x_np), cb, ratio)

    def assign_top2(self, x, cb, ratio=0.8):
        d, i = self._top(x, cb, 2)
        d32 = d.astype(np.float32)
        mask = (d32[:, 0] < d32[:, 1] * np.float32(ratio)).astype(np.uint8)
        return (torch.from_numpy(i[:, 0].astype(np.int32)), torch.from_numpy(d32[:, 0].copy()),
                torch.from_numpy(i[:, 1].astype(np.int32)), torch.from_numpy(d32[:, 1].copy()),
                torch.from_numpy(mask))

    def assign_top2_rec(self, x, cb):
        d, i = self._top(x, cb, 2)
        rec = np.empty((x.shape[0], 4), dtype=np.int32)
        rec[:, 0] = d[:, 0].astype(np.float32).view(np.int32)
        rec[:, 1] = i[:, 0]
        rec[:, 2] = d[:, 1].astype(np.float32).view(np.int32)
        rec[:, 3] = i[:, 1]
        return torch.from_numpy(rec)

    def top2_merge(self, gathered, shard_base, ratio=0.8):
        g = gathered.numpy()
        if shard_base is None:       # bases packed into a trailing record per shard
            shard_base = [int(np.ascontiguousarray(g[s, -1, :2]).view(np.int64)[0])
                          for s in range(g.shape[0])]
            g = g[:, :-1]
        shards = []
        for s in range(g.shape[0]):
            d = np.stack([g[s, :, 0].copy().view(np.float32), g[s, :, 2].copy().view(np.float32)], 1)
            i = np.stack([g[s, :, 1], g[s, :, 3]], 1).astype(np.int64)
            shards.append((d.astype(np.float64), i))
        d, i = oracle.top2_merge(shards, list(shard_base))
        d32 = d.astype(np.float32)
        mask = (d32[:, 0] < d32[:, 1] * np.float32(ratio)).astype(np.uint8)
        return (torch.from_numpy(i[:, 0].copy()), torch.from_numpy(d32[:, 0].copy()),
                torch.from_numpy(i[:, 1].copy()), torch.from_numpy(d32[:, 1].copy()),
                torch.from_numpy(mask))

    def hist(self, codes, offsets, k, norm="l1", power=0.5):
        self.calls.append(("hist", int(codes.shape[0]), int(offsets.shape[0]) - 1))
        c = codes.numpy()
        off = offsets.numpy()
        per = [c[off[i]:off[i + 1]] for i in range(len(off) - 1)]
        return torch.from_numpy(oracle.build_histogram(per, int(k), norm, power).reshape(len(per), int(k)))

    def f64(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.float64)

    def spatial_hist(self, codes, kp_cells, cell_weights, k):
        c = codes.numpy().astype(np.int64)
        cells = kp_cells.numpy()
        w = cell_weights.numpy()
        out = np.zeros((c.shape[0], len(w), int(k)))
        rows = np.arange(c.shape[0])
        for lev in range(cells.shape[0]):
            for kp in range(cells.shape[1]):
                if cells[lev, kp] >= 0:
                    out[rows, cells[lev, kp], c[:, kp]] += 1
        out *= w[None, :, None]
        return torch.from_numpy(out.reshape(c.shape[0], -1))

    def centroid_partials(self, x, codes, k, out=None):
        _, _, sums, counts = oracle.update_cluster_means(x.numpy(), codes.numpy().astype(np.int64), int(k))
        sums, counts = torch.from_numpy(sums), torch.from_numpy(counts.astype(np.int64))
        if out is not None:
            out[0].copy_(sums)
            out[1].copy_(counts)
            return out
        return sums, counts

    def centroid_finalize(self, sums, counts, prev_c):
        has = counts > 0
        new = prev_c.clone()
        new[has] = (sums[has] / counts[has][:, None].double()).float()
        return new, has.to(torch.uint8)

    def pairwise(self, x, c, metric="euclidean"):
        self.calls.append(("pairwise", tuple(x.shape), tuple(c.shape), metric))
        return torch.from_numpy(oracle.pairwise_metric(x.numpy(), c.numpy(), metric))

    def knn_rows(self, x, c, k, metric="euclidean"):
        self.calls.append(("knn_rows", tuple(x.shape), tuple(c.shape), metric, k))
        d, i = oracle.kneighbors_metric(c.numpy(), x.numpy(), k, metric)
        return torch.from_numpy(d), torch.from_numpy(i.astype(np.int32))

    def to_host(self, t):
        return t.numpy()

    def u8(self, a):
        return torch.as_tensor(np.asarray(a), dtype=torch.uint8)

    def compact_matches(self, mask, i1, row_offsets, kp_query=None, kp_rows=None):
        m = mask.numpy().astype(bool)
        off = row_offsets.numpy()
        rows = m.shape[0]
        out_off = np.zeros(len(off), dtype=np.int64)
        mq = np.zeros(rows, dtype=np.int32)
        mi = np.zeros(rows, dtype=np.int32)
        pts = np.zeros((rows, 4)) if kp_query is not None else None
        pos = 0
        for c in range(len(off) - 1):
            local = np.nonzero(m[off[c]:off[c + 1]])[0]
            n = len(local)
            mq[pos:pos + n] = i1.numpy()[off[c] + local]
            mi[pos:pos + n] = local
            if pts is not None:
                pts[pos:pos + n, :2] = kp_query.numpy()[mq[pos:pos + n]]
                pts[pos:pos + n, 2:] = kp_rows.numpy()[off[c] + local]
            pos += n
            out_off[c + 1] = pos
        return (torch.from_numpy(out_off), torch.from_numpy(mq), torch.from_numpy(mi),
                None if pts is None else torch.from_numpy(pts))

    def ransac_homography(self, pts, offsets, n_iter, th, samples=None, seed=0, min_inliers=10,
                          accept_mean_err=True, cv_rules=False):
        from oracle import ransac_ref
        off = offsets.numpy()
        n_cand = len(off) - 1
        H = np.zeros((n_cand, 3, 3))
        n_inl = np.zeros(n_cand, dtype=np.int32)
        err = np.full(n_cand, np.inf)
        best_it = np.full(n_cand, -1, dtype=np.int32)
        mask = np.zeros(pts.shape[0], dtype=np.uint8)
        for c in range(n_cand):
            data = pts.numpy()[off[c]:off[c + 1]]
            n = data.shape[0]
            if n < max(4, min_inliers) or n_iter < 1:
                continue
            if samples is not None:
                parts = []
                for it in range(n_iter):
                    cand = samples.numpy()[c, it].astype(np.int64)
                    parts.append((cand, np.setdiff1d(np.arange(n), cand)))
            else:
                rng = np.random.RandomState((int(seed) + 7919 * c) % (2 ** 32))
                parts = [ransac_ref.random_partition(4, n, rng) for _ in range(n_iter)]
            info = {}
            if cv_rules:
                fit, m = ransac_ref.cv_find_homography_ransac(
                    data, th, [p_[0] for p_ in parts], max_iters=n_iter, info=info)
                e = np.inf if fit is None else \
                    float(ransac_ref.HomographyModel.get_error(data[m.astype(bool)], fit).mean())
                if fit is not None and m.sum() < min_inliers:
                    fit, m = None, np.zeros_like(m)
            else:
                fit, m, e = ransac_ref.ransac(
                    data, ransac_ref.HomographyModel, 4, 0, th, partitions=parts, num_it=n_iter,
                    min_inliers=min_inliers, accept_mean_err=accept_mean_err, info=info)
            err[c] = e
            best_it[c] = info["best_it"]
            if fit is not None:
                H[c] = fit
                n_inl[c] = int(m.sum())
                mask[off[c]:off[c + 1]] = m
        return (torch.from_numpy(H), torch.from_numpy(n_inl), torch.from_numpy(err),
                torch.from_numpy(best_it), torch.from_numpy(mask))
